"""CTC label decode.  Mirror of reference pytocr/postprocess/rec_postprocess.py: `BaseRecLabelDecode` (:5-62:
dictionary load, blank skip FIRST then duplicate collapse against the previous RAW index, np.mean confidence,
nan for empty) and `CTCLabelDecode` (:65-93: "blank" prepended, argmax / max over classes).

The class reductions run on the GPU (`ptocr_ctc_greedy_f32`); what is left on the host is the per-line string
assembly over B*T int32 indices.  Accepted inputs:
  * the reference contract: softmax probabilities f32[T,B,C] (torch tensor on any device, numpy, or a tuple
    whose last element is one), or
  * the fast path: (idx int32[B,T], prob f32[B,T]) from `BaseModel.forward_greedy`.
"""
import numpy as np
import torch

from ..modeling import ops


class BaseRecLabelDecode(object):
    """ Convert between text-label and text-index """

    def __init__(self, character_dict_path=None, use_space_char=False):
        self.beg_str = "sos"
        self.end_str = "eos"
        self.character_str = []
        if character_dict_path is None:
            self.character_str = "0123456789abcdefghijklmnopqrstuvwxyz"
            dict_character = list(self.character_str)
        else:
            with open(character_dict_path, "rb") as fin:
                for line in fin.readlines():
                    self.character_str.append(line.decode("UTF-8").strip("\n").strip("\r\n"))
            if use_space_char:
                self.character_str.append(" ")
            dict_character = list(self.character_str)
        dict_character = self.add_special_char(dict_character)
        self.dict = {char: i for i, char in enumerate(dict_character)}
        self.character = dict_character

    def add_special_char(self, dict_character):
        return dict_character

    def get_ignored_tokens(self):
        return [0]  # for ctc blank

    def decode(self, text_index, text_prob=None, is_remove_duplicate=False):
        """ convert text-index into text-label. """
        text_index = np.asarray(text_index)
        ignored = self.get_ignored_tokens()
        keep = ~np.isin(text_index, ignored)
        if is_remove_duplicate and text_index.shape[1] > 1:
            keep[:, 1:] &= text_index[:, 1:] != text_index[:, :-1]
        chars = self.character
        result_list = []
        for b in range(text_index.shape[0]):
            k = keep[b]
            text = "".join([chars[int(i)] for i in text_index[b][k]])
            if text_prob is not None:
                conf_list = np.asarray(text_prob[b])[k]
            else:
                conf_list = np.ones(int(k.sum()))
            with np.errstate(all="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    conf = np.mean(conf_list.tolist()) if len(conf_list) else np.mean([])
            result_list.append((text, conf))
        return result_list


class CTCLabelDecode(BaseRecLabelDecode):
    """ Convert between text-label and text-index """

    def __init__(self, character_dict_path=None, use_space_char=False, **kwargs):
        super(CTCLabelDecode, self).__init__(character_dict_path, use_space_char)

    def __call__(self, preds, label=None, *args, **kwargs):
        if isinstance(preds, tuple) and len(preds) == 2 and torch.is_tensor(preds[0]) and preds[0].dtype in (torch.int32, torch.int64) \
                and preds[0].dim() == 2:
            preds_idx = preds[0].cpu().numpy()
            preds_prob = preds[1].cpu().numpy()
        else:
            if isinstance(preds, tuple):
                preds = preds[-1]
            if isinstance(preds, np.ndarray):
                preds = torch.from_numpy(np.ascontiguousarray(preds, np.float32))
            if not preds.is_cuda:
                preds = preds.to("cuda:%d" % torch.cuda.current_device())     # host buffers accepted; work stays on the GPU
            preds = preds.contiguous().float()
            T, B, Cn = preds.shape
            x = preds.reshape(T * B, Cn)
            if Cn % 4:                                                         # kernel wants 16-B aligned rows
                xp = torch.zeros((T * B, (Cn + 3) // 4 * 4), dtype=torch.float32, device=x.device)
                xp[:, :Cn] = x
                x = xp
            idx, prob = ops.ctc_greedy(x, Cn, is_prob=True)
            preds_idx = idx.reshape(T, B).cpu().numpy().T
            preds_prob = prob.reshape(T, B).cpu().numpy().T
        ops.lstm_check()                    # the copies above synchronised the stream: a timed-out LSTM exchange surfaces here
        text = self.decode(preds_idx, preds_prob, is_remove_duplicate=True)
        if label is None:
            return text
        label = self.decode(label)
        return text, label

    def submit(self, greedy):
        """Asynchronous form for the device-resident fast path: `greedy` = (idx int32[B,T], prob f32[B,T]) from
        `BaseModel.forward_greedy`.  The two small tensors are copied to pinned host memory on a side stream behind the work
        already queued on the caller's stream; `.result()` waits for that copy only and runs the (host, Python) string
        assembly -- so it overlaps with whatever the caller queues next (the next batch's forward pass).  Same return value
        as `__call__`."""
        idx, prob = greedy
        dev = idx.device
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        h_idx = torch.empty(idx.shape, dtype=idx.dtype, pin_memory=True)
        h_prob = torch.empty(prob.shape, dtype=prob.dtype, pin_memory=True)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            h_idx.copy_(idx, non_blocking=True)
            h_prob.copy_(prob, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        idx.record_stream(self._copy_stream); prob.record_stream(self._copy_stream)
        decode = self.decode

        class _Pending:
            def result(self_inner):
                done.synchronize()
                ops.lstm_check()
                return decode(h_idx.numpy(), h_prob.numpy(), is_remove_duplicate=True)
        return _Pending()

    def add_special_char(self, dict_character):
        dict_character = ["blank"] + dict_character
        return dict_character
