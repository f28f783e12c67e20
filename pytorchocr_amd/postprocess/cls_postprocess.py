"""ClsPostProcess: mirror of reference pytocr/postprocess/cls_postprocess.py:4-21 (arg-max over the two direction classes)."""
import torch


class ClsPostProcess(object):
    def __init__(self, label_list, **kwargs):
        super().__init__()
        self.label_list = label_list

    def __call__(self, preds, label=None, *args, **kwargs):
        if isinstance(preds, torch.Tensor):
            preds = preds.detach().cpu().numpy()
        best = preds.argmax(axis=1)
        decode_out = [(self.label_list[k], preds[row, k]) for row, k in enumerate(best)]
        if label is None:
            return decode_out
        return decode_out, [(self.label_list[k], 1.0) for k in label]
