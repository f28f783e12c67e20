"""Direction-classifier post-process: mirror of reference pytocr/postprocess/cls_postprocess.py:4-21.

Input: the classifier's softmax f32[N, len(label_list)] (tensor or array).  Output: per line (label of the most probable class,
its probability); with `label` given (class indices, the eval loop's ground truth) also their (label text, 1.0) pairs."""
import numpy as np
import torch


class ClsPostProcess(object):
    def __init__(self, label_list, **kwargs):
        self.label_list = label_list

    def _names(self, indices):
        return [self.label_list[int(k)] for k in indices]

    def __call__(self, preds, label=None, *args, **kwargs):
        probs = preds.detach().cpu().numpy() if torch.is_tensor(preds) else np.asarray(preds)
        winner = probs.argmax(axis=1)
        top = probs[np.arange(probs.shape[0]), winner]
        decoded = list(zip(self._names(winner), top))
        if label is None:
            return decoded
        return decoded, [(name, 1.0) for name in self._names(label)]
