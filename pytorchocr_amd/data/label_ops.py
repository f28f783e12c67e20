"""Label operators of the evaluation data pipeline, with the semantics of reference pytocr/data/imaug/label_ops.py:
`DetLabelEncode` (:19-71), `BaseRecLabelEncode` / `CTCLabelEncode` (:74-177), `ClsLabelEncode` (:6-16).  They run on the host
before the path (needed to demonstrate hmean / accuracy on labelled data through `pytorchocr_amd.eval`)."""
import json
import logging

import numpy as np


class ClsLabelEncode(object):
    """label string -> index in label_list; samples with an unknown label are dropped (None)"""

    def __init__(self, label_list, **kwargs):
        self.label_list = label_list

    def __call__(self, data):
        if data["label"] not in self.label_list:
            return None
        data["label"] = self.label_list.index(data["label"])
        return data


class DetLabelEncode(object):
    """JSON annotation [{"points": [[x, y], ...], "transcription": str}, ...] -> polys f32[K, P, 2] (polygons with fewer points
    are padded by repeating their last point), texts, ignore_tags bool[K] (transcription in ignore_txt); no box -> None."""

    def __init__(self, ignore_txt=("*", "###"), **kwargs):
        self.ignore_txt = list(ignore_txt)

    @staticmethod
    def expand_points_num(boxes):
        most = max(len(b) for b in boxes)
        return [list(b) + [b[-1]] * (most - len(b)) for b in boxes]

    def __call__(self, data):
        items = json.loads(data["label"])
        if len(items) == 0:
            return None
        boxes = [it["points"] for it in items]
        texts = [it["transcription"] for it in items]
        data["polys"] = np.array(self.expand_points_num(boxes), dtype=np.float32)
        data["texts"] = texts
        data["ignore_tags"] = np.array([t in self.ignore_txt for t in texts], dtype=bool)
        return data


class BaseRecLabelEncode(object):
    """text <-> class indices over the dictionary file (one character per line) or, without one, 0-9a-z lower-cased"""

    def __init__(self, max_text_length, character_dict_path=None, use_space_char=False, lower=False, cn2en=False):
        self.max_text_len = max_text_length
        self.beg_str, self.end_str = "sos", "eos"
        self.lower, self.cn2en = lower, cn2en
        if character_dict_path is None:
            logging.getLogger("root").warning("The character_dict_path is None, model can only recognize number and lower letters")
            self.character_str = "0123456789abcdefghijklmnopqrstuvwxyz"
            self.lower = True
        else:
            with open(character_dict_path, "rb") as fin:
                self.character_str = "".join(line.decode("UTF-8").strip("\n").strip("\r\n") for line in fin.readlines())
            if use_space_char:
                self.character_str += " "
        self.character = self.add_special_char(list(self.character_str))
        self.dict = {ch: i for i, ch in enumerate(self.character)}

    def add_special_char(self, dict_character):
        return dict_character

    _CN2EN = str.maketrans({"（": "(", "）": ")", "：": ":", "；": ";", "！": "!", "？": "?"})

    def encode(self, text):
        """text -> list of indices; None for empty / over-long texts and for texts none of whose characters are known
        (unknown characters are skipped with a warning, as the reference does)"""
        if len(text) == 0 or len(text) > self.max_text_len:
            return None
        if self.lower:
            text = text.lower()
        if self.cn2en:
            text = text.translate(self._CN2EN)
        out = []
        for ch in text:
            if ch in self.dict:
                out.append(self.dict[ch])
            else:
                logging.getLogger("root").warning("{} is not in dict".format(ch))
        return out or None


class CTCLabelEncode(BaseRecLabelEncode):
    """label -> indices zero-padded to max_text_length (index 0 is the CTC blank), `length`, and the character histogram
    `label_ace` over all classes (the padding zeros count towards class 0, as in the reference)"""

    def __init__(self, max_text_length, character_dict_path=None, use_space_char=False, cn2en=False, **kwargs):
        # Reference behaviour kept on purpose (label_ops.py:157-158 passes cn2en as the FOURTH POSITIONAL argument of the base
        # class, which is `lower`): cn2en=True lower-cases the label and leaves the full-width punctuation alone.  Pinned by
        # tests/golden/label_encode.json, recorded from the reference class.
        super(CTCLabelEncode, self).__init__(max_text_length, character_dict_path, use_space_char, lower=cn2en, cn2en=False)

    def __call__(self, data):
        idx = self.encode(data["label"])
        if idx is None:
            return None
        data["length"] = np.array(len(idx))
        padded = idx + [0] * (self.max_text_len - len(idx))
        data["label"] = np.array(padded)
        data["label_ace"] = np.bincount(np.asarray(padded, dtype=np.int64), minlength=len(self.character))
        return data

    def add_special_char(self, dict_character):
        return ["blank"] + dict_character
