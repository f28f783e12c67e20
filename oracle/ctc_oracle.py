"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the reference CTC greedy decode:
  pytocr/postprocess/rec_postprocess.py:77-89  (transpose T,B,C -> B,T,C; argmax / max over C)
  pytocr/postprocess/rec_postprocess.py:35-59  (skip blank(0) first, then skip if equal to
                                                previous RAW index; conf = np.mean of kept probs)
  pytocr/postprocess/rec_postprocess.py:8-30,91-93 (dictionary: one char per line, "blank" prepended)
Pinned by tests/golden/ctc_decode.json, produced by running the reference class itself
(tools/gen_golden.py, importing rec_postprocess.py by file path).
"""
import numpy as np


def load_characters(character_dict_path=None, use_space_char=False):
    if character_dict_path is None:
        chars = list("0123456789abcdefghijklmnopqrstuvwxyz")
    else:
        chars = []
        with open(character_dict_path, "rb") as fin:
            for line in fin.readlines():
                chars.append(line.decode("UTF-8").strip("\n").strip("\r\n"))
        if use_space_char:
            chars.append(" ")
    return ["blank"] + chars


def greedy_indices(preds_tbc):
    """preds f32[T,B,C] -> (idx int64[B,T], prob f32[B,T]); first-max-index ties like np.argmax."""
    p = np.asarray(preds_tbc).transpose((1, 0, 2))
    return p.argmax(axis=2), p.max(axis=2)


def decode(idx_bt, prob_bt, characters):
    out = []
    for b in range(len(idx_bt)):
        chars, confs = [], []
        for t in range(len(idx_bt[b])):
            k = int(idx_bt[b][t])
            if k == 0:
                continue
            if t > 0 and int(idx_bt[b][t - 1]) == k:
                continue
            chars.append(characters[k])
            confs.append(prob_bt[b][t])
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                conf = np.mean(confs)
        out.append(("".join(chars), conf))
    return out


def ctc_label_decode(preds_tbc, characters):
    idx, prob = greedy_indices(preds_tbc)
    return decode(idx, prob, characters)
