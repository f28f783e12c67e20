"""Shared pieces of the inference entry points (image decode without cv2, transform-list rewriting)."""
import os

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_DICT = os.path.join(PKG, "utils", "char_dict_6623.txt")


def read_image_bgr(img):
    """path (str / Path) or ndarray -> uint8 BGR HxWx3, what cv2.imdecode(..., IMREAD_COLOR) returns in the reference."""
    if isinstance(img, np.ndarray):
        return img
    from PIL import Image
    with Image.open(str(img)) as im:
        return np.ascontiguousarray(np.array(im.convert("RGB"))[:, :, ::-1])


def inference_transforms(cfg, keep_keys):
    """The deploy scripts drop DecodeImage and *Label* ops and force KeepKeys (reference infer_det.py:66-78)."""
    out, img_mode = [], None
    for op in cfg["Eval"]["dataset"]["transforms"]:
        op_name = list(op)[0]
        if "DecodeImage" in op_name:
            img_mode = op[op_name]["img_mode"]
            continue
        if "Label" in op_name:
            continue
        if op_name == "KeepKeys":
            op = {op_name: dict(op[op_name], keep_keys=list(keep_keys))}
        out.append(op)
    return out, img_mode
