"""End-to-end OCR: counterpart of reference deploy/pytorch/run_ocr.py (`OCRer`, :51-231): detect -> sort boxes ->
per box perspective crop (`get_part_img`), rotate 90 deg when h >= 1.5 w -> recognise.

The reference recognises every box with a batch-1 CRNN forward and one device->host sync per box
(run_ocr.py:187-229); here all crops of an image go through the CRNN as ONE batch.  Per-box results are the same
(the network is batch-independent).  The optional direction classifier (cls) is not built."""
import os

import numpy as np
import torch

from ..utils.warp import get_part_img
from .common import read_image_bgr
from .infer_det import Deter
from .infer_rec import Recer


class OCRer(object):
    def __init__(self, det_cfg, det_ckpt, rec_cfg, rec_ckpt, cls_cfg=None, cls_ckpt=None, character_dict_path=None, gpu_id=0) -> None:
        if cls_cfg is not None and cls_ckpt is not None:
            raise NotImplementedError("pytorchocr_amd run_ocr: the optional direction classifier is outside the built hot path")
        self.det = Deter(det_cfg, det_ckpt, gpu_id)
        self.rec = Recer(rec_cfg, rec_ckpt, character_dict_path, gpu_id)

    @torch.no_grad()
    def run(self, img_path):
        img = read_image_bgr(img_path)
        boxes = self.det.run(img)
        crops = []
        for box in boxes:
            part_img = get_part_img(img, box)
            h, w = part_img.shape[:2]
            if h >= 1.5 * w:
                part_img = np.rot90(part_img, 1)
            crops.append(np.ascontiguousarray(part_img))
        texts = self.rec.run_batch(crops)
        return [[box, text, prob] for box, (text, prob) in zip(boxes, texts)]
