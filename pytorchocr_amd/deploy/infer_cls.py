"""Direction-classifier entry point: counterpart of reference deploy/pytorch/infer_cls.py (`Clser`, :46-106).

`run(img)` keeps the reference contract (one text-line image -> (label, prob rounded to 2 digits)); `run_batch(imgs)` classifies
a list of lines with ONE forward."""
import argparse
import os
from pathlib import Path

import numpy as np
import torch

from ..data import create_operators, transform
from ..data.imaug import bgr_to_gray
from ..modeling.architectures import build_model
from ..postprocess import build_post_process
from ..utils.config import load_config
from ..utils.save_load import load_pretrained_params
from .common import inference_transforms, read_image_bgr


class Clser(object):
    def __init__(self, cls_cfg=None, cls_ckpt=None, gpu_id=0) -> None:
        cls_cfg = load_config(cls_cfg) if isinstance(cls_cfg, (str, os.PathLike)) else cls_cfg
        cls_cfg["Global"]["distributed"] = False
        clser = build_model(cls_cfg["Architecture"])
        if not (cls_cfg["Global"].get("use_gpu", True) and torch.cuda.is_available()):
            raise RuntimeError("pytorchocr_amd needs a ROCm GPU; no CPU path")
        self.cls_device = torch.device("cuda:{}".format(gpu_id))
        clser = clser.to(self.cls_device).eval()
        if cls_ckpt is not None:
            clser = load_pretrained_params(clser, cls_ckpt)
        self.clser = clser
        self.cls_post_process_class = build_post_process(cls_cfg["PostProcess"], cls_cfg["Global"])
        cls_transforms, mode = inference_transforms(cls_cfg, ["image"])
        self.cls_img_mode = mode or "RGB"
        self.cls_ops = create_operators(cls_transforms, cls_cfg["Global"])

    def _prep(self, img):
        if self.cls_img_mode == "GRAY":
            cls_img = bgr_to_gray(img)
        elif self.cls_img_mode == "RGB":
            cls_img = np.ascontiguousarray(img[:, :, ::-1])
        else:
            cls_img = img.copy()
        return transform({"image": cls_img}, self.cls_ops)[0]

    @torch.no_grad()
    def run_batch(self, imgs):
        if len(imgs) == 0:
            return []
        x = torch.stack([self._prep(read_image_bgr(i)) for i in imgs]).to(self.cls_device)
        return [(lab, round(float(p), 2)) for lab, p in self.cls_post_process_class(self.clser(x))]

    @torch.no_grad()
    def run(self, img_path):
        return self.run_batch([img_path])[0]


def main():
    ap = argparse.ArgumentParser(description="pytorchocr_amd cls_model infer")
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--model_path", type=str, default=None)
    ap.add_argument("--img_path", type=str, required=True)
    ap.add_argument("--gpu_id", type=int, default=0)
    args = ap.parse_args()
    clser = Clser(args.config, args.model_path, args.gpu_id)
    paths = [Path(args.img_path)] if os.path.isfile(args.img_path) else sorted(Path(args.img_path).glob("*.[jp][pn]g"))
    for p, (lab, prob) in zip(paths, clser.run_batch([str(p) for p in paths])):
        print(p.name, lab + "," + str(prob))


if __name__ == "__main__":
    main()
