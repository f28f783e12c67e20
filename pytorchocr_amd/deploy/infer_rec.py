"""Recognition entry point: counterpart of reference deploy/pytorch/infer_rec.py (`Recer`, :46-111).

`run(img)` keeps the reference contract (one crop -> (text, prob rounded to 2 digits)); `run_batch(crops)` sends all
crops through the CRNN as ONE batch and decodes from the fused arg-max / max-prob path (no softmax tensor)."""
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
from .common import DEFAULT_DICT, inference_transforms, read_image_bgr


class Recer(object):
    def __init__(self, rec_cfg, rec_ckpt=None, character_dict_path=None, gpu_id=0) -> None:
        rec_cfg = load_config(rec_cfg) if isinstance(rec_cfg, (str, os.PathLike)) else rec_cfg
        rec_cfg["Global"]["distributed"] = False
        if character_dict_path is not None:
            rec_cfg["Global"]["character_dict_path"] = character_dict_path
        elif not rec_cfg["Global"].get("character_dict_path"):
            rec_cfg["Global"]["character_dict_path"] = DEFAULT_DICT
        self.rec_post_process_class = build_post_process(rec_cfg["PostProcess"], rec_cfg["Global"])
        rec_cfg["Architecture"]["Head"]["out_channels"] = len(getattr(self.rec_post_process_class, "character"))
        recer = build_model(rec_cfg["Architecture"])
        if not (rec_cfg["Global"].get("use_gpu", True) and torch.cuda.is_available()):
            raise RuntimeError("pytorchocr_amd needs a ROCm GPU; no CPU path")
        self.rec_device = torch.device("cuda:{}".format(gpu_id))
        recer = recer.to(self.rec_device).eval()
        if rec_ckpt is not None:
            recer = load_pretrained_params(recer, rec_ckpt)
        self.recer = recer
        rec_transforms, mode = inference_transforms(rec_cfg, ["image"])
        self.rec_img_mode = mode or "GRAY"
        self.rec_ops = create_operators(rec_transforms, rec_cfg["Global"])

    def _prep(self, img):
        if self.rec_img_mode == "GRAY":
            rec_img = bgr_to_gray(img)
        elif self.rec_img_mode == "RGB":
            rec_img = np.ascontiguousarray(img[:, :, ::-1])
        else:
            rec_img = img.copy()
        return transform({"image": rec_img}, self.rec_ops)[0]

    @torch.no_grad()
    def run(self, img_path):
        rec_img = self._prep(read_image_bgr(img_path)).unsqueeze(dim=0).to(self.rec_device)
        text, prob_rec = self.rec_post_process_class(self.recer(rec_img))[0]
        return text, round(prob_rec, 2)

    @torch.no_grad()
    def run_batch(self, imgs):
        if len(imgs) == 0:
            return []
        x = torch.stack([self._prep(read_image_bgr(i)) for i in imgs]).to(self.rec_device)
        res = self.rec_post_process_class(self.recer.forward_greedy(x))
        return [(t, round(p, 2)) for t, p in res]


def main():
    ap = argparse.ArgumentParser(description="pytorchocr_amd rec_model infer")
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--model_path", type=str, default=None)
    ap.add_argument("--img_path", type=str, required=True)
    ap.add_argument("--character_dict_path", type=str, default=None)
    ap.add_argument("--gpu_id", type=int, default=0)
    args = ap.parse_args()
    recer = Recer(args.config, args.model_path, args.character_dict_path, args.gpu_id)
    paths = [Path(args.img_path)] if os.path.isfile(args.img_path) else sorted(Path(args.img_path).glob("*.[jp][pn]g"))
    for p, (text, prob) in zip(paths, recer.run_batch([str(p) for p in paths])):
        print(p.name, text, prob)


if __name__ == "__main__":
    main()
