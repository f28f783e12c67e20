"""Detection entry point: counterpart of reference deploy/pytorch/infer_det.py (`Deter`, :46-103; `main`, :107-145).

    python -m pytorchocr_amd.deploy.infer_det --config <yml> --model_path <ckpt> --img_path <file|dir> [--out_dir d]

Same flow: load_config -> build_model -> load_pretrained_params -> build_post_process -> create_operators; per image
decode -> DetResizeForTest/ToTensor/Normalize -> model -> DBPostProcess -> sort_boxes.  `run_batch` additionally
feeds a list of equally-sized images through the model as ONE batch (the MI355X path is batch-first)."""
import argparse
import os
from pathlib import Path

import numpy as np
import torch

from ..data import create_operators, transform
from ..modeling.architectures import build_model
from ..postprocess import build_post_process
from ..utils.config import load_config
from ..utils.save_load import load_pretrained_params
from ..utils.utility import sort_boxes
from .common import inference_transforms, read_image_bgr


class Deter(object):
    def __init__(self, det_cfg, det_ckpt=None, gpu_id=0, gpu_preprocess=False) -> None:
        self.gpu_preprocess = gpu_preprocess
        det_cfg = load_config(det_cfg) if isinstance(det_cfg, (str, os.PathLike)) else det_cfg
        det_cfg["Global"]["distributed"] = False
        deter = build_model(det_cfg["Architecture"])
        if not (det_cfg["Global"].get("use_gpu", True) and torch.cuda.is_available()):
            raise RuntimeError("pytorchocr_amd needs a ROCm GPU (Global.use_gpu and torch.cuda.is_available()); no CPU path")
        self.det_device = torch.device("cuda:{}".format(gpu_id))
        deter = deter.to(self.det_device).eval()
        if det_ckpt is not None:
            deter = load_pretrained_params(deter, det_ckpt)
        self.deter = deter
        self.det_post_process_class = build_post_process(det_cfg["PostProcess"], det_cfg["Global"])
        det_transforms, mode = inference_transforms(det_cfg, ["image", "shape"])
        self.det_img_mode = mode or "RGB"
        self.det_ops = create_operators(det_transforms, det_cfg["Global"])

    def _prep(self, img):
        det_img = img[:, :, ::-1] if self.det_img_mode == "RGB" else img.copy()
        return transform({"image": np.ascontiguousarray(det_img)}, self.det_ops)

    def _gpu_ops(self):
        from ..data.imaug import DetResizeForTest, Normalize
        rs = [o for o in self.det_ops if isinstance(o, DetResizeForTest)]
        nm = [o for o in self.det_ops if isinstance(o, Normalize)]
        if len(rs) != 1 or len(nm) != 1:
            raise NotImplementedError("gpu_preprocess needs exactly DetResizeForTest + ToTensor + Normalize in the transform list")
        return rs[0], nm[0]

    @torch.no_grad()
    def run_gpu(self, img_path):
        """same result as run(), with resize / normalise / layout on the GPU (only the u8 image crosses PCIe)"""
        from ..data.gpu_preprocess import det_preprocess
        img = read_image_bgr(img_path)
        rs, nm = self._gpu_ops()
        src_h, src_w = img.shape[:2]
        rh, rw = rs.target_size(src_h, src_w)
        x4 = det_preprocess(img, (rh, rw), nm.mean, nm.std, self.det_device, swap_rb=self.det_img_mode == "RGB")
        shape = np.array([[src_h, src_w, rh / float(src_h), rw / float(src_w)]])
        res = self.det_post_process_class(self.deter.forward_nhwc4(x4), shape)
        return sort_boxes(res[0]["points"])

    @torch.no_grad()
    def run(self, img_path):
        if self.gpu_preprocess:
            return self.run_gpu(img_path)
        img = read_image_bgr(img_path)
        det_batch = self._prep(img)
        det_img = det_batch[0].unsqueeze(dim=0).to(self.det_device)
        det_shape_list = np.expand_dims(det_batch[1], axis=0)
        det_preds = self.deter(det_img)
        det_post_result = self.det_post_process_class(det_preds, det_shape_list)
        return sort_boxes(det_post_result[0]["points"])

    @torch.no_grad()
    def run_batch(self, imgs):
        batches = [self._prep(read_image_bgr(i)) for i in imgs]
        sizes = {tuple(b[0].shape) for b in batches}
        if len(sizes) != 1:
            return [self.run(i) for i in imgs]
        x = torch.stack([b[0] for b in batches]).to(self.det_device)
        shapes = np.stack([b[1] for b in batches])
        res = self.det_post_process_class(self.deter(x), shapes)
        return [sort_boxes(r["points"]) for r in res]


def main():
    ap = argparse.ArgumentParser(description="pytorchocr_amd det_model infer")
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--model_path", type=str, default=None)
    ap.add_argument("--img_path", type=str, required=True)
    ap.add_argument("--out_dir", type=str, default="./output")
    ap.add_argument("--gpu_id", type=int, default=0)
    args = ap.parse_args()
    deter = Deter(args.config, args.model_path, args.gpu_id)
    assert os.path.exists(args.img_path), "img_path not exists"
    paths = [Path(args.img_path)] if os.path.isfile(args.img_path) else sorted(Path(args.img_path).glob("*.[jp][pn]g"))
    out_dir = Path(args.out_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    for p in paths:
        boxes = deter.run(str(p))
        with open(str(out_dir.joinpath("res_" + p.stem + ".txt")), "w", encoding="UTF-8") as fp:
            for box in boxes:
                fp.write(",".join(str(c) for c in box.reshape(-1).tolist()) + "\n")


if __name__ == "__main__":
    main()
