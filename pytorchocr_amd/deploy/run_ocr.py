"""End-to-end OCR: counterpart of reference deploy/pytorch/run_ocr.py (`OCRer`, :51-231): detect -> sort boxes ->
per box perspective crop (`get_part_img`), rotate 90 deg when h >= 1.5 w -> recognise.

The reference recognises every box with a batch-1 CRNN forward and one device->host sync per box
(run_ocr.py:187-229); here all crops of an image go through the CRNN as ONE batch.  Per-box results are the same
(the network is batch-independent).  With a direction classifier (cls_cfg + cls_ckpt, run_ocr.py:130-165,192-211) every crop
is classified first -- again all crops in one batch -- and the crops it calls "180" are rotated by 180 degrees before recognition
(on the GPU paths the rotation is a flag of the recognition pre-process, no copy)."""
import os

import numpy as np
import torch

from ..utils.warp import get_part_img
from .common import read_image_bgr
from .infer_cls import Clser
from .infer_det import Deter
from .infer_rec import Recer


class OCRer(object):
    def __init__(self, det_cfg, det_ckpt, rec_cfg, rec_ckpt, cls_cfg=None, cls_ckpt=None, character_dict_path=None, gpu_id=0,
                 gpu_preprocess=False) -> None:
        self.gpu_preprocess = gpu_preprocess
        self.det = Deter(det_cfg, det_ckpt, gpu_id, gpu_preprocess=gpu_preprocess)
        self.rec = Recer(rec_cfg, rec_ckpt, character_dict_path, gpu_id)
        # the reference builds the classifier only when BOTH are given (run_ocr.py:131); a dict config stands in for the yml path
        self.cls = Clser(cls_cfg, cls_ckpt, gpu_id) if cls_cfg is not None and (cls_ckpt is not None or isinstance(cls_cfg, dict)) else None

    def _cls_flips(self, buf, metas, dev):
        """packed crops -> bool per valid crop: the classifier's label is "180" (one batched forward)"""
        from ..data.gpu_preprocess import cls_preprocess
        from ..data.imaug import ClsResizeImg
        if self.cls.cls_img_mode == "GRAY":
            raise NotImplementedError("the GPU classifier pre-process implements the 3-channel input")
        shape = [o for o in self.cls.cls_ops if isinstance(o, ClsResizeImg)][0].image_shape
        x4 = cls_preprocess(buf, metas, shape, dev, swap_rb=self.cls.cls_img_mode == "RGB")
        if x4.shape[0] == 0:
            return np.zeros(0, bool)
        flips = []
        for c0 in range(0, int(x4.shape[0]), 1024):
            res = self.cls.cls_post_process_class(self.cls.clser.forward_nhwc4(x4[c0:c0 + 1024]))
            flips += [lab == "180" for lab, _ in res]
        return np.asarray(flips, bool)

    @torch.no_grad()
    def run_gpu(self, img_path):
        """detect, crop (batched perspective warp), resize/normalise the crops and recognise, all on the GPU"""
        from ..data.gpu_preprocess import rec_preprocess, warp_crops
        from ..data.imaug import RecResizeImg
        img = read_image_bgr(img_path)
        boxes = self.det.run(img)
        if len(boxes) == 0:
            return []
        if self.rec.rec_img_mode != "GRAY":
            raise NotImplementedError("the GPU recognition pre-process implements the GRAY CRNN input")
        shape = [o for o in self.rec.rec_ops if isinstance(o, RecResizeImg)][0].image_shape
        img_dev = torch.from_numpy(np.ascontiguousarray(img)).to(self.rec.rec_device)
        buf, metas = warp_crops(img_dev, boxes)
        flip = self._cls_flips(buf, metas, self.rec.rec_device) if self.cls is not None else None
        x4 = rec_preprocess(buf, metas, shape, self.rec.rec_device, flip=flip)
        keep = [b for b, m in zip(boxes, metas) if m is not None]
        res = self.rec.rec_post_process_class(self.rec.recer.forward_greedy_nhwc4(x4)) if len(keep) else []
        return [[box, t, round(p, 2)] for box, (t, p) in zip(keep, res)]

    @torch.no_grad()
    def run_batch(self, images, rec_batch=512, stats=None, det_batch=32):
        """The reference's per-image `run` (run_ocr.py:167-231) over a LIST of images as one batched pipeline on the GPU:
        ONE detector forward per group of equally sized images (pre-process of the whole group in one launch), ONE perspective
        crop launch over all boxes of all images, the CRNN over the crops in chunks of at most `rec_batch` lines, results
        regrouped per image: [[box, text, prob], ...] per image, equal to [self.run_gpu(i) for i in images].
        `images`: paths, u8 BGR arrays, or ONE u8[N,H,W,3] device tensor (already resident in HBM)."""
        from ..data.gpu_preprocess import det_preprocess_batch, rec_preprocess, warp_crops_batch
        from ..data.imaug import RecResizeImg
        from ..utils.utility import sort_boxes
        if self.rec.rec_img_mode != "GRAY":
            raise NotImplementedError("the GPU recognition pre-process implements the GRAY CRNN input")
        dev = self.det.det_device
        if torch.is_tensor(images):
            groups = [(list(range(images.shape[0])), images.to(dev))]
            n_img = int(images.shape[0])
        else:
            arrs = [read_image_bgr(i) for i in images]
            n_img = len(arrs)
            by_shape = {}
            for k, a in enumerate(arrs):
                by_shape.setdefault(a.shape, []).append(k)
            groups = [(idx, torch.from_numpy(np.stack([arrs[k] for k in idx])).to(dev)) for idx in by_shape.values()]
        rs, nm = self.det._gpu_ops()
        shape = [o for o in self.rec.rec_ops if isinstance(o, RecResizeImg)][0].image_shape
        out = [None] * n_img
        n_boxes = n_lines = 0
        # Software pipeline over sub-groups of at most `det_batch` images: the detector of sub-group i+1 is queued BEFORE the host stages of
        # sub-group i (box order, perspective solves, crop planning: ~10 ms per 32 images of one core) and those of i+1 run while the
        # GPU recognises the lines of i; every recogniser chunk is queued without waiting, the texts are collected at the end.
        subs = []
        for idx, stack in groups:
            for c0 in range(0, len(idx), det_batch):
                subs.append((idx[c0:c0 + det_batch], stack[c0:c0 + det_batch]))

        def detect(sub):
            idx, stack = sub
            src_h, src_w = int(stack.shape[1]), int(stack.shape[2])
            rh, rw = rs.target_size(src_h, src_w)
            x4 = det_preprocess_batch(stack, (rh, rw), nm.mean, nm.std, swap_rb=self.det.det_img_mode == "RGB")
            shapes = np.array([[src_h, src_w, rh / float(src_h), rw / float(src_w)]] * len(idx))
            return self.det.det_post_process_class.submit(self.det.deter.forward_nhwc4(x4), shapes)

        def recognise(sub, fut):
            idx, stack = sub
            boxes = [sort_boxes(r["points"]) for r in fut.result()]
            buf, metas = warp_crops_batch(stack, boxes)
            flat = [m for per in metas for m in per]
            flip = self._cls_flips(buf, flat, dev) if self.cls is not None else None
            x_rec = rec_preprocess(buf, flat, shape, dev, flip=flip)
            futs = [self.rec.rec_post_process_class.submit(self.rec.recer.forward_greedy_nhwc4(x_rec[c0:c0 + rec_batch]))
                    for c0 in range(0, int(x_rec.shape[0]), rec_batch)]         # decode runs on the label decoder's worker as each chunk lands
            return boxes, metas, futs

        pending = []
        det_fut = detect(subs[0]) if subs else None
        for i, sub in enumerate(subs):
            nxt = detect(subs[i + 1]) if i + 1 < len(subs) else None
            pending.append((sub[0],) + recognise(sub, det_fut))
            det_fut = nxt
        for idx, boxes, metas, futs in pending:
            texts = [t for f in futs for t in f.result()]
            # round(prob, 2) of run_ocr.py:228 for all lines at once: np.round on the array is the same ufunc round() calls on a numpy
            # scalar (one call per line cost 37 ms per 11 000 lines)
            probs = list(np.round(np.array([p for _, p in texts], dtype=np.float64), 2)) if texts else []
            words = [t for t, _ in texts]
            pos = 0
            for k, bx, per in zip(idx, boxes, metas):
                kept = [b for b, m in zip(bx, per) if m is not None] if None in per else bx
                out[k] = [[b, t, p] for b, t, p in zip(kept, words[pos:pos + len(kept)], probs[pos:pos + len(kept)])]
                pos += len(kept)
                n_boxes += len(bx)
            n_lines += len(texts)
        if stats is not None:
            stats["boxes"] = stats.get("boxes", 0) + n_boxes
            stats["lines"] = stats.get("lines", 0) + n_lines
        return out

    @torch.no_grad()
    def run(self, img_path):
        if self.gpu_preprocess:
            return self.run_gpu(img_path)
        img = read_image_bgr(img_path)
        boxes = self.det.run(img)
        crops = []
        for box in boxes:
            part_img = get_part_img(img, box)
            h, w = part_img.shape[:2]
            if h >= 1.5 * w:
                part_img = np.rot90(part_img, 1)
            crops.append(np.ascontiguousarray(part_img))
        if self.cls is not None:
            crops = [np.ascontiguousarray(c[::-1, ::-1]) if lab == "180" else c for c, (lab, _) in zip(crops, self.cls.run_batch(crops))]
        texts = self.rec.run_batch(crops)
        return [[box, text, prob] for box, (text, prob) in zip(boxes, texts)]
