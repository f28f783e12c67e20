"""torch custom ops over the C ABI (last row): the two post-process entry points a maintainer would call from
torch code, registered under the namespace the survey names.

    import pytorchocr_amd.torch_ops          # registers torch.ops.pytocr_mi355.*
    boxes, counts = torch.ops.pytocr_mi355.db_postprocess(maps, shape_list, 0.3, 0.5, 1.7, 1000, False)
    idx, prob = torch.ops.pytocr_mi355.ctc_greedy(probs)

db_postprocess(Tensor maps f32[N,1,H,W] (ROCm device), Tensor shape_list f64[N,4] (host: src_h, src_w, ratio_h, ratio_w -- the
    reference's `shape_list`, db_postprocess.py:40-47), float thresh, float box_thresh, float unclip_ratio, int max_candidates,
    bool use_padding_resize) -> (Tensor boxes i16[sum K,4,2] (host, images concatenated in order), Tensor counts i32[N] (host))
ctc_greedy(Tensor x f32[T,B,C] (ROCm device; softmax probabilities or logits: arg-max and max are taken over C, for logits the max
    probability is computed from them)) -> (Tensor idx i32[B,T], Tensor prob f32[B,T]) on the device
Errors surface as Python RuntimeError (the C ABI's status + ptocr_last_error); both ops are stream-ordered on the current HIP
stream, db_postprocess synchronises it once for the host result -- like the reference's return value.  There is no CPU kernel:
a CPU tensor raises."""
import numpy as np
import torch

from .modeling import ops
from .postprocess import db_postprocess as _dbp

_lib_def = torch.library.Library("pytocr_mi355", "DEF")
_lib_def.define("db_postprocess(Tensor maps, Tensor shape_list, float thresh, float box_thresh, float unclip_ratio, int max_candidates, "
                "bool use_padding_resize) -> (Tensor, Tensor)")
_lib_def.define("ctc_greedy(Tensor x, bool is_prob=True) -> (Tensor, Tensor)")


def _db_postprocess(maps, shape_list, thresh, box_thresh, unclip_ratio, max_candidates, use_padding_resize):
    if maps.dim() != 4 or maps.shape[1] != 1:
        raise RuntimeError("pytocr_mi355::db_postprocess: maps must be f32[N,1,H,W]")
    if max_candidates != _dbp.MAX_CANDIDATES:
        raise RuntimeError("pytocr_mi355::db_postprocess: max_candidates is %d in the C++ post-process this mirrors "
                           "(db_postprocess.cpp:237); got %d" % (_dbp.MAX_CANDIDATES, max_candidates))
    sl = np.asarray(shape_list.detach().cpu().numpy(), np.float64).reshape(-1, 4)
    src_wh = np.stack([sl[:, 1], sl[:, 0]], 1).astype(np.int32)                 # (src_w, src_h)
    boxes, _ = _dbp.device_boxes(maps[:, 0], src_wh, float(thresh), float(box_thresh), float(unclip_ratio),
                                 use_padding_resize=bool(use_padding_resize))
    counts = torch.tensor([len(b) for b in boxes], dtype=torch.int32)
    flat = np.concatenate(boxes, 0) if len(boxes) and sum(len(b) for b in boxes) else np.zeros((0, 4, 2), np.int16)
    return torch.from_numpy(np.ascontiguousarray(flat, np.int16)), counts


def _ctc_greedy(x, is_prob=True):
    if x.dim() != 3:
        raise RuntimeError("pytocr_mi355::ctc_greedy: x must be f32[T,B,C]")
    T, B, Cn = x.shape
    ld = (Cn + 3) // 4 * 4                                                      # the kernel reads 16-byte pieces: row stride padded,
    rows = torch.zeros((B * T, ld), dtype=torch.float32, device=x.device)       # columns >= C are ignored
    rows[:, :Cn] = x.float().permute(1, 0, 2).reshape(B * T, Cn)               # (B,T) rows like preds.transpose(1,0,2)
    idx, prob = ops.ctc_greedy(rows, Cn, bool(is_prob))
    return idx.reshape(B, T), prob.reshape(B, T)


def _no_cpu(*a, **k):
    raise RuntimeError("pytocr_mi355: this op runs on a ROCm device only; there is no CPU fallback")


_lib_impl = torch.library.Library("pytocr_mi355", "IMPL")
_lib_impl.impl("db_postprocess", _db_postprocess, "CUDA")
_lib_impl.impl("ctc_greedy", _ctc_greedy, "CUDA")
_lib_impl.impl("db_postprocess", _no_cpu, "CPU")
_lib_impl.impl("ctc_greedy", _no_cpu, "CPU")
