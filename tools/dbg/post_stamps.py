#!/usr/bin/env python3
"""GPU box: where the cycles of the two per-border stage kernels go (s_memtime stamps recorded by the kernels themselves,
PTOCR_DBPOST_STAMPS=1).  usage: post_stamps.py [batch]"""
import ctypes as C, os, sys
os.environ["PTOCR_DBPOST_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorchocr_amd import _lib
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_prob_maps

B, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 736, 1280
post = build_post_process(dict(name="DBPostProcess", thresh=0.3, box_thresh=0.5, max_candidates=1000, unclip_ratio=1.7, cpp_speedup=True), {})
maps = torch.from_numpy(synth_prob_maps(min(4, B), H, W, seed=7)).cuda().repeat(max(B // 4, 1), 1, 1)[:B, None].contiguous()
shape_list = np.array([[H, W, 1.0, 1.0]] * B)
for _ in range(3):
    post({"maps": maps}, shape_list)
nrec = B * 1000
st = np.zeros((nrec, 16), np.int64)
lib = _lib.lib()
lib.ptocr_dbpost_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
_lib.check(lib.ptocr_dbpost_debug_stamps(post._ws.handle, st.ctypes.data_as(C.c_void_p), nrec), "stamps")
# quad kernel: records are per wave (blockIdx.y * gridDim.x + blockIdx.x), slots 0..11
q = st[: B * 63][:, :12]
q = q[(q[:, 0] > 0) & (q[:, 11] > 0)]
q = q[q[:, 0] > q[:, 11].max() - 2000000]
names = ["load pts", "hull 1", "calipers 1", "angle etc.", "mini boxes", "offset", "(union) + sort", "hull 2", "calipers 2", "angle etc. 2", "final box"]
d = np.diff(q, axis=1)
print("quad kernel: %d waves with a full record; cycles per phase (median / mean / max)" % len(q))
for i, nm in enumerate(names):
    print("  %-16s %8.0f %8.0f %8.0f" % (nm, np.median(d[:, i]), d[:, i].mean(), d[:, i].max()))
print("  %-16s %8.0f %8.0f %8.0f" % ("total", np.median(q[:, 11] - q[:, 0]), (q[:, 11] - q[:, 0]).mean(), (q[:, 11] - q[:, 0]).max()))
w = st[:, 12:15]
w = w[(w[:, 0] > 0) & (w[:, 2] > 0)]
w = w[w[:, 0] > w[:, 2].max() - 2000000]          # the last call only (records of borders that exist in earlier calls only stay behind)
dw = np.diff(w, axis=1)
print("wave kernel: %d borders; cycles (median / mean / max)" % len(w))
for i, nm in enumerate(["hull candidates", "score"]):
    print("  %-16s %8.0f %8.0f %8.0f" % (nm, np.median(dw[:, i]), dw[:, i].mean(), dw[:, i].max()))
print("  kernel span: quad %.0f cycles, wave %.0f cycles" % (q[:, 11].max() - q[:, 0].min(), w[:, 2].max() - w[:, 0].min()))
t0 = w[:, 0].min()
print("wave kernel timeline (cycles after the first start): starts p50 %.0f p90 %.0f p99 %.0f max %.0f; ends p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
    *np.percentile(w[:, 0] - t0, [50, 90, 99, 100]), *np.percentile(w[:, 2] - t0, [50, 90, 99, 100])))
late = np.argsort(w[:, 2])[-8:]
print("the eight borders that end last: start, hull cycles, score cycles")
for i in late:
    print("   %8d %8d %8d" % (w[i, 0] - t0, w[i, 1] - w[i, 0], w[i, 2] - w[i, 1]))
