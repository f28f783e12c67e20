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
_lib.check(lib.ptocr_dbpost_debug_stamps(post._ws.handle, None, nrec), "clear stamps")
post({"maps": maps}, shape_list)
_lib.check(lib.ptocr_dbpost_debug_stamps(post._ws.handle, st.ctypes.data_as(C.c_void_p), nrec), "stamps")
# quad kernel: records are per wave (blockIdx.y * gridDim.x + blockIdx.x), slots 0..11
q = st[: B * 63][:, :12]
q = q[(q[:, 0] > 0) & (q[:, 11] > 0)]
if len(q):
    q = q[q[:, 0] > q[:, 11].max() - 2000000]
    names = ["load pts", "hull 1", "calipers 1", "angle etc.", "mini boxes", "offset", "(union) + sort", "hull 2", "calipers 2", "angle etc. 2", "final box"]
    d = np.diff(q, axis=1)
    print("quad kernel: %d waves with a full record; cycles per phase (median / mean / max)" % len(q))
    for i, nm in enumerate(names):
        print("  %-16s %8.0f %8.0f %8.0f" % (nm, np.median(d[:, i]), d[:, i].mean(), d[:, i].max()))
    print("  %-16s %8.0f %8.0f %8.0f" % ("total", np.median(q[:, 11] - q[:, 0]), (q[:, 11] - q[:, 0]).mean(), (q[:, 11] - q[:, 0]).max()))
hw = st[:, 12:14]
hw = hw[(hw[:, 0] > 0) & (hw[:, 1] > 0)]
sw = st[:, 14:16]
sw = sw[(sw[:, 0] > 0) & (sw[:, 1] > 0)]
both = [v for v in (hw, sw) if len(v)]
t0 = min(v[:, 0].min() for v in both)
print("wave kernel: %d hull items, %d scored borders; 10-ns ticks of the shared clock (median / mean / max)" % (len(hw), len(sw)))
for nm, v in (("hull candidates", hw), ("score (band 0 start -> border scored)", sw)):
    if len(v):
        dv = v[:, 1] - v[:, 0]
        print("  %-40s %8.0f %8.0f %8.0f" % (nm, np.median(dv), dv.mean(), dv.max()))
print("  kernel span: wave %.0f ticks" % (max(v[:, 1].max() for v in both) - t0))
for nm, v in (("hull starts", hw[:, 0]), ("hull ends", hw[:, 1]), ("score band-0 starts", sw[:, 0]), ("score ends", sw[:, 1])):
    if len(v):
        print("  %-22s p1 %8.0f p50 %8.0f p90 %8.0f p99 %8.0f max %8.0f" % ((nm,) + tuple(np.percentile(v - t0, [1, 50, 90, 99, 100]))))
if len(hw):
    # starts per image (hull records are indexed img * 1000 + k): does the dispatcher walk the images in order?
    h_all = st[:, 12]
    for img in (0, B // 2, B - 1):
        v = h_all[img * 1000:(img + 1) * 1000]
        v = v[v > 0]
        if len(v):
            print("  image %2d: %4d hull items start at %8.0f .. %8.0f" % (img, len(v), v.min() - t0, v.max() - t0))

# the ten borders scored last / longest, with their boxes
class Info(C.Structure):
    _fields_ = [("npts", C.c_int), ("off", C.c_int), ("xmin", C.c_short), ("xmax", C.c_short), ("ymin", C.c_short), ("ymax", C.c_short)]
sc = st[:, 14:16]
dur = np.where((sc[:, 0] > 0) & (sc[:, 1] > 0), sc[:, 1] - sc[:, 0], 0)
hd = np.where((st[:, 12] > 0) & (st[:, 13] > 0), st[:, 13] - st[:, 12], 0)
for title, order in (("longest score", np.argsort(dur)[-10:]), ("longest hull", np.argsort(hd)[-6:])):
    print(title + ": image, border, ticks, start, bbox w x h, contour points, state offset")
    for r in order[::-1]:
        img, k = int(r) // 1000, int(r) % 1000
        tot = C.c_int(0); res = (C.c_char * (68 * 1000))(); cands = (C.c_char * 8000)(); info = (Info * 1000)()
        _lib.check(lib.ptocr_dbpost_debug_results(post._ws.handle, img, C.byref(tot), res, cands, info), "dbg")
        i = info[k]
        print("   %2d %4d %6d %6d   %4d x %3d  npts %5d off %7d" % (img, k, (dur if title.startswith("longest s") else hd)[r], (sc[r, 0] if title.startswith("longest s") else st[r, 12]) - t0, i.xmax - i.xmin + 1, i.ymax - i.ymin + 1, i.npts, i.off))

# per-item phases of the score role (bands 0 and 1 of the borders of images >= 4)
for band in (0, 1):
    it = st[4000:, band * 6:band * 6 + 6]
    it = it[(it > 0).all(axis=1)]
    if len(it):
        dd = np.diff(it, axis=1)
        print("score items, band %d: %d records; ticks median / mean / p99 / max" % (band, len(it)))
        for i, nm in enumerate(["zero + state scan", "prefix pass", "masked sum", "(raster)", "block reduce"]):
            print("  %-20s %7.0f %7.0f %7.0f %7.0f" % (nm, np.median(dd[:, i]), dd[:, i].mean(), np.percentile(dd[:, i], 99), dd[:, i].max()))
