"""ad-hoc (not collected): one fuzz seed of test_random_scenes_bit_exact, the neighbourhood and the GPU states of one border"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import test_gpu_dbpost as T
from oracle import dbpost
seed, img, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1000 + seed)
h = int(rng.integers(40, 400)); w = int(rng.integers(40, 700)); n = int(rng.integers(1, 4))
maps = np.stack([T._random_scene(rng, h, w) for _ in range(n)])
src = [[int(rng.integers(20, 2000)), int(rng.integers(20, 2000))] for _ in range(n)]
bt, ratio = float(rng.choice([0.3, 0.5, 0.7])), float(rng.choice([1.5, 1.7, 2.0]))
print("H W n", h, w, n)
T._gpu(maps, src, 0.3, bt, ratio)
tot, res, cands, info = T._debug(img, w)
bm = dbpost.binarize(maps[img], 0.3)
exp_r, dbg_r, ncont = dbpost.boxes_from_bitmap(maps[img], bm, bt, ratio, src[img][0], src[img][1], True)
print("gpu total", tot, "oracle contours", ncont)
c = cands[k]; y, x = c.p // w, c.p % w
print("border", k, "trigger", (y, x), "hole", c.is_hole, "gpu npts", info[k].npts, "oracle npts", dbg_r[k].npts, "bbox gpu", info[k].xmin, info[k].xmax, info[k].ymin, info[k].ymax)
y0, y1, x0, x1 = max(0, y - 6), min(h, y + 7), max(0, x - 10), min(w, x + 12)
print("bitmap rows %d..%d cols %d..%d" % (y0, y1 - 1, x0, x1 - 1))
for yy in range(y0, y1):
    print("%4d " % yy + "".join("#" if bm[yy, xx] else "." for xx in range(x0, x1)))
from pytorchocr_amd import _lib
from pytorchocr_amd.postprocess import db_postprocess as m
st = (C.c_uint32 * 4096)(); nn = C.c_int(0)
_lib.check(_lib.lib().ptocr_dbpost_debug_states(m._ws.handle, img, k, st, 4096, C.byref(nn)))
S = sorted((int(s) >> 11 & 0x7fff, int(s) & 0x7ff, int(s) >> 26 & 7, int(s) >> 29) for s in st[:nn.value])
print("gpu states (y, x, s_out, s_in):", S)
# neighbours in the candidate list around k
for kk in range(max(0, k - 3), min(tot, k + 4)):
    print("  cand", kk, "p", (cands[kk].p // w, cands[kk].p % w), "hole", cands[kk].is_hole, "npts gpu/oracle", info[kk].npts, dbg_r[kk].npts if kk < len(dbg_r) else None)
