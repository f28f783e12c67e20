import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import dbpost
from pytorchocr_amd import _lib
from pytorchocr_amd.postprocess import db_postprocess as m
from pytorchocr_amd.utils.synth import synth_prob_maps
class Res(C.Structure):
    _fields_ = [("status", C.c_int), ("box", C.c_int * 8), ("score", C.c_float), ("rect", C.c_float * 5), ("npix", C.c_int), ("distance", C.c_float)]
class Cand(C.Structure):
    _fields_ = [("p", C.c_int), ("is_hole", C.c_int)]
class Info(C.Structure):
    _fields_ = [("npts", C.c_int), ("off", C.c_int), ("xmin", C.c_short), ("xmax", C.c_short), ("ymin", C.c_short), ("ymax", C.c_short)]
maps = synth_prob_maps(3, 96, 160, seed=5)
got, flags = m.device_boxes(torch.from_numpy(maps).cuda(), [[160, 96]] * 3, 0.3, 0.5, 1.7)
img = 0
tot = C.c_int(0); res = (Res * 1000)(); cands = (Cand * 1000)(); info = (Info * 1000)()
_lib.check(_lib.lib().ptocr_dbpost_debug_results(m._ws.handle, img, C.byref(tot), res, cands, info))
bm = dbpost.binarize(maps[img], 0.3)
exp, dbg, ncont = dbpost.boxes_from_bitmap(maps[img], bm, 0.5, 1.7, 160, 96, True)
print("tot", tot.value, ncont)
for k in range(min(tot.value, 40)):
    d = dbg[k]
    print(k, "p", cands[k].p, "hole", cands[k].is_hole, "npts", info[k].npts, "oracle", d.npts, "bbox", info[k].xmin, info[k].xmax, info[k].ymin, info[k].ymax, "st", res[k].status, d.status)
# states of border 1 vs the CPU enumeration
from scipy import ndimage
k = 1
st = (C.c_uint32 * 100000)(); n = C.c_int(0)
_lib.check(_lib.lib().ptocr_dbpost_debug_states(m._ws.handle, img, k, st, 100000, C.byref(n)))
gpu = sorted((s & 0x7ff, (s >> 11) & 0x7fff, (s >> 29) & 7, (s >> 26) & 7) for s in st[:n.value])
H, W = bm.shape
fg, _ = ndimage.label(bm, structure=np.ones((3, 3)))
bgl, _ = ndimage.label(np.pad(1 - bm, 1, constant_values=1)); bg = bgl[1:-1, 1:-1]; FR = bgl[0, 0]
DX = [1, 1, 0, -1, -1, -1, 0, 1]; DY = [0, -1, -1, -1, 0, 1, 1, 1]
pix = lambda x, y: bm[y, x] if 0 <= x < W and 0 <= y < H else 0
bl = lambda x, y: bg[y, x] if 0 <= x < W and 0 <= y < H else FR
p = cands[k].p; ty, tx = divmod(p, W)
F = fg[ty, tx]; S = bl(tx - 1, ty)
cpu = []
ys, xs = np.nonzero(fg == F)
for x, y in zip(xs, ys):
    nb = [pix(x + DX[d], y + DY[d]) for d in range(8)]
    for s_in in range(8):
        if nb[s_in] and not nb[(s_in + 1) % 8]:
            L = 0
            while not nb[(s_in + 1 + L) % 8]: L += 1
            s_out = (s_in + 1 + L) % 8
            if not (L >= 2 or (L == 1 and (s_in + 1) % 2 == 0)): continue
            g4 = (s_in + 2) % 8 if (s_in + 1) % 2 else (s_in + 1) % 8
            if bl(x + DX[g4], y + DY[g4]) != S: continue
            cpu.append((int(x), int(y), s_in, s_out))
cpu.sort()
print("gpu states", len(gpu), "cpu", len(cpu))
sg, sc = set(gpu), set(cpu)
print("missing on gpu", sorted(sc - sg)[:40])
print("extra on gpu", sorted(sg - sc)[:40])
lab = np.zeros((H, W), np.int32); wlb = np.zeros((H, (W + 31) // 32), np.int32)
_lib.check(_lib.lib().ptocr_dbpost_debug_labels(m._ws.handle, img, H, W, lab.ctypes.data_as(C.c_void_p), wlb.ctypes.data_as(C.c_void_p)))
print("row 50 bits", "".join(str(int(v)) for v in bm[50, :40]))
print("row 49 bits", "".join(str(int(v)) for v in bm[49, :40]))
print("lab[50,7]", lab[50, 7], "lab[49,17]", lab[49, 17], "lab[44,141]", lab[44, 141], "wl row49", wlb[49], "wl row50", wlb[50], "wl row 44", wlb[44])
print("lab[49,0]", lab[49, 0], "lab[50,0]", lab[50, 0])
