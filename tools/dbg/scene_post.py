"""GPU box: device time of the post-process on the scene checkpoint's own maps (32 x 736x1280, bench.py's default input) next to the
text-like stress maps.  usage: scene_post.py [r18|mbv3s]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_prob_maps, synth_scene_inputs
which = sys.argv[1] if len(sys.argv) > 1 else "r18"
cfg, contract, _, scene = bench.DET_VARIANTS[which]
model = bench.build_and_sync_weights(cfg, contract, torch.device("cuda:0"), 0, 1, scene=scene)
x = torch.from_numpy(synth_scene_inputs(4, 736, 1280, seed=2022)).cuda().repeat(8, 1, 1, 1).contiguous()
with torch.no_grad():
    maps = model(x)["maps"]
post = build_post_process(bench.DET_POST, {})
sl = np.array([[736, 1280, 1, 1]] * 32)
text = torch.from_numpy(synth_prob_maps(4, 736, 1280, seed=7)).cuda().repeat(8, 1, 1)[:, None].contiguous()
only = os.environ.get("ONLY")
for name, mp in (("model", maps), ("text-like", text)):
    if only and only != name:
        continue
    post.device_ms_log = []
    for _ in range(8):
        r = post({"maps": mp}, sl)
    print(name, "device ms", [round(v, 3) for v in post.device_ms_log], "boxes/img", sum(len(i["points"]) for i in r) / 32.0)
    if os.environ.get("TOTALS"):
        import ctypes as C
        from pytorchocr_amd import _lib
        tots = []
        for i in range(0, 32, 4):
            tot = C.c_int(0); res = (C.c_char * (68 * 1000))(); cands = (C.c_char * 8000)(); info = (C.c_char * 16000)()
            _lib.check(_lib.lib().ptocr_dbpost_debug_results(post._ws.handle, i, C.byref(tot), res, cands, info), "dbg")
            st = np.frombuffer(res, np.int32).reshape(1000, 17)[:, 0][:min(tot.value, 1000)]
            tots.append((tot.value, np.bincount(st, minlength=8)[:8].tolist()))
        print(name, "borders per image and their statuses (OK, npts<=2, ssid<3, score, unclip, ssid2, none, defer):", tots)
