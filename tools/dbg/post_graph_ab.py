"""GPU box: device and wall time of the post-process call on 32 text-like maps with PTOCR_DBPOST_PARTS from the env (the hipGraph replay
this script also measured -- PTOCR_DBPOST_GRAPH, DESIGN 3.3 -- was not kept in the library)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_prob_maps
post = build_post_process(bench.DET_POST, {})
sl = np.array([[736, 1280, 1, 1]] * 32)
text = torch.from_numpy(synth_prob_maps(4, 736, 1280, seed=7)).cuda().repeat(8, 1, 1)[:, None].contiguous()
post.device_ms_log = []
torch.cuda.synchronize()
with torch.cuda.stream(torch.cuda.Stream()):                       # not the null stream: a capture needs a real one
    for _ in range(12):
        r = post({"maps": text}, sl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r = post({"maps": text}, sl)
    wall = (time.perf_counter() - t0) / 20 * 1e3
print("graph", os.environ.get("PTOCR_DBPOST_GRAPH", "0"), "parts", os.environ.get("PTOCR_DBPOST_PARTS", "1"),
      "device ms median %.3f" % float(np.median(post.device_ms_log[12:])), "wall ms %.3f" % wall, "boxes/img", sum(len(i["points"]) for i in r) / 32.0)
