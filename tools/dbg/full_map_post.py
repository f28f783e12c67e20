"""GPU box: device time of the post-process on all-foreground maps (what a random-weight detector produces: one giant component per
image, the full-size pass) and on text-like maps.  usage: full_map_post.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_prob_maps
post = build_post_process(bench.DET_POST, {})
sl = np.array([[736, 1280, 1, 1]] * 32)
full = torch.full((32, 1, 736, 1280), 0.48, device="cuda")
text = torch.from_numpy(synth_prob_maps(4, 736, 1280, seed=7)).cuda().repeat(8, 1, 1)[:, None].contiguous()
for name, maps in (("all-foreground", full), ("text-like", text), ("all-foreground", full)):
    post.device_ms_log = []
    for _ in range(6):
        r = post({"maps": maps}, sl)
    print(name, "device ms", [round(v, 3) for v in post.device_ms_log], "boxes", sum(len(i["points"]) for i in r))
