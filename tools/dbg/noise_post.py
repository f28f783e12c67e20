import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import json, numpy as np, torch
from pytorchocr_amd.modeling.architectures import build_model
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_images, synth_state_dict
import bench
cfg, name, _ = bench.DET_VARIANTS["mbv3s"]
m = build_model(dict(cfg)); sd = synth_state_dict(bench.load_contract(name))
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval().set_compute_dtype("bf16")
x = torch.from_numpy(synth_images(4, 3, 736, 1280, seed=2022)).cuda().repeat(8, 1, 1, 1)
with torch.no_grad(): maps = m(x)["maps"]
mp = maps[0, 0].cpu().numpy(); print("frac > 0.3:", (mp > 0.3).mean(), "min/max", mp.min(), mp.max())
post = build_post_process(bench.DET_POST, {})
sl = np.array([[736, 1280, 1, 1]] * 32)
post.device_ms_log = []
for _ in range(5): r = post({"maps": maps}, sl)
print("device ms", post.device_ms_log, "boxes", sum(len(i["points"]) for i in r))
