"""GPU box: is a detector forward bound by the HOST's launch rate?  Times (a) the host's wall time to ENQUEUE one forward (no sync),
(b) the device time of a forward (events), (c) the same forward replayed from a captured HIP graph (torch.cuda.CUDAGraph).
usage: host_bound_probe.py [r18|mbv3s] [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorchocr_amd.utils.synth import synth_scene_inputs
name = sys.argv[1] if len(sys.argv) > 1 else "mbv3s"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg, contract, _, scene = bench.DET_VARIANTS[name]
model = bench.build_and_sync_weights(cfg, contract, dev, 0, 1, scene=scene)
if dtype == "bf16":
    model.set_compute_dtype("bf16")
x = torch.from_numpy(synth_scene_inputs(32, 736, 1280, seed=2022)).to(dev).contiguous()
with torch.no_grad():
    for _ in range(5):
        out = model(x)
torch.cuda.synchronize()
host, devms = [], []
with torch.no_grad():
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        out = model(x)
        e1.record()
        host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        devms.append(e0.elapsed_time(e1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        out = model(x)
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / 50 * 1e3
print("%s %s: host enqueue of a forward %.3f ms (median), device time of a lone forward %.3f ms, back-to-back loop %.3f ms per forward" % (
    name, dtype, float(np.median(host)), float(np.median(devms)), loop))
ref = out["maps"].clone()
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            model(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        gout = model(x)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay: %.3f ms per forward; maps equal to the eager forward's: %s" % ((time.perf_counter() - t0) / 50 * 1e3, bool(torch.equal(gout["maps"], ref))))
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
# what two timing events per forward cost (bench.py's bf16 line recorded them inside the timed region until round 6)
with torch.no_grad():
    for mode in ("no events", "two timing events per forward", "two events without timing"):
        torch.cuda.synchronize()
        keep = []
        t0 = time.perf_counter()
        for _ in range(100):
            if mode != "no events":
                e0 = torch.cuda.Event(enable_timing=mode.startswith("two timing")); e0.record()
            out = model(x)
            if mode != "no events":
                e1 = torch.cuda.Event(enable_timing=mode.startswith("two timing")); e1.record(); keep.append((e0, e1))
        torch.cuda.synchronize()
        print("%-34s %.3f ms per forward" % (mode, (time.perf_counter() - t0) / 100 * 1e3))
