"""Per-launch convolution times of one CRNN batch: python tools/dbg/crnn_layers.py [batch] [width]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pytorchocr_amd.modeling import ops
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 320
dev = torch.device("cuda:0")
model = bench.build_and_sync_weights(bench.crnn_cfg(), "rec_vgg_bilstm_ctc", dev, 0, 1)
x = torch.randn(B, 1, 32, W, device=dev)
with torch.no_grad():
    for _ in range(2):
        model.forward_greedy(x)
    ops.PROFILE_LABELS, ops.PROFILE = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.forward_greedy(x)
    e1.record()
    torch.cuda.synchronize()
tot = 0.0
for lab, (a, b) in zip(ops.PROFILE_LABELS, ops.PROFILE):
    t = a.elapsed_time(b); tot += t
    print("%-60s %.3f ms" % (lab, t))
print("convs %.3f ms, forward %.3f ms" % (tot, e0.elapsed_time(e1)))
