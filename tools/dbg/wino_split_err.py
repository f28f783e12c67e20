"""GPU box: error of the DBNet-r18 maps / features with the F(4x4) Winograd kernel's split-operand form (PTOCR_WINO_SPLIT=1) against
the reference-generated golden and the scene golden.  usage: PTOCR_WINO_SPLIT=0|1 wino_split_err.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorchocr_amd.modeling.architectures import build_model
from pytorchocr_amd.utils.synth import synth_images, synth_state_dict, synth_scene_inputs
gd = os.path.join(bench.ROOT, "tests", "golden")
def run(sd, x):
    m = build_model(dict(bench.DET_R18, return_all_feats=True))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        return m(torch.from_numpy(x).cuda())
g = np.load(os.path.join(gd, "det_r18_db_1x3x64x96.npz"))
y = run(synth_state_dict(bench.load_contract("det_r18_db")), synth_images(1, 3, 64, 96, seed=int(g["seed"])))
print("split", os.environ.get("PTOCR_WINO_SPLIT", "0"), "64x96: maps max err %.3g" % np.abs(y["maps"].cpu().numpy() - g["maps"]).max(),
      "c5 rel %.3g" % (np.abs(y["backbone_out"][3].cpu().numpy() - g["c5"]).max() / np.abs(g["c5"]).max()),
      "neck rel %.3g" % (np.abs(y["neck_out"].cpu().numpy() - g["neck"]).max() / np.abs(g["neck"]).max()))
g = np.load(os.path.join(gd, "det_r18_scene_1x3x224x320.npz"))
y = run(bench.det_state_dict("det_r18_db", "r18"), synth_scene_inputs(1, 224, 320, seed=int(g["seed"])))
p = y["maps"].cpu().numpy()
print("   scene 224x320 (gain 14): maps max err %.3g, flips at 0.3: %d of %d" % (np.abs(p - g["maps"]).max(), ((p > 0.3) != (g["maps"] > 0.3)).sum(), p.size))
