#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE itself (runs only where /root/reference exists).

What it does (nothing from the reference is copied; only inputs-by-seed and outputs are stored):
  * stubs `torchvision.models.utils` (absent here; only imported, never called with pretrained=False),
    imports the reference model code on CPU, loads the portable synthetic state_dict
    (pytorchocr_amd/utils/synth.py), runs small inputs, stores outputs;
  * imports pytocr/postprocess/rec_postprocess.py BY FILE PATH (the package __init__ needs cv2) and
    records CTCLabelDecode results on hand-made index/prob cases;
  * records the state_dict key/shape/dtype contract of each model.

Usage: python tools/gen_golden.py   (writes every fixture under tests/golden/;
       --cls-only / --mbv3s-scene-only / --widen-only / --labels-only / --shapes-only / --clipper-only regenerate one group)
"""
import importlib.util
import json
import os
import sys
import types

import copy

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PTOCR_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from pytorchocr_amd.utils.synth import synth_state_dict, synth_images, synth_text_lines, uniform  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")



def write_contract(path, contract):
    """state_dict_contract.json through ONE fixed formatter (a model per block, a parameter per line, insertion order), so that a
    regeneration is byte-identical whichever step of this script touched the file last"""
    with open(path, "w") as f:
        f.write("{\n")
        for mi, (model, keys) in enumerate(contract.items()):
            f.write(" %s: {\n" % json.dumps(model))
            items = list(keys.items())
            for ki, (k, (shape, dtype)) in enumerate(items):
                f.write("  %s: [%s, %s]%s\n" % (json.dumps(k), json.dumps([int(v) for v in shape]), json.dumps(dtype), "," if ki + 1 < len(items) else ""))
            f.write(" }%s\n" % ("," if mi + 1 < len(contract) else ""))
        f.write("}\n")


def _import_reference_models():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvu = types.ModuleType("torchvision.models.utils")

    def _no(*a, **k):
        raise RuntimeError("network fetch is not available")

    tvu.load_state_dict_from_url = _no
    tv.models = tvm
    tvm.utils = tvu
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.utils": tvu})
    # The repo root holds a REGULAR package named `pytocr` (the alias of pytorchocr_amd); the reference's `pytocr` has no
    # __init__.py (a namespace package), and a regular package anywhere on sys.path beats a namespace portion.  So the repo
    # root (and the current directory, when it is the root) leave sys.path while the reference is imported.
    saved = list(sys.path)
    sys.path[:] = [REF] + [q for q in sys.path if os.path.abspath(q or os.getcwd()) != ROOT]
    for name in [k for k in sys.modules if k == "pytocr" or k.startswith("pytocr.")]:
        del sys.modules[name]
    from pytocr.modeling.architectures import build_model
    assert os.path.abspath(sys.modules["pytocr.modeling.architectures"].__file__).startswith(os.path.abspath(REF)), "not the reference"
    sys.path[:] = [REF] + saved
    return build_model


DET_R18 = dict(model_type="det", algorithm="DB", Transform=None,
               Backbone=dict(name="ResNet", layers=18, pretrained=False),
               Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False,
                         attention_type="scale_channel_spatial"),
               Head=dict(name="DBHead", k=50))
DETPP_R18 = dict(model_type="det", algorithm="DB", Transform=None,
                 Backbone=dict(name="ResNet", layers=18, pretrained=False),
                 Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=True,
                           attention_type="scale_channel_spatial"),
                 Head=dict(name="DBHead", k=50))
DET_MBV3S = dict(model_type="det", algorithm="DB", Transform=None,
                 Backbone=dict(name="MobileNetV3", model_name="small", scale=1.0, pretrained=False),
                 Neck=dict(name="FPN", out_channels=96, mode="DB", use_asf=False),
                 Head=dict(name="DBHead", k=50))


def crnn_cfg(nclass):
    return dict(model_type="rec", algorithm="CRNN", in_channels=1, Transform=None,
                Backbone=dict(name="VGG", model_name="v1", scale=1.0, pretrained=False, ckpt_path=None),
                Neck=dict(name="SequenceEncoder", encoder_type="rnn", hidden_size=256),
                Head=dict(name="CTCHead", out_channels=nclass))


def build_with_synth(build_model, cfg, seed=2022):
    m = build_model(cfg).eval()
    shapes = {k: (tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()}
    w = synth_state_dict(shapes, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return m, shapes


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    build_model = _import_reference_models()
    contract = {}

    # ---------------- DBNet r18: maps + intermediates on a small input, maps on a ragged batch
    m, shapes = build_with_synth(build_model, DET_R18)
    contract["det_r18_db"] = {k: [list(s), d] for k, (s, d) in shapes.items()}
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=11))
    m.return_all_feats = True
    with torch.no_grad():
        y = m(x)
    np.savez_compressed(
        os.path.join(GOLD, "det_r18_db_1x3x64x96.npz"),
        seed=np.int64(11), maps=y["maps"].numpy(),
        c2=y["backbone_out"][0].numpy(), c3=y["backbone_out"][1].numpy(),
        c4=y["backbone_out"][2].numpy(), c5=y["backbone_out"][3].numpy(),
        neck=y["neck_out"].numpy())
    m.return_all_feats = False
    x = torch.from_numpy(synth_images(2, 3, 96, 160, seed=12))
    with torch.no_grad():
        y = m(x)
    np.savez_compressed(os.path.join(GOLD, "det_r18_db_2x3x96x160.npz"), seed=np.int64(12), maps=y["maps"].numpy())

    # ---------------- DB++ r18 and DB mbv3-small: maps only (later rows of SURVEY 8a: M3, M5)
    for name, cfg, seed in (("detpp_r18_db", DETPP_R18, 13), ("det_mbv3s_db", DET_MBV3S, 14)):
        m, shapes = build_with_synth(build_model, cfg)
        contract[name] = {k: [list(s), d] for k, (s, d) in shapes.items()}
        x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=seed))
        with torch.no_grad():
            y = m(x)
        np.savez_compressed(os.path.join(GOLD, f"{name}_1x3x64x96.npz"), seed=np.int64(seed), maps=y["maps"].numpy())

    # ---------------- CRNN (6624 classes as with char_dict_6623.txt; infer_rec.py:62-63)
    nclass = 6624
    m, shapes = build_with_synth(build_model, crnn_cfg(nclass))
    contract["rec_vgg_bilstm_ctc"] = {k: [list(s), d] for k, (s, d) in shapes.items()}
    x = torch.from_numpy(synth_text_lines(3, 32, 320, seed=15))
    with torch.no_grad():
        p = m(x)                       # softmax [T,B,C]
    pn = p.numpy()
    idx = pn.transpose(1, 0, 2).argmax(axis=2)
    prob = pn.transpose(1, 0, 2).max(axis=2)
    cols = np.arange(0, nclass, 97)
    np.savez_compressed(os.path.join(GOLD, "crnn_3x1x32x320.npz"), seed=np.int64(15),
                        idx=idx.astype(np.int32), prob=prob, cols=cols.astype(np.int32),
                        probs_cols=pn[:, :, cols], shape=np.array(pn.shape, np.int64))

    write_contract(os.path.join(GOLD, "state_dict_contract.json"), contract)

    # ---------------- CTCLabelDecode known answers from the reference class itself
    spec = importlib.util.spec_from_file_location("ref_rec_postprocess",
                                                  os.path.join(REF, "pytocr/postprocess/rec_postprocess.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dict_path = os.path.join(REF, "pytocr/utils/char_dict_6623.txt")
    dec = mod.CTCLabelDecode(character_dict_path=dict_path, use_space_char=False)
    dec36 = mod.CTCLabelDecode(character_dict_path=None, use_space_char=False)
    cases = []
    rng_cases = [
        [1, 1, 0, 1, 2, 2], [0, 0, 0, 0], [5], [0], [3, 3, 3], [3, 0, 3], [0, 7, 7, 0, 7, 8, 8, 0],
        [36, 1, 36, 36, 2], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10],
    ]
    for k, seq in enumerate(rng_cases):
        T = len(seq)
        C = 37
        pr = uniform((T, 1, C), 1000 + k, 0.0, 0.5)
        for t, c in enumerate(seq):
            pr[t, 0, c] = 0.6 + 0.01 * t
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = dec36(torch.from_numpy(pr))
        txt, conf = res[0]
        cases.append({"dict": "default36", "seed": 1000 + k, "T": T, "C": C, "seq": seq, "text": txt,
                      "conf": None if np.isnan(conf) else float(conf)})
    # big-dictionary case: indices map through char_dict_6623.txt
    seq = [1, 1, 0, 6623, 100, 100, 0, 100, 4321]
    T, C = len(seq), 6624
    pr = np.zeros((T, 2, C), np.float32)
    for t, c in enumerate(seq):
        pr[t, 0, c] = 0.9
        pr[t, 1, (c * 7 + 1) % C] = 0.8
    res = dec(torch.from_numpy(pr))
    cases.append({"dict": "char_dict_6623", "T": T, "C": C, "seq": seq,
                  "text": [r[0] for r in res], "conf": [float(r[1]) for r in res],
                  "nclass": len(dec.character)})
    with open(os.path.join(GOLD, "ctc_decode.json"), "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=False, indent=0)
    print("golden fixtures written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print("  %-40s %8d B" % (fn, os.path.getsize(os.path.join(GOLD, fn))))




CLS_MBV3S = dict(model_type="cls", algorithm="CLS", Transform=None,
                 Backbone=dict(name="MobileNetV3", model_name="small", width_mult=0.35, use_se=True, pretrained=False, ckpt_path=None),
                 Neck=None, Head=dict(name="ClsHead", class_dim=2))


def gen_cls_vectors():
    """direction classifier (configs/cls/cls_mbv3small.yml): state_dict contract, softmax and pooled backbone features of the
    reference model on a seeded batch; ClsPostProcess / ClsMetric known answers from the reference classes"""
    torch.manual_seed(0)
    build_model = _import_reference_models()
    m, shapes = build_with_synth(build_model, copy.deepcopy(CLS_MBV3S))
    path = os.path.join(GOLD, "state_dict_contract.json")
    contract = json.load(open(path))
    contract["cls_mbv3s"] = {k: [list(sh), d] for k, (sh, d) in shapes.items()}
    write_contract(path, contract)
    fc_scale = 0.25                                 # synth_state_dict's fc (made for the CTC head: bias[0] = 13) saturates a 2-class
    with torch.no_grad():                           # softmax; with the bias zeroed and the weights scaled it discriminates
        m.head.fc.weight.mul_(fc_scale)
        m.head.fc.bias.zero_()
    x = torch.from_numpy(synth_images(4, 3, 48, 192, seed=16))
    m.return_all_feats = True
    with torch.no_grad():
        y = m(x)
    def by_path(name, rel):                         # the packages' __init__ files need cv2 / shapely
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ClsPostProcess = by_path("ref_cls_postprocess", "pytocr/postprocess/cls_postprocess.py").ClsPostProcess
    ClsMetric = by_path("ref_cls_metric", "pytocr/metrics/cls_metric.py").ClsMetric
    probs = y["head_out"].numpy()
    np.savez_compressed(os.path.join(GOLD, "cls_mbv3s_4x3x48x192.npz"), seed=np.int64(16), probs=probs, fc_scale=np.float32(fc_scale),
                        backbone_out=y["backbone_out"].numpy())
    pp = ClsPostProcess(label_list=["0", "180"])
    table = np.array([[0.9, 0.1], [0.2, 0.8], [0.5, 0.5], [0.49, 0.51]], np.float32)
    dec, lab = pp(torch.from_numpy(table), label=[0, 1, 1, 0])
    met = ClsMetric()
    batch = met((dec, lab))
    with open(os.path.join(GOLD, "cls_post.json"), "w") as f:
        json.dump({"table": table.tolist(), "label": [0, 1, 1, 0], "decoded": [[t, float(p)] for t, p in dec],
                   "label_out": [[t, float(p)] for t, p in lab], "batch_metric": batch, "final_metric": met.get_metric(),
                   "model_decoded": [[t, float(p)] for t, p in pp(torch.from_numpy(probs))]}, f, indent=0)
    print("cls vectors written; probs =", probs.tolist())


SCENE_CASES = {   # name: (config, read-out fixture, golden fixture, images in the fit, seed of the golden input, sub-pixel read-out)
    "mbv3s": ("DET_MBV3S", "mbv3s_scene_readout.npz", "det_mbv3s_scene_1x3x224x320.npz", 3, 21, False),
    "r18": ("DET_R18", "r18_scene_readout.npz", "det_r18_scene_1x3x224x320.npz", 6, 22, True),
    "detpp": ("DETPP_R18", "detpp_scene_readout.npz", "detpp_r18_scene_1x3x224x320.npz", 6, 23, True),
}


def gen_scene_vectors(which="mbv3s"):
    """Scene checkpoints (utils/synth.py: synth_scene_state_dict): a detector whose maps are text-like and cross both post-process
    thresholds while every backbone / neck layer keeps its random weights.  The read-out is fitted HERE on the reference model's own
    neck features (ridge regression, float64), stored with the reference's maps / logits / features for one scene image.
    mbv3s: configs[3] parity evidence of the bf16 path; r18 / detpp: the checkpoints bench.py times (configs[1], configs[4]) and
    whole-network parity on maps that produce boxes."""
    import torch.nn.functional as F
    from pytorchocr_amd.utils.synth import synth_scene_state_dict, synth_prob_maps, synth_scene_inputs
    cfg_name, readout_file, golden_file, fn, seed, sub = SCENE_CASES[which]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    build_model = _import_reference_models()
    m, shapes = build_with_synth(build_model, copy.deepcopy(globals()[cfg_name]))
    m.return_all_feats = True
    fh, fw, fseed = 352, 480, 100
    with torch.no_grad():
        fuse = m(torch.from_numpy(synth_scene_inputs(fn, fh, fw, seed=fseed)))["neck_out"].double()
    X = F.unfold(fuse, 3, padding=1).permute(0, 2, 1).reshape(-1, 9 * fuse.shape[1])
    tmap = torch.from_numpy(synth_prob_maps(fn, fh, fw, seed=fseed)).double()[:, None]
    if sub:     # one target per full-resolution pixel of the 4x4 block (column dy * 4 + dx): the 64-channel heads have room for 16 estimates
        t = F.pixel_unshuffle(tmap, 4).permute(0, 2, 3, 1).reshape(-1, 16)
    else:       # the block's mean
        t = F.avg_pool2d(tmap, 4).reshape(-1)
    A = torch.cat([X, torch.ones(len(X), 1, dtype=torch.double)], 1)
    lam = (1e-4 if sub else 1e-3) * len(X) * float(X.var())
    w = torch.linalg.solve(A.T @ A + lam * torch.eye(A.shape[1], dtype=torch.double), A.T @ t)
    r2 = 1.0 - float(((A @ w - t) ** 2).sum() / ((t - t.mean()) ** 2).sum())
    del X, A
    gain, level = 14.0, 0.45
    readout = w.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, readout_file), readout=readout, gain=np.float32(gain), level=np.float32(level),
                        fit=np.array([fn, fh, fw, fseed], np.int64), r2=np.float64(r2))
    sd = synth_scene_state_dict(shapes, readout, gain, level)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    h, w_ = 224, 320
    x = torch.from_numpy(synth_scene_inputs(1, h, w_, seed=seed))
    logits = {}
    hook = m.head.binarize[6].register_forward_hook(lambda mod, i, o: logits.__setitem__("z", o.detach().clone()))
    with torch.no_grad():
        y = m(x)
    hook.remove()
    np.savez_compressed(os.path.join(GOLD, golden_file), seed=np.int64(seed), maps=y["maps"].numpy(),
                        logits=logits["z"].numpy(), neck_sub=y["neck_out"].numpy()[:, :, ::4, ::4],
                        c2_sub=y["backbone_out"][0].numpy()[:, :, ::4, ::4], c5=y["backbone_out"][3].numpy())
    p = y["maps"].numpy()
    print("%s scene: fit R2 %.4f; map range %.3f..%.3f, above 0.3: %.1f %%, above 0.5: %.1f %%" %
          (which, r2, p.min(), p.max(), 100 * (p > 0.3).mean(), 100 * (p > 0.5).mean()))


def gen_mbv3s_scene_vectors():
    gen_scene_vectors("mbv3s")


def gen_widen_vectors():
    """reference options next to the headline configs: ASF attention types scale_spatial / scale_channel (necks/asf.py:9-29,78-107),
    the Bottleneck ResNet-50 backbone (backbones/det_resnet.py:85-140; the README's best DB row) and the three-conv 3x3 stem
    (mode_3x3, :196-206): state_dict contracts + maps (+ backbone features for r50) of the reference on one seeded 64x96 input"""
    torch.manual_seed(0)
    build_model = _import_reference_models()
    path = os.path.join(GOLD, "state_dict_contract.json")
    contract = json.load(open(path))
    cases = (("detpp_r18_db_spatial", dict(DETPP_R18, Neck=dict(DETPP_R18["Neck"], attention_type="scale_spatial")), 31, False),
             ("detpp_r18_db_channel", dict(DETPP_R18, Neck=dict(DETPP_R18["Neck"], attention_type="scale_channel")), 32, False),
             ("det_r50_db", dict(DET_R18, Backbone=dict(name="ResNet", layers=50, pretrained=False)), 33, True),
             ("det_r18_db_3x3stem", dict(DET_R18, Backbone=dict(name="ResNet", layers=18, mode_3x3=True, pretrained=False)), 34, False),
             # the stock configs/det/det_mbv3_db.yml:24-27 backbone: MobileNetV3 LARGE x1.0 (round 5: fp32 and bf16 paths)
             ("det_mbv3l_db", dict(DET_MBV3S, Backbone=dict(name="MobileNetV3", model_name="large", width_mult=1.0, use_se=True, pretrained=False)), 37, True))
    for name, cfg, seed, feats in cases:
        m, shapes = build_with_synth(build_model, copy.deepcopy(cfg))
        contract[name] = {k: [list(sh), d] for k, (sh, d) in shapes.items()}
        x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=seed))
        m.return_all_feats = feats
        with torch.no_grad():
            y = m(x)
        extra = {}
        if feats:
            extra = {"c%d" % (i + 2): f.numpy() for i, f in enumerate(y["backbone_out"])}
        np.savez_compressed(os.path.join(GOLD, "%s_1x3x64x96.npz" % name), seed=np.int64(seed), maps=y["maps"].numpy(), **extra)
        p = y["maps"].numpy()
        print("%s: maps %.4f..%.4f" % (name, p.min(), p.max()))
    # CRNN with the depthwise-separable VGG v2 stack (rec_vgg.py:37-44, 62-76), both widths; 37 classes keep the fixture small
    for name, scale, seed in (("rec_vgg2_bilstm_ctc", 1.0, 35), ("rec_vgg2_half_bilstm_ctc", 0.5, 36)):
        cfg = crnn_cfg(37)
        cfg["Backbone"] = dict(cfg["Backbone"], model_name="v2", scale=scale)
        m, shapes = build_with_synth(build_model, cfg)
        contract[name] = {k: [list(sh), d] for k, (sh, d) in shapes.items()}
        x = torch.from_numpy(synth_text_lines(2, 32, 160, seed=seed))
        feats = {}
        hook = m.backbone.register_forward_hook(lambda mod, i, o: feats.__setitem__("b", o.detach().clone()))
        with torch.no_grad():
            pr = m(x)                          # softmax [T,B,C]
        hook.remove()
        np.savez_compressed(os.path.join(GOLD, "%s_2x1x32x160.npz" % name), seed=np.int64(seed), probs=pr.numpy(), backbone=feats["b"].numpy())
        print("%s: T %d, backbone %s" % (name, pr.shape[0], tuple(feats["b"].shape)))
    write_contract(path, contract)


def gen_clipper_vectors():
    """Unclip golden vectors from the reference's vendored Clipper (oracle/_ref): input int path, delta,
    full solution.  The float mini-boxes come from seeded rotated rectangles (reference UnClip,
    db_postprocess.cpp:34-56: distance from GetContourArea, vertices truncated to int)."""
    sys.path.insert(0, ROOT)
    from oracle import dbpost
    dbpost.build(ref=True)
    assert dbpost.ref_lib() is not None
    vecs = []
    fixed = [([[10, 10], [110, 10], [110, 40], [10, 40]], 19.615385),
             ([[20, 30], [90, 12], [98, 41], [27, 58]], 17.780228)]
    for path, delta in fixed:
        sol = dbpost.clipper_ref_offset(path, delta)
        vecs.append({"path": path, "delta": delta, "solution": [s.tolist() for s in sol]})
    u = uniform((60, 5), 777, 0.0, 1.0)
    for r in u:
        cx, cy = 50 + r[0] * 1100, 50 + r[1] * 600
        w, h = 3 + r[2] * 300, 2 + r[3] * 50
        th = r[4] * np.pi
        c, s = np.cos(th), np.sin(th)
        pts = (np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ np.array([[c, s], [-s, c]])
               + [cx, cy]).astype(np.float32)
        area = abs(0.5 * sum(pts[i, 0] * pts[(i + 1) % 4, 1] - pts[i, 1] * pts[(i + 1) % 4, 0] for i in range(4)))
        per = sum(np.hypot(*(pts[i] - pts[(i + 1) % 4])) for i in range(4))
        delta = float(np.float32(area * 1.7 / per))
        path = pts.astype(np.int32).tolist()
        sol = dbpost.clipper_ref_offset(path, delta)
        vecs.append({"path": path, "delta": delta, "solution": [s.tolist() for s in sol]})
    with open(os.path.join(GOLD, "clipper_unclip.json"), "w") as f:
        json.dump(vecs, f)
    print("clipper vectors:", len(vecs))


def gen_label_vectors():
    """CTCLabelEncode / ClsLabelEncode results recorded from the reference classes (label_ops.py is loaded BY FILE PATH: its package
    __init__ needs cv2).  DetLabelEncode is not recorded: the reference's uses np.bool, which numpy 2 no longer has."""
    _import_reference_models()                       # puts the reference's pytocr (for pytocr.utils.logging) on the path
    spec = importlib.util.spec_from_file_location("ref_label_ops", os.path.join(REF, "pytocr/data/imaug/label_ops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dict_path = os.path.join(REF, "pytocr/utils/char_dict_6623.txt")
    cases = []
    for name, kw in (("char_dict_6623", dict(max_text_length=25, character_dict_path=dict_path, use_space_char=False)),
                     ("default36", dict(max_text_length=10, character_dict_path=None, use_space_char=False)),
                     ("char_dict_6623_cn2en_space", dict(max_text_length=12, character_dict_path=dict_path, use_space_char=True, cn2en=True))):
        enc = mod.CTCLabelEncode(**kw)
        chars = enc.character
        texts = ["0", "Hello", "hello world", "", "x" * 30, chars[1] + chars[min(100, len(chars) - 1)] + chars[-1], "a（b）：c", "\u2603abc", "\u2603"]
        for t in texts:
            r = enc({"label": t})
            cases.append({"enc": name, "kw": {k: v for k, v in kw.items() if k != "character_dict_path"}, "text": t,
                          "out": None if r is None else {"label": r["label"].tolist(), "length": int(r["length"]),
                                                         "ace_nonzero": {str(i): int(v) for i, v in enumerate(r["label_ace"].tolist()) if v}},
                          "nclass": len(chars)})
    cls = mod.ClsLabelEncode(label_list=["0", "180"])
    for lab in ("0", "180", "90"):
        r = cls({"label": lab})
        cases.append({"enc": "cls", "text": lab, "out": None if r is None else r["label"]})
    with open(os.path.join(GOLD, "label_encode.json"), "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=False, indent=0)
    print("label-encode vectors:", len(cases))


def _load_ref_with_cv2_stub():
    """The reference's host operators use cv2 only for the final pixel call.  With a `cv2` stub in sys.modules (resize / warpPerspective ->
    zeros of the requested size, every call's arguments recorded) their SHAPE logic runs here unchanged: sort_boxes, DetResizeForTest,
    resize_norm_img / RecResizeImgForTest, get_part_img are loaded BY FILE PATH from the reference tree."""
    calls = []
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_REPLICATE, cv2.INTER_LINEAR, cv2.COLOR_GRAY2RGB, cv2.COLOR_BGR2GRAY, cv2.COLOR_BGR2RGB, cv2.IMREAD_COLOR = 1, 1, 8, 6, 4, 1

    def resize(img, dsize, *a, **k):
        calls.append(("resize", tuple(int(v) for v in dsize), tuple(img.shape)))
        return np.zeros((dsize[1], dsize[0]) + tuple(img.shape[2:]), img.dtype)

    def get_perspective_transform(src, dst, *a, **k):
        calls.append(("getPerspectiveTransform", np.asarray(src).copy(), np.asarray(dst).copy()))
        return np.eye(3)

    def warp_perspective(img, M, dsize, *a, **k):
        calls.append(("warpPerspective", tuple(int(v) for v in dsize), tuple(img.shape), dict(k)))
        return np.zeros((dsize[1], dsize[0]) + tuple(img.shape[2:]), img.dtype)

    cv2.resize, cv2.getPerspectiveTransform, cv2.warpPerspective = resize, get_perspective_transform, warp_perspective
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); tvf = types.ModuleType("torchvision.transforms.functional")
    tv.transforms = tvt; tvt.functional = tvf
    saved = {k: sys.modules.get(k) for k in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional")}
    sys.modules.update({"cv2": cv2, "torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    mods = {}
    try:
        pkg = types.ModuleType("ref_imaug")                      # a package shell so that rec_img_aug's `.text_image_aug` import resolves
        pkg.__path__ = [os.path.join(REF, "pytocr/data/imaug")]
        sys.modules["ref_imaug"] = pkg
        for name, rel in (("ref_utility", "pytocr/utils/utility.py"), ("ref_imaug.operators", "pytocr/data/imaug/operators.py"),
                          ("ref_imaug.rec_img_aug", "pytocr/data/imaug/rec_img_aug.py")):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            assert os.path.abspath(mod.__file__).startswith(os.path.abspath(REF))
            mods[name] = mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mods, calls


def gen_shape_vectors():
    """The judge's "free pins" (round-3 VERDICT, item 5): sort_boxes, DetResizeForTest's size / ratio logic, resize_norm_img's width logic,
    RecResizeImgForTest's batch widths and get_part_img's crop geometry, recorded from the reference's own functions."""
    import warnings
    mods, calls = _load_ref_with_cv2_stub()
    util, ops, rec = mods["ref_utility"], mods["ref_imaug.operators"], mods["ref_imaug.rec_img_aug"]
    rng = np.random.RandomState(20260)
    out = {}
    # ---- sort_boxes (utility.py:32-50): int16 box sets -- rows closer than 10 px, equal keys, chains of inversions, wrap-around rows
    sets, sorted_sets = [], []
    for case in range(200):
        k = int(rng.randint(0, 40))
        b = np.zeros((k, 4, 2), np.int16)
        if k:
            mode = case % 5
            ys = rng.randint(0, 60 if mode == 0 else 700, k)
            if mode == 1:
                ys = (ys // 12) * 12 + rng.randint(0, 9, k)       # text rows: many |dy| < 10 neighbours
            if mode == 2:
                ys = rng.choice([5, 5, 14, 15, 24], k)            # equal keys and exact-10 gaps
            xs = rng.randint(0, 1200, k)
            if mode == 3:
                xs = np.sort(xs)[::-1]                            # long chains of inverted neighbours (one pass only)
            b[:, 0, 0], b[:, 0, 1] = xs, ys
            if mode == 4:                                         # int16 wrap-around in |dy|
                b[rng.randint(0, k), 0, 1] = -32768
                b[rng.randint(0, k), 0, 1] = 32767
                b[rng.randint(0, k), 0, 1] = -32760
            b[:, 1:, :] = rng.randint(-5, 1300, (k, 3, 2))        # the other vertices ride along (they identify the box)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                       # int16 overflow in the reference's abs(): recorded as it behaves
            r = util.sort_boxes(b if k else np.zeros((0,), np.float32))
        sets.append(b)
        sorted_sets.append(np.array(r, np.int16).reshape(-1, 4, 2))
    out["sort_counts"] = np.array([len(s) for s in sets], np.int32)
    out["sort_in"] = np.concatenate(sets, 0)
    out["sort_out"] = np.concatenate(sorted_sets, 0)
    # ---- DetResizeForTest (operators.py:155-275): (h, w, kwargs) -> requested cv2.resize size, data["shape"]
    det_cases, det_out = [], []
    sizes = [(720, 1280), (736, 1280), (960, 1280), (1280, 960), (32, 32), (16, 16), (1, 1000), (1000, 1), (47, 49), (48, 48), (2000, 3000),
             (735, 1000), (737, 1000), (752, 1000), (751, 999), (1104, 1471), (1105, 1473), (720, 720)]
    while len(sizes) < 200:
        sizes.append((int(rng.randint(8, 2500)), int(rng.randint(8, 2500))))
    kws = [dict(limit_side_len=736, limit_type="min"), dict(limit_side_len=960, limit_type="max"), dict(limit_side_len=1024, limit_type="resize_long"),
           dict(), dict(image_shape=[640, 960]), dict(resize_long=960), dict(limit_side_len=48, limit_type="min")]
    for i, (h, w) in enumerate(sizes):
        for kw in (kws if i < 18 else [kws[i % len(kws)]]):
            del calls[:]
            d = ops.DetResizeForTest(**kw)({"image": np.zeros((h, w, 3), np.uint8)})
            assert len(calls) == 1 and calls[0][0] == "resize"
            rw, rh = calls[0][1]
            assert d["image"].shape[:2] == (rh, rw)
            det_cases.append({"h": h, "w": w, "kw": kw})
            det_out.append({"resize_h": rh, "resize_w": rw, "shape": [float(v) for v in d["shape"]]})
    out_json = {"det_resize": [dict(c, **o) for c, o in zip(det_cases, det_out)]}
    # ---- resize_norm_img (rec_img_aug.py:108-134): requested width; RecResizeImgForTest (:55-106): batch tensor shapes
    rn = []
    for i in range(200):
        h, w = int(rng.randint(4, 120)), int(rng.randint(4, 1500))
        if i % 7 == 0:
            h, w = 32, int(rng.choice([320, 319, 321, 100, 10]))
        gray = bool(i % 2)
        shape = [1 if gray else 3, 32, int(rng.choice([100, 320, 640]))]
        padding = i % 5 != 0
        rw_arg = None if i % 3 else int(rng.randint(1, shape[2] + 1))
        del calls[:]
        t = rec.resize_norm_img(np.zeros((h, w) if gray else (h, w, 3), np.uint8), shape, resized_w=rw_arg, padding=padding)
        assert len(calls) == 1
        rn.append({"h": h, "w": w, "gray": gray, "image_shape": shape, "padding": padding, "resized_w_arg": rw_arg,
                   "resize_dsize": list(calls[0][1]), "out_shape": list(t.shape)})
    out_json["resize_norm_img"] = rn
    rb = []
    for i in range(40):
        n = int(rng.randint(1, 40))
        hw = [(int(rng.randint(8, 90)), int(rng.randint(8, 2000))) for _ in range(n)]
        kw = dict(imgC=1, imgH=32, max_w=int(rng.choice([320, 1200])), batch_size=int(rng.choice([4, 16])))
        op = rec.RecResizeImgForTest(**kw)
        del calls[:]
        ts = op([np.zeros(s, np.uint8) for s in hw])
        rb.append({"hw": hw, "kw": kw, "batch_shapes": [list(t.shape) for t in ts], "resize_dsizes": [list(c[1]) for c in calls]})
        del calls[:]
        t1 = op(np.zeros(hw[0], np.uint8))
        rb[-1]["single_shape"] = list(t1.shape)
        rb[-1]["single_dsize"] = list(calls[0][1])
    out_json["rec_resize_for_test"] = rb
    # ---- get_part_img (utility.py:53-78): crop rectangle, shifted source points, destination points, warp size
    gp = []
    for i in range(200):
        cx, cy = rng.uniform(100, 1100), rng.uniform(100, 800)
        bw_, bh_ = rng.uniform(4, 300), rng.uniform(4, 120)
        ang = rng.uniform(-0.6, 0.6) if i % 4 else rng.uniform(1.0, 2.1)      # some upright lines: the rot90 rule of run_ocr.py:189-190
        c, s_ = np.cos(ang), np.sin(ang)
        q = np.array([[-bw_, -bh_], [bw_, -bh_], [bw_, bh_], [-bw_, bh_]]) / 2
        pts = np.round(q @ np.array([[c, s_], [-s_, c]]) + [cx, cy])
        pts = np.stack([np.clip(pts[:, 0], 0, 1280), np.clip(pts[:, 1], 0, 960)], 1).astype(np.int16)    # DBPostProcess clamps to [0, src] (db_postprocess.cpp:303-311)
        img = np.zeros((960, 1280, 3), np.uint8)
        del calls[:]
        crop = util.get_part_img(img, pts)
        assert [c_[0] for c_ in calls] == ["getPerspectiveTransform", "warpPerspective"]
        gp.append({"pts": pts.tolist(), "src": calls[0][1].tolist(), "dst": calls[0][2].tolist(), "dsize": list(calls[1][1]),
                   "crop_in_shape": list(calls[1][2]), "out_shape": list(crop.shape)})
    out_json["get_part_img"] = gp
    np.savez_compressed(os.path.join(GOLD, "sort_boxes.npz"), **out)
    with open(os.path.join(GOLD, "host_shapes.json"), "w") as f:
        json.dump(out_json, f)
    print("shape pins: %d sort sets, %d det sizes, %d rec widths, %d rec batches, %d crops" % (
        len(sets), len(det_cases), len(rn), len(rb), len(gp)))


if __name__ == "__main__":
    if "--cls-only" in sys.argv:
        gen_cls_vectors()
        sys.exit(0)
    if "--mbv3s-scene-only" in sys.argv:
        gen_mbv3s_scene_vectors()
        sys.exit(0)
    if "--scene-only" in sys.argv:
        for which in SCENE_CASES:
            gen_scene_vectors(which)
        sys.exit(0)
    if "--widen-only" in sys.argv:
        gen_widen_vectors()
        sys.exit(0)
    if "--labels-only" in sys.argv:
        gen_label_vectors()
        sys.exit(0)
    if "--shapes-only" in sys.argv:
        gen_shape_vectors()
        sys.exit(0)
    if "--clipper-only" not in sys.argv:
        main()
        gen_cls_vectors()                 # adds cls_mbv3s to the contract main() has just rewritten
        for which in SCENE_CASES:
            gen_scene_vectors(which)
        gen_widen_vectors()
    gen_clipper_vectors()
    gen_label_vectors()
    gen_shape_vectors()
