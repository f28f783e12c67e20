"""DBNet-r18 on the HIP engine against outputs of the REFERENCE model (tests/golden) and the oracle (-m gpu).
Tolerance from BASELINE.json north_star: fp32 probability maps within 1e-4."""
import os

import numpy as np
import pytest
import torch

from pytorchocr_amd.utils.synth import synth_images, synth_state_dict

pytestmark = pytest.mark.gpu

DET_R18 = dict(model_type="det", algorithm="DB", Transform=None,
               Backbone=dict(name="ResNet", layers=18, pretrained=False),
               Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False, attention_type="scale_channel_spatial"),
               Head=dict(name="DBHead", k=50))


def _model(contract, **extra):
    from pytorchocr_amd.modeling.architectures import build_model
    cfg = dict(DET_R18, **extra)
    m = build_model(cfg)
    sd = synth_state_dict(contract["det_r18_db"])
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval()


def test_maps_and_features_match_reference_small(gold_dir, contract):
    g = np.load(os.path.join(gold_dir, "det_r18_db_1x3x64x96.npz"))
    m = _model(contract, return_all_feats=True)
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=int(g["seed"]))).to("cuda:0")
    with torch.no_grad():
        y = m(x)
    for k, f in zip(("c2", "c3", "c4", "c5"), y["backbone_out"]):
        err = np.abs(f.cpu().numpy() - g[k]).max()
        assert err <= 1e-4 * max(1.0, np.abs(g[k]).max()), (k, err)
    assert np.abs(y["neck_out"].cpu().numpy() - g["neck"]).max() <= 1e-4 * np.abs(g["neck"]).max()
    maps = y["maps"].cpu().numpy()
    assert maps.shape == g["maps"].shape and maps.dtype == np.float32
    assert np.abs(maps - g["maps"]).max() <= 1e-4


def test_maps_match_reference_batch(gold_dir, contract):
    g = np.load(os.path.join(gold_dir, "det_r18_db_2x3x96x160.npz"))
    m = _model(contract)
    x = torch.from_numpy(synth_images(2, 3, 96, 160, seed=int(g["seed"]))).to("cuda:0")
    with torch.no_grad():
        y = m(x)
    assert set(y.keys()) == {"maps"}
    assert np.abs(y["maps"].cpu().numpy() - g["maps"]).max() <= 1e-4


def test_full_size_against_oracle(contract):
    """BASELINE config size 736x1280 (one image): HIP maps vs the torch-fp32 oracle on the same input."""
    from oracle import model_oracle
    m = _model(contract)
    xs = synth_images(1, 3, 736, 1280, seed=3)
    with torch.no_grad():
        y = m(torch.from_numpy(xs).to("cuda:0"))["maps"].cpu().numpy()
    ref = model_oracle.dbnet_r18_forward(synth_state_dict(contract["det_r18_db"]), torch.from_numpy(xs))["maps"].numpy()
    assert y.shape == (1, 1, 736, 1280)
    assert np.abs(y - ref).max() <= 1e-4


def test_bench_batch_properties(contract):
    """BASELINE configs[1] size, 32 x 736 x 1280 (too big for the CPU oracle in seconds): size-independent properties --
    an image's maps do not depend on its position in the batch or on the batch size (every kernel, incl. the Winograd
    patch geometries and the persistent tile walkers, computes each output in a fixed order), repeated runs are
    bit-identical, values are probabilities, and the DB post-process of the batch equals the post-process per image."""
    from pytorchocr_amd.postprocess import build_post_process
    m = _model(contract)
    base = torch.from_numpy(synth_images(4, 3, 736, 1280, seed=11)).to("cuda:0")
    x = base.repeat(8, 1, 1, 1).contiguous()
    with torch.no_grad():
        y = m(x)["maps"]
        y2 = m(x)["maps"]
        y1 = m(base[:1].contiguous())["maps"]
    assert y.shape == (32, 1, 736, 1280)
    assert torch.equal(y, y2)
    for i in range(4, 32):
        assert torch.equal(y[i], y[i % 4]), i
    assert torch.equal(y[0], y1[0])
    assert float(y.min()) >= 0.0 and float(y.max()) <= 1.0 and bool(torch.isfinite(y).all())
    post = build_post_process(dict(name="DBPostProcess", thresh=0.3, box_thresh=0.5, max_candidates=1000, unclip_ratio=1.7,
                                   score_mode="poly", cpp_speedup=True), dict())
    shape_list = np.array([[736, 1280, 1.0, 1.0]] * 32)
    allb = post({"maps": y}, shape_list)
    one = post({"maps": y[:1].contiguous()}, shape_list[:1])
    assert np.array_equal(allb[0]["points"], one[0]["points"])
    for i in range(4, 32):
        assert np.array_equal(allb[i]["points"], allb[i % 4]["points"]), i


def test_cpu_input_fails_loudly(contract):
    m = _model(contract)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))


def test_dbpp_asf_matches_reference(gold_dir, contract):
    """DB++ (use_asf=True, scale_channel_spatial): maps vs the reference's own output, and a larger ragged batch vs the oracle."""
    from oracle import model_oracle
    from pytorchocr_amd.modeling.architectures import build_model
    cfg = dict(DET_R18, Neck=dict(DET_R18["Neck"], use_asf=True))
    m = build_model(cfg)
    sd = synth_state_dict(contract["detpp_r18_db"])
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    g = np.load(os.path.join(gold_dir, "detpp_r18_db_1x3x64x96.npz"))
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=int(g["seed"]))).cuda()
    with torch.no_grad():
        y = m(x)["maps"].cpu().numpy()
    assert np.abs(y - g["maps"]).max() <= 1e-4
    xs = synth_images(3, 3, 160, 224, seed=21)
    with torch.no_grad():
        y = m(torch.from_numpy(xs).cuda())["maps"].cpu().numpy()
    ref = model_oracle.dbnet_r18_forward(sd, torch.from_numpy(xs))["maps"].numpy()
    assert np.abs(y - ref).max() <= 1e-4


def test_mobilenetv3_small_db_matches_reference(gold_dir, contract):
    """DB with the MobileNetV3-small x1.0 backbone (FPN 96): maps vs the reference's own output and vs the oracle."""
    from oracle import model_oracle
    from pytorchocr_amd.modeling.architectures import build_model
    cfg = dict(model_type="det", algorithm="DB", Transform=None,
               Backbone=dict(name="MobileNetV3", model_name="small", scale=1.0, pretrained=False),
               Neck=dict(name="FPN", out_channels=96, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50), return_all_feats=True)
    m = build_model(cfg)
    sd = synth_state_dict(contract["det_mbv3s_db"])
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    g = np.load(os.path.join(gold_dir, "det_mbv3s_db_1x3x64x96.npz"))
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=int(g["seed"]))).cuda()
    with torch.no_grad():
        y = m(x)
    assert [f.shape[1] for f in y["backbone_out"]] == [16, 24, 48, 576]
    assert np.abs(y["maps"].cpu().numpy() - g["maps"]).max() <= 1e-4
    xs = synth_images(2, 3, 224, 320, seed=31)
    ref = model_oracle.dbnet_forward(sd, torch.from_numpy(xs), return_feats=True)
    with torch.no_grad():
        y = m(torch.from_numpy(xs).cuda())
    for a, b in zip(y["backbone_out"], ref["backbone_out"]):
        assert (a.cpu() - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    assert np.abs(y["maps"].cpu().numpy() - ref["maps"].numpy()).max() <= 1e-4


MBV3L = dict(model_type="det", algorithm="DB", Transform=None,
             Backbone=dict(name="MobileNetV3", model_name="large", width_mult=1.0, use_se=True, pretrained=False),
             Neck=dict(name="FPN", out_channels=96, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50), return_all_feats=True)


def test_mobilenetv3_large_db_matches_reference(gold_dir, contract):
    """DB with the MobileNetV3-LARGE x1.0 backbone -- what the stock configs/det/det_mbv3_db.yml:24-27 builds: maps and C2..C5 vs the
    reference's own outputs, and vs the oracle at a second size."""
    from oracle import model_oracle
    from pytorchocr_amd.modeling.architectures import build_model
    m = build_model(dict(MBV3L))
    sd = synth_state_dict(contract["det_mbv3l_db"])
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    g = np.load(os.path.join(gold_dir, "det_mbv3l_db_1x3x64x96.npz"))
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=int(g["seed"]))).cuda()
    with torch.no_grad():
        y = m(x)
    assert [f.shape[1] for f in y["backbone_out"]] == [24, 40, 112, 960]
    assert np.abs(y["maps"].cpu().numpy() - g["maps"]).max() <= 1e-4
    for i, f in enumerate(y["backbone_out"]):
        ref = g["c%d" % (i + 2)]
        assert np.abs(f.cpu().numpy() - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max())), "C%d" % (i + 2)
    xs = synth_images(2, 3, 224, 320, seed=32)
    ref = model_oracle.dbnet_forward(sd, torch.from_numpy(xs), return_feats=True)
    with torch.no_grad():
        y = m(torch.from_numpy(xs).cuda())
    for a, b in zip(y["backbone_out"], ref["backbone_out"]):
        assert (a.cpu() - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    assert np.abs(y["maps"].cpu().numpy() - ref["maps"].numpy()).max() <= 1e-4


WIDENED = {
    "detpp_r18_db_spatial": dict(DET_R18, Neck=dict(DET_R18["Neck"], use_asf=True, attention_type="scale_spatial")),
    "detpp_r18_db_channel": dict(DET_R18, Neck=dict(DET_R18["Neck"], use_asf=True, attention_type="scale_channel")),
    "det_r50_db": dict(DET_R18, Backbone=dict(name="ResNet", layers=50, pretrained=False)),
    "det_r18_db_3x3stem": dict(DET_R18, Backbone=dict(name="ResNet", layers=18, mode_3x3=True, pretrained=False)),
}


@pytest.mark.parametrize("name", sorted(WIDENED))
def test_reference_options_next_to_the_headline_configs(name, gold_dir, contract):
    """ASF attention types scale_spatial / scale_channel (necks/asf.py:9-29,78-107), Bottleneck ResNet-50 (det_resnet.py:85-140) and the
    three-conv 3x3 stem (det_resnet.py:196-206): reference key names (strict load), maps within 1e-4 of outputs of the reference itself"""
    from pytorchocr_amd.modeling.architectures import build_model
    g = np.load(os.path.join(gold_dir, "%s_1x3x64x96.npz" % name))
    m = build_model(dict(WIDENED[name], return_all_feats="c2" in g))
    sd = synth_state_dict(contract[name])
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=int(g["seed"]))).to("cuda:0")
    with torch.no_grad():
        y = m(x)
    if "c2" in g:
        for k, f in zip(("c2", "c3", "c4", "c5"), y["backbone_out"]):
            err = np.abs(f.cpu().numpy() - g[k]).max()
            assert f.shape == g[k].shape and err <= 1e-4 * max(1.0, np.abs(g[k]).max()), (k, err)
    maps = y["maps"].cpu().numpy()
    assert maps.shape == g["maps"].shape and np.abs(maps - g["maps"]).max() <= 1e-4, np.abs(maps - g["maps"]).max()
    # a batch of four copies at another size: the same bits per image, and images do not influence each other
    xb = torch.from_numpy(synth_images(2, 3, 96, 160, seed=5)).to("cuda:0")
    with torch.no_grad():
        a = m(xb.repeat(2, 1, 1, 1))["maps"]
        b = m(xb[:1])["maps"]
    assert torch.equal(a[0], a[2]) and torch.equal(a[1], a[3]) and torch.equal(a[0], b[0])


# ---- scene checkpoints (utils/synth.py: synth_scene_state_dict): random backbone / neck, a fitted read-out of the scene's brightness in
# two head channels, logit gain 14 -- the maps are text-like, cross thresh and box_thresh, and produce boxes.  bench.py times these
# checkpoints (configs[1], configs[4]); golden = outputs of the REFERENCE model (tools/gen_golden.py --scene-only).
SCENES = {"r18": (DET_R18, "det_r18_db", "det_r18_scene_1x3x224x320.npz"),
          "detpp": (dict(DET_R18, Neck=dict(DET_R18["Neck"], use_asf=True)), "detpp_r18_db", "detpp_r18_scene_1x3x224x320.npz")}


@pytest.mark.parametrize("which", sorted(SCENES))
def test_scene_checkpoint_maps_and_boxes_against_the_reference(which, gold_dir, contract):
    from oracle import dbpost
    from pytorchocr_amd.modeling.architectures import build_model
    from pytorchocr_amd.postprocess import build_post_process
    from pytorchocr_amd.utils.synth import load_scene_readout, synth_scene_inputs, synth_scene_state_dict
    cfg, key, fixture = SCENES[which]
    g = np.load(os.path.join(gold_dir, fixture))
    sd = synth_scene_state_dict(contract[key], *load_scene_readout(which))
    m = build_model(dict(cfg, return_all_feats=True))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    h, w = 224, 320
    x = torch.from_numpy(synth_scene_inputs(1, h, w, seed=int(g["seed"]))).cuda()
    with torch.no_grad():
        y = m(x)
    ref = g["maps"]
    # the comparison means something: the reference's map lives on both sides of both thresholds
    assert ref.max() - ref.min() > 0.99 and 0.15 <= (ref > 0.3).mean() <= 0.8 and 0.15 <= (ref > 0.5).mean() <= 0.8
    for got, exp in ((y["neck_out"].cpu().numpy()[:, :, ::4, ::4], g["neck_sub"]), (y["backbone_out"][0].cpu().numpy()[:, :, ::4, ::4], g["c2_sub"]),
                     (y["backbone_out"][3].cpu().numpy(), g["c5"])):
        assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max()
    maps = y["maps"].cpu().numpy()
    assert np.abs(maps - ref).max() <= 1e-4, np.abs(maps - ref).max()         # north_star's fp32 bar, gain 14 included
    assert ((maps > 0.3) != (ref > 0.3)).mean() <= 1e-4
    # boxes: the HIP post-process on the HIP maps against the oracle pipeline on the REFERENCE's maps
    post = build_post_process(dict(name="DBPostProcess", thresh=0.3, box_thresh=0.5, unclip_ratio=1.7, cpp_speedup=True), {})
    got = post({"maps": y["maps"]}, np.array([[h, w, 1.0, 1.0]]))[0]["points"]
    exp = dbpost.boxes_from_bitmap(ref[0, 0], dbpost.binarize(ref[0, 0], 0.3), 0.5, 1.7, w, h).astype(np.int16)
    assert len(exp) >= 6, len(exp)
    same = sum(any(np.array_equal(a, b) for b in exp) for a in got)
    assert abs(len(got) - len(exp)) <= 1 and same >= len(exp) - 1, (len(got), len(exp), same)


def test_post_process_beside_the_next_forward_returns_the_boxes_it_returns_alone(contract):
    """The bench's (and a service's) configuration: DBPostProcess.submit queues the post-process of batch i on its own stream and the
    convolutions of batch i + 1 run beside it.  The boxes of every overlapped call must be the boxes of the same maps post-processed
    alone -- different scenes per step, so that a result leaking from one call into the next (a stale count, a ready word of another
    call) would show; the maps themselves must not change either.  Twelve steps of 8 scenes at the bench's 736 x 1280."""
    from pytorchocr_amd.modeling.architectures import build_model
    from pytorchocr_amd.postprocess import build_post_process
    from pytorchocr_amd.utils.synth import load_scene_readout, synth_scene_inputs, synth_scene_state_dict
    cfg, key, _ = SCENES["r18"]
    sd = synth_scene_state_dict(contract[key], *load_scene_readout("r18"))
    m = build_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    n, h, w = 8, 736, 1280
    xs = [torch.from_numpy(synth_scene_inputs(n, h, w, seed=900 + i)).cuda() for i in range(3)]
    shapes = np.array([[h, w, 1.0, 1.0]] * n)
    post = build_post_process(dict(name="DBPostProcess", thresh=0.3, box_thresh=0.5, unclip_ratio=1.7, cpp_speedup=True), {})
    alone, maps_alone = [], []
    with torch.no_grad():
        for x in xs:
            mp = m(x)["maps"]
            torch.cuda.synchronize()
            maps_alone.append(mp.clone())
            alone.append(post({"maps": mp}, shapes))
    assert min(len(b["points"]) for r in alone for b in r) >= 50
    pending, got = None, []
    with torch.no_grad():
        for step in range(12):
            mp = m(xs[step % 3])["maps"]
            assert torch.equal(mp, maps_alone[step % 3])
            fut = post.submit({"maps": mp}, shapes)
            if pending is not None:
                got.append(pending.result())
            pending = fut
        got.append(pending.result())
    torch.cuda.synchronize()
    for step, res in enumerate(got):
        exp = alone[step % 3]
        for a, b in zip(res, exp):
            assert len(a["points"]) == len(b["points"]) and np.array_equal(np.asarray(a["points"]), np.asarray(b["points"])), step


@pytest.mark.parametrize("seed", range(6 + int(os.environ.get("PTOCR_MODEL_FUZZ", "0"))))      # PTOCR_MODEL_FUZZ=n: n more seeds
def test_detectors_random_sizes_against_the_oracle(seed, contract):
    """DBNet-r18 / DBNet++-r18 / DBNet-mbv3s at random input sizes (multiples of 32, as DetResizeForTest makes them) and batch sizes
    against the torch-fp32 oracle: every patch geometry / kernel choice the size-dependent cost models make must give the same maps"""
    from oracle import model_oracle
    from pytorchocr_amd.modeling.architectures import build_model
    rng = np.random.default_rng(6000 + seed)
    which = ("det_r18_db", "detpp_r18_db", "det_mbv3s_db")[int(rng.integers(0, 3))]
    cfg = {"det_r18_db": DET_R18, "detpp_r18_db": dict(DET_R18, Neck=dict(DET_R18["Neck"], use_asf=True)),
           "det_mbv3s_db": dict(model_type="det", algorithm="DB", Transform=None,
                                Backbone=dict(name="MobileNetV3", model_name="small", scale=1.0, pretrained=False),
                                Neck=dict(name="FPN", out_channels=96, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50))}[which]
    sd = synth_state_dict(contract[which])
    m = build_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    n, h, w = int(rng.integers(1, 4)), 32 * int(rng.integers(1, 12)), 32 * int(rng.integers(1, 14))
    xs = synth_images(n, 3, h, w, seed=600 + seed)
    with torch.no_grad():
        y = m(torch.from_numpy(xs).cuda())["maps"].cpu().numpy()
    fwd = model_oracle.dbnet_r18_forward if which == "detpp_r18_db" else model_oracle.dbnet_forward     # (the r18 form also runs the ASF neck)
    ref = fwd(sd, torch.from_numpy(xs))["maps"].numpy()
    assert y.shape == ref.shape and np.abs(y - ref).max() <= 1e-4, (which, n, h, w, np.abs(y - ref).max())


@pytest.mark.parametrize("shape", [(1, 64, 96), (3, 40, 72), (2, 184, 320), (5, 8, 136)])
def test_pyramid_conv_equals_the_conv_on_the_materialised_concat(shape):
    """ptocr_conv3x3_wino4r_pyramid_f32: the head conv reading the FPN's four planes in place (pixel (y >> s, x >> s) of a plane) gives
    the bits of ptocr_conv3x3_wino4r_f32 on cat(up8(p5), up4(p4), up2(p3), p2) (reference fpn.py:118-131, det_db_head.py:9-17)"""
    from torch import nn
    from pytorchocr_amd.modeling import ops
    N, H, W = shape
    torch.manual_seed(N * 1000 + H)
    conv, bn = nn.Conv2d(256, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64).eval()
    bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.1)
    pc = ops.PackedConv(conv, bn, torch.device("cuda:0"), relu=True)
    if not ops.pyramid_conv_ok(pc, N, H, W):
        pytest.skip("the F(4x4) kernel is not the one chosen for this map")
    pyr = ops.Pyramid(N, H, W, (3, 2, 1, 0), torch.device("cuda:0"))
    pyr.buf.copy_(torch.randn(pyr.buf.numel(), device="cuda:0"))
    got = ops.conv3x3_pyramid(pyr, pc)
    want = ops.conv2d(pyr.materialize(), pc)
    assert got.shape == want.shape
    assert torch.equal(got, want), float((got - want).abs().max())


def test_detector_with_the_fpn_pyramid_equals_the_concat_path(contract):
    """DBNet-r18 with the FPN output handed to the head as a pyramid (default) and as the concat tensor (PTOCR_FPN_PYRAMID=0): same maps,
    bit for bit, and the concat the reference would build is still there when features are asked for"""
    from pytorchocr_amd.modeling import ops
    m = _model(contract)
    x = torch.from_numpy(synth_images(2, 3, 256, 448, seed=5)).to("cuda:0")
    keep = ops.USE_PYRAMID
    try:
        ops.USE_PYRAMID = True
        with torch.no_grad():
            a = m(x)["maps"]
        feats = m.backbone.forward_from_nchw(x) if hasattr(m.backbone, "forward_from_nchw") else None
        if feats is not None:
            neck = m._neck_nhwc(feats)
            assert isinstance(neck, ops.Pyramid), "the pyramid path was not taken at 64 x 112 (H / 4 x W / 4)"
        ops.USE_PYRAMID = False
        with torch.no_grad():
            b = m(x)["maps"]
        if feats is not None:
            assert torch.equal(neck.materialize(), m._neck_nhwc(feats))
    finally:
        ops.USE_PYRAMID = keep
    assert torch.equal(a, b)


@pytest.mark.parametrize("attention", ["scale_channel_spatial", "scale_spatial", "scale_channel"])
def test_dbpp_on_the_fpn_pyramid_equals_the_concat_path(attention, contract):
    """DB++ (round 6): the ASF's 3x3 conv reads the four own-resolution planes in place and its re-weighting writes the concat once
    (pytocr/modeling/necks/fpn.py:118-131 + asf.py:146-162 without the upsampled copies); with PTOCR_FPN_PYRAMID=0 the round-1 form
    (upsampling stores into the concat, re-weighted in place).  Same neck output and same maps, bit for bit, for all three attention types."""
    from pytorchocr_amd.modeling import ops
    from pytorchocr_amd.modeling.architectures import build_model
    name = {"scale_channel_spatial": "detpp_r18_db", "scale_spatial": "detpp_r18_db_spatial", "scale_channel": "detpp_r18_db_channel"}[attention]
    m = build_model(dict(DET_R18, Neck=dict(DET_R18["Neck"], use_asf=True, attention_type=attention)))
    sd = synth_state_dict(contract[name])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    x = torch.from_numpy(synth_images(3, 3, 256, 448, seed=8)).to("cuda:0")
    keep = ops.USE_PYRAMID
    calls = []
    orig = ops.conv3x3_pyramid
    try:
        ops.USE_PYRAMID = True
        ops.conv3x3_pyramid = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        with torch.no_grad():
            a = m(x)["maps"]
            feats = m.backbone.forward_from_nchw(x)
            neck_a = m._neck_nhwc(feats)
        assert calls, "the pyramid path was not taken at 64 x 112 (H / 4 x W / 4)"
        ops.USE_PYRAMID = False
        n_before = len(calls)
        with torch.no_grad():
            b = m(x)["maps"]
            neck_b = m._neck_nhwc(feats)
        assert len(calls) == n_before
    finally:
        ops.USE_PYRAMID = keep
        ops.conv3x3_pyramid = orig
    assert torch.is_tensor(neck_a) and torch.equal(neck_a, neck_b)
    assert torch.equal(a, b)
