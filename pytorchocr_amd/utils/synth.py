"""Portable synthetic weights / inputs (no torch RNG, no files).

BASELINE.json asks for synthetic data of the reference's shapes; the GPU box has
no checkpoints and no reference tree, so every weight tensor and every input is
regenerated from a counter-based generator (splitmix64) that gives the same
float32 bits on any host.  The golden fixtures under tests/golden/ were produced
by feeding exactly these tensors to the reference model (tools/gen_golden.py).

State-dict key contract: SURVEY.md Appendix A (reference
pytocr/utils/save_load.py:81-101 loads with strict=True).
"""
import zlib

import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(idx, seed):
    """idx: array of uint64 counters; returns uint64 hash per counter."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(n, seed):
    """n float32 values in [0,1) with 24 random bits each."""
    z = splitmix64(np.arange(n, dtype=np.uint64), seed)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def uniform(shape, seed, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(n, seed)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32).reshape(shape)


def key_seed(key, seed):
    return (zlib.crc32(key.encode("utf-8")) * 2654435761 + seed * 97) & 0xFFFFFFFFFFFF


def _fan_in(shape, transposed):
    if len(shape) == 4:
        rf = shape[2] * shape[3]
        return (shape[0] if transposed else shape[1]) * rf
    if len(shape) == 2:
        return shape[1]
    return 1


def synth_state_dict(ref_shapes, seed=2022):
    """ref_shapes: ordered {key: (shape tuple, dtype str)} -> {key: np.ndarray}.

    Rules (by key suffix) keep activations O(1) through the net so that the 1e-4
    probability-map tolerance is a real test, not a saturated sigmoid:
      conv / linear weights        : U(-a, a), a = sqrt(3 / fan_in)   (var = 1/fan_in);
                                     CRNN VGG convs and LSTM weight_ih use sqrt(6 / fan_in),
                                     weight_hh 1/sqrt(H), head.fc 12x, embedding 2x (logit spread)
      *.bias (conv/linear/lstm)    : U(-0.1, 0.1)
      BN weight U(0.6, 1.4); BN bias U(-0.2, 0.2); running_mean U(-0.2, 0.2);
      running_var U(0.5, 1.5); num_batches_tracked = 0
    """
    out = {}
    bn_prefixes = {k[: -len(".running_var")] for k in ref_shapes if k.endswith(".running_var")}
    for key, (shape, dtype) in ref_shapes.items():
        s = key_seed(key, seed)
        prefix = key.rsplit(".", 1)[0]
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros(shape, dtype=np.int64)
        elif prefix in bn_prefixes:
            if key.endswith(".weight"):
                out[key] = uniform(shape, s, 0.6, 1.4)
            elif key.endswith(".running_var"):
                out[key] = uniform(shape, s, 0.5, 1.5)
            else:  # bias, running_mean
                out[key] = uniform(shape, s, -0.2, 0.2)
        elif "bias" in key.rsplit(".", 1)[-1]:
            out[key] = uniform(shape, s, -0.1, 0.1)
            if key == "head.fc.bias":                  # make the CTC blank (index 0) win often
                out[key][0] = np.float32(13.0)
        else:
            transposed = (".binarize.3." in key or ".binarize.6." in key
                          or ".thresh.3." in key or ".thresh.6." in key)
            fan = max(_fan_in(shape, transposed), 1)
            if "weight_hh" in key:
                a = 1.0 / np.sqrt(shape[1])
            elif "weight_ih" in key:
                a = np.sqrt(6.0 / fan)
            elif key.startswith("backbone.cnn."):      # CRNN VGG stack (plain conv+ReLU chain)
                a = np.sqrt(6.0 / fan)
            elif key == "head.fc.weight":              # CTC logits with a real spread
                a = 12.0 * np.sqrt(3.0 / fan)
            elif ".embedding." in key:
                a = 2.0 * np.sqrt(3.0 / fan)
            else:
                a = np.sqrt(3.0 / fan)
            out[key] = uniform(shape, s, -a, a)
    return out


def synth_images(n, c, h, w, seed=2022):
    """Normalised float32 NCHW images: u8 uniform -> /255 -> ImageNet mean/std
    (reference pytocr/data/imaug/operators.py:41-112) for c==3; (x/255-0.5)/0.5
    for c==1 (reference pytocr/data/imaug/rec_img_aug.py:108-134)."""
    u8 = (uniform01(n * c * h * w, seed ^ 0x5EED) * np.float32(256.0)).astype(np.uint8)
    x = u8.astype(np.float32).reshape(n, c, h, w) / np.float32(255.0)
    if c == 3:
        mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
        std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
        return ((x - mean) / std).astype(np.float32)
    return ((x - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def synth_text_lines(n, h, w, seed=2022):
    """Gray text-line-like crops f32[n,1,h,w]: 8-px column bands with their own level plus
    noise, so the CRNN output varies along T (uniform noise gives a constant argmax).
    Normalised as reference pytocr/data/imaug/rec_img_aug.py:108-134: (x/255 - 0.5)/0.5."""
    nb = (w + 7) // 8
    base = uniform01(n * nb, seed ^ 0xBA5E).reshape(n, 1, 1, nb)
    base = np.repeat(base, 8, axis=3)[:, :, :, :w]
    noise = uniform01(n * h * w, seed ^ 0x5EED).reshape(n, 1, h, w)
    u8 = np.clip(base * np.float32(224.0) + noise * np.float32(32.0), 0, 255).astype(np.uint8)
    x = u8.astype(np.float32) / np.float32(255.0)
    return ((x - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def synth_prob_maps(n, h, w, seed=0, noise=0.0):
    """Text-like probability maps f32[n,h,w] for the post-process: text lines laid out in jittered rows
    (about 150 per 736x1280 map: 20-400 px long, 8-40 px high, small rotations, some steep, some touching,
    some with holes), soft edges, values kept >= 2e-3 away from 0.3 and 0.5
    (SURVEY.md 8d 'post-process stress input')."""
    out = np.zeros((n, h, w), np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for i in range(n):
        r = uniform01(4096 * 8, seed * 7919 + i * 104729 + 17).reshape(4096, 8)
        m = np.zeros((h, w), np.float32)
        qi = 0
        y0 = 6.0
        while y0 < h - 8 and qi < 4000:
            row_h = 10 + r[qi, 0] * 34
            qi += 1
            x0 = r[qi, 1] * 40
            while x0 < w - 10 and qi < 4000:
                q = r[qi]
                qi += 1
                L = 20 + q[2] * q[2] * 380
                T = min(row_h, 8 + q[3] * 32)
                th = (q[4] - 0.5) * min(0.25, 16.0 / L) if (q[5] < 0.85 or L > 70) else (q[4] - 0.5) * 1.6
                cx, cy = x0 + L / 2, y0 + row_h / 2 + (q[6] - 0.5) * 6
                c, s_ = np.float32(np.cos(th)), np.float32(np.sin(th))
                ys, ye = int(max(0, cy - L / 2 - T - 3)), int(min(h, cy + L / 2 + T + 3))
                xs, xe = int(max(0, cx - L / 2 - T - 3)), int(min(w, cx + L / 2 + T + 3))
                if ye > ys and xe > xs:
                    X, Y = xx[ys:ye, xs:xe], yy[ys:ye, xs:xe]
                    u = (X - cx) * c + (Y - cy) * s_
                    v = -(X - cx) * s_ + (Y - cy) * c
                    d = np.maximum(np.abs(u) - L / 2, np.abs(v) - T / 2)
                    base = 0.58 + 0.4 * q[7]
                    val = (base - (base - 0.34) * np.clip((d + 3.0) / 3.0, 0, 1)).astype(np.float32)
                    val[d > 0] = 0
                    if q[7] > 0.8 and T > 12:                   # a hole in the middle of the line
                        val[(np.abs(u) < L / 6) & (np.abs(v) < T / 5)] = 0.05
                    m[ys:ye, xs:xe] = np.maximum(m[ys:ye, xs:xe], val)
                gap = -2 if q[0] > 0.93 else 8 + q[1] * 30       # now and then two lines touch
                x0 += L + gap
            y0 += row_h + 10 + r[qi % 4096, 2] * 8
        if noise > 0:
            m = np.clip(m + (uniform01(h * w, seed * 31 + i).reshape(h, w) - 0.5) * np.float32(noise), 0, 1)
        for t in (0.3, 0.5):
            near = np.abs(m - np.float32(t)) < 2e-3
            m[near] = np.float32(t + 4e-3)
        out[i] = m.astype(np.float32)
    return out


def synth_brightness_detector_state_dict(ref_shapes, use_asf=False, gain=14.0, level=0.45):
    """A hand-made DBNet / DBNet++ ResNet-18 checkpoint (reference key contract, strict) whose probability map is
    sigmoid(gain * (brightness - level)) of the input image, at 1/4 resolution upsampled x4: the stem's centre tap averages the
    de-normalised RGB, the residual blocks of layer1 pass it through (zero conv weights), FPN in2 -> out2 -> head carry channel 0,
    both transposed convs replicate it and the last one applies gain / level.  Every other weight is zero, every BatchNorm is
    the identity.  With random weights a detector's maps are speckle; this one turns images whose BRIGHTNESS is a text-like
    map (synth_scene_images) into text-like probability maps, so run_ocr has real boxes to crop (BASELINE configs[4])."""
    out = {}
    bn_prefixes = {k[: -len(".running_var")] for k in ref_shapes if k.endswith(".running_var")}
    for key, (shape, dtype) in ref_shapes.items():
        prefix = key.rsplit(".", 1)[0]
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros(shape, dtype=np.int64)
        elif prefix in bn_prefixes and (key.endswith(".weight") or key.endswith(".running_var")):
            out[key] = np.ones(shape, np.float32)
        else:
            out[key] = np.zeros(shape, np.float32)
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    out["backbone.conv1.weight"][0, :, 3, 3] = std / np.float32(3.0)
    out["backbone.bn1.bias"][0] = mean.mean()
    out["neck.in2.0.weight"][0, 0, 0, 0] = 1.0
    out["neck.out2.0.weight"][0, 0, 1, 1] = 1.0
    fuse_ch = ref_shapes["head.binarize.0.weight"][0][1] - ref_shapes["neck.out2.0.weight"][0][0]     # p2 is the last slice of the concat
    out["head.binarize.0.weight"][0, fuse_ch, 1, 1] = 1.0
    out["head.binarize.3.weight"][0, 0, :, :] = 1.0
    g = 0.5 if use_asf else 1.0                     # ASF with zero weights scales every feature by sigmoid(0) = 0.5
    out["head.binarize.6.weight"][0, 0, :, :] = np.float32(gain / g)
    out["head.binarize.6.bias"][0] = np.float32(-gain * level)
    return out


def synth_scene_images(n, h, w, seed=0):
    """n synthetic 'document photos' u8[n,h,w,3] (BGR): brightness = 255 x a text-like map (synth_prob_maps: ~150 bright rotated
    bars on black), plus +-4 levels of per-pixel noise.  With synth_brightness_detector_state_dict the detector finds the bars."""
    maps = synth_prob_maps(n, h, w, seed=seed)
    noise = (uniform01(n * h * w * 3, seed ^ 0xC0FFEE).reshape(n, h, w, 3) - np.float32(0.5)) * np.float32(8.0)
    img = maps[..., None] * np.float32(255.0) + noise
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_scene_inputs(n, h, w, seed=0):
    """synth_scene_images as the detector's input tensor f32[n,3,h,w]: BGR -> RGB, /255, ImageNet mean / std
    (reference pytocr/data/imaug/operators.py:41-112), no resize."""
    img = synth_scene_images(n, h, w, seed=seed)[..., ::-1].astype(np.float32) / np.float32(255.0)
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    return np.ascontiguousarray(((img - mean) / std).astype(np.float32).transpose(0, 3, 1, 2))


def synth_scene_state_dict(ref_shapes, readout, gain=14.0, level=0.45, seed=2022):
    """A DBNet / DBNet++ checkpoint (any backbone: the read-out sits in the head) whose probability maps on synth_scene_inputs are
    TEXT-LIKE and cross both post-process thresholds, with every backbone / neck layer carrying its random synth_state_dict weights
    (so the error of all those layers reaches the map, amplified by `gain`).  `readout` is a linear read-out of the 3x3 neighbourhood
    of the neck output (C channels: 96 for MobileNetV3-small, 256 for ResNet-18; row index c*9 + kh*3 + kw, last row the intercept)
    fitted by tools/gen_golden.py on the REFERENCE model's neck features to the scene's text map
    (tests/golden/{mbv3s,r18,detpp}_scene_readout.npz), in one of two forms:
      f32[9C + 1]      one estimate per 1/4-resolution pixel (its 4x4 mean).  binarize.0 channel 0 = +readout, channel 1 = -readout
                       (ReLU passes z as relu(z) - relu(-z)), BN bias -+(intercept - level), both transposed convs replicate channels
                       0 / 1, the last one applies +-gain: map ~ sigmoid(gain * (estimate - level)) in 4x4 blocks.
      f32[9C + 1, 16]  one estimate per FULL-resolution pixel of the 4x4 block, column (2a + a') * 4 + (2b + b') for the pixel
                       (4y + 2a + a', 4x + 2b + b').  Estimate k travels in channels 2k / 2k + 1 (+-); binarize.3 hands it to
                       sub-position (a, b) only and binarize.6 to (a', b') only, so each output pixel gets its own estimate (thin
                       bars stay apart; needs 32 head channels, i.e. the 64-channel head of the ResNet detectors).
    The remaining head channels (22 of 24, resp. 32 of 64) keep their random weights (a few 1e-2 of logit noise)."""
    out = synth_state_dict(ref_shapes, seed)
    c = ref_shapes["head.binarize.0.weight"][0][1]
    mid = ref_shapes["head.binarize.0.weight"][0][0]
    readout = np.asarray(readout, np.float32)
    sub = readout.ndim == 2
    assert readout.shape == ((9 * c + 1, 16) if sub else (9 * c + 1,))
    cols = readout if sub else readout[:, None]
    n = 2 * cols.shape[1]
    assert n <= mid, "the sub-pixel read-out needs %d head channels, this head has %d" % (n, mid)
    for k in range(cols.shape[1]):
        wr = cols[:-1, k].reshape(c, 3, 3)
        out["head.binarize.0.weight"][2 * k] = wr
        out["head.binarize.0.weight"][2 * k + 1] = -wr
        out["head.binarize.1.bias"][2 * k] = cols[-1, k] - np.float32(level)
        out["head.binarize.1.bias"][2 * k + 1] = np.float32(level) - cols[-1, k]
    for bn in ("head.binarize.1", "head.binarize.4"):
        out[bn + ".weight"][:n] = 1.0
        out[bn + ".running_var"][:n] = 1.0
        out[bn + ".running_mean"][:n] = 0.0
    out["head.binarize.4.bias"][:n] = 0.0
    w3 = out["head.binarize.3.weight"]                  # [Cin, Cout, 2, 2]
    w3[:, :n] = 0.0
    w3[:n, :] = 0.0
    out["head.binarize.3.bias"][:n] = 0.0
    w6 = out["head.binarize.6.weight"]                  # [Cin, 1, 2, 2]
    w6[:n] = 0.0
    for k in range(cols.shape[1]):
        dy, dx = (k >> 2, k & 3) if sub else (0, 0)
        for sgn, ch in ((1.0, 2 * k), (-1.0, 2 * k + 1)):
            if sub:
                w3[ch, ch, dy >> 1, dx >> 1] = 1.0
                w6[ch, 0, dy & 1, dx & 1] = np.float32(sgn * gain)
            else:
                w3[ch, ch] = 1.0
                w6[ch, 0] = np.float32(sgn * gain)
    out["head.binarize.6.bias"][:] = 0.0
    return out


synth_mbv3s_scene_state_dict = synth_scene_state_dict


def load_scene_readout(name):
    """(readout, gain, level) of a scene checkpoint from the committed fixture tests/golden/<name>_scene_readout.npz
    (name: mbv3s, r18, detpp); data only -- the fit was made by tools/gen_golden.py on the reference model"""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", name + "_scene_readout.npz")
    r = np.load(path)
    return r["readout"], float(r["gain"]), float(r["level"])
