#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the detection hot path over one batch that is already resident in HBM:
DBNet-r18 fp32 forward (HIP MFMA convs) on f32[32,3,736,1280] -> probability maps -> DB post-process on the
device -> int16 boxes on the host (BASELINE.json configs[1]).  With N > 1 every rank (one process per GPU)
runs the same per-GPU batch on its own images (weak scaling, no data-path collective); the only collective is
the RCCL broadcast of the weights from rank 0 before the timed region.  Rank 0 prints ONE JSON line.

The second half of BASELINE.json's metric (CRNN text-lines/sec, configs[2]: batch 512 of 32x320 crops) is timed in the same
run with the same K / W and travels in the same line under "crnn" (--workload crnn prints it as the top-level line instead;
--workload ocr times configs[4], the run_ocr pipeline).  The default line also carries "mbv3s_bf16" (configs[3]: the bf16
MobileNetV3-small detector, 32 images per GPU) and "ocr" (configs[4], a few steps of the 64-image run_ocr batch), each with its
own roofline and cpu_baseline, so that the driver's one run times every BASELINE config that fits the box.

`value` is wall clock over exactly K steps (barrier + synchronize on both sides, max over ranks); `ms_per_step_median` is the
median over the K steps of HIP-event step times on the launch stream.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide, dense bf16 matrix peak
PEAK_HBM_GBPS = 8000.0             # same guide, HBM3E spec peak (6.3 TB/s measured achievable)
WINOGRAD_GAIN = 2.25               # F(2x2,3x3): 16 multiplies instead of 36 per 2x2 output tile and channel pair
WINOGRAD4_GAIN = 4.0               # F(4x4,3x3): 36 multiplies instead of 144 per 4x4 output tile and channel pair
DET_TAIL_GFLOP_PER_IMG = 2.05      # ConvT 64->64 (1.93) + ConvT 64->1 (0.12) run in db_head_tail_kernel, not in the conv kernels
CRNN_GFLOP_PER_LINE = 4.980
POST_BYTES_PER_PIXEL = 18          # SURVEY 8d accounting of the DB post-process

DET_R18 = dict(model_type="det", algorithm="DB", Transform=None,
               Backbone=dict(name="ResNet", layers=18, pretrained=False),
               Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False, attention_type="scale_channel_spatial"),
               Head=dict(name="DBHead", k=50))
# the other detectors of SURVEY 8a (--det-model); the headline metric is DBNet-r18
DET_VARIANTS = {                                       # config, state_dict contract, GFLOP per image, scene read-out fixture
    "r18": (DET_R18, "det_r18_db", 114.195, "r18"),
    "r18pp": (dict(DET_R18, Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=True, attention_type="scale_channel_spatial")),
              "detpp_r18_db", 131.59, "detpp"),
    "mbv3s": (dict(model_type="det", algorithm="DB", Transform=None,
                   Backbone=dict(name="MobileNetV3", model_name="small", scale=1.0, pretrained=False),
                   Neck=dict(name="FPN", out_channels=96, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50)),
              "det_mbv3s_db", 8.43, "mbv3s"),
}
DET_POST = dict(name="DBPostProcess", thresh=0.3, box_thresh=0.5, max_candidates=1000, unclip_ratio=1.7,
                score_mode="poly", cpp_speedup=True, out_polygon=False)


def crnn_cfg(nclass=6624):
    return dict(model_type="rec", algorithm="CRNN", in_channels=1, Transform=None,
                Backbone=dict(name="VGG", model_name="v1", scale=1.0, pretrained=False, ckpt_path=None),
                Neck=dict(name="SequenceEncoder", encoder_type="rnn", hidden_size=256),
                Head=dict(name="CTCHead", out_channels=nclass))


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local, world


def load_contract(name):
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_contract.json")) as f:
        c = json.load(f)[name]
    return {k: (tuple(s), d) for k, (s, d) in c.items()}


def median(v):
    v = sorted(v)
    n = len(v)
    return 0.0 if n == 0 else (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]))


def self_launch(argv, n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD `python -m torch.distributed.run`
    (this parent never touches HIP -- a process that initialised the GPU must not exec, and need not), relay rank 0's JSON
    line and exit with the child's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    # (no OMP_NUM_THREADS here: round 3 pinned 4, which would have throttled the numpy / MKL parts of rank 0's CPU baseline on an
    # 8-rank run; the baselines set their own torch thread count and report it as `cores`)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in p.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        rc = 1
    raise SystemExit(rc)


def det_state_dict(contract_name, scene):
    """Random-init weights of the named architecture (utils/synth.py, seeded).  With `scene` (the default) two of the head's channels
    carry a linear read-out of the neck features fitted to the scene images' text map, and the last layer a logit gain of 14
    (synth_scene_state_dict; the fit is data under tests/golden): every backbone / neck convolution still runs on its dense random
    weights, but the probability maps are text-like -- ~140 boxes per image -- instead of the speckle a fully random head gives."""
    from pytorchocr_amd.utils.synth import load_scene_readout, synth_scene_state_dict, synth_state_dict
    if scene:
        return synth_scene_state_dict(load_contract(contract_name), *load_scene_readout(scene))
    return synth_state_dict(load_contract(contract_name))


def build_and_sync_weights(cfg, contract_name, device, rank, world, scene=None):
    """Synthetic weights of the named architecture: rank 0 makes them, RCCL broadcast puts them on every GPU."""
    import torch
    from pytorchocr_amd.modeling.architectures import build_model
    model = build_model(cfg)
    if rank == 0:
        sd = det_state_dict(contract_name, scene)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(device).eval()
    if world > 1:
        from pytorchocr_amd.parallel import broadcast_model_
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        broadcast_model_(model, src=0)
        torch.cuda.synchronize()
        BROADCAST_MS.append(round((time.perf_counter() - t0) * 1e3, 3))
    return model


def det_cpu_baseline(n_img, H, W, det_model="r18", scene=True, tags=("model",)):
    """The oracle (CPU restatement of the reference path: torch-CPU fp32 forward + C post-process with cpp_speedup=True semantics)
    on a bounded sample of the same workload -- same checkpoint, same kind of input, the same maps post-processed as in the GPU step.
    With the scene checkpoint this is BASELINE configs[0] as SURVEY 8(d) words it: u8 BGR SOURCE images, half of them 720 x 1280 (the host
    DetResizeForTest takes them to 736 x 1312), half 736 x 1280, each through DetResizeForTest + ToTensor + Normalize (the host operators
    of pytorchocr_amd/data/imaug.py; reference operators.py:41-112,155-252), batch 1, forward, post-process against the source size
    (infer_det.py:85-103).  The GPU pre-process of the same sources is timed beside it (`gpu_preprocess_ms_per_image`).
    mbv3s: the reference has no reduced-precision mode, so configs[3] stands beside the fp32 forward."""
    import numpy as np
    import torch
    from oracle import dbpost, model_oracle
    from pytorchocr_amd.data.imaug import DetResizeForTest, Normalize, ToTensor
    from pytorchocr_amd.utils.synth import synth_images, synth_prob_maps, synth_scene_images
    torch.set_num_threads(min(16, os.cpu_count() or 1))            # the GPU box's CPU share for one GPU is 16 cores
    _, contract_name, _, scene_name = DET_VARIANTS[det_model]
    sd = {k: torch.from_numpy(v) for k, v in det_state_dict(contract_name, scene_name if scene else None).items()}
    fwd = model_oracle.dbnet_forward
    stress = synth_prob_maps(1, H, W, seed=2022)[0]
    fwd(sd, torch.zeros(1, 3, 64, 64))                             # warm-up of the thread pool
    if scene:
        srcs = [synth_scene_images(1, 720 if i % 2 == 0 else H, W, seed=2022 + i)[0] for i in range(n_img)]
    else:
        srcs = [np.clip(np.rint((synth_images(1, 3, H, W, seed=2022 + i)[0].transpose(1, 2, 0) * 0.25 + 0.5) * 255), 0, 255).astype(np.uint8) for i in range(n_img)]
    resize, to_tensor = DetResizeForTest(limit_side_len=736, limit_type="min"), ToTensor()
    norm = Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    t_pre = t_model = t_post = 0.0
    nbox = 0
    sizes = set()
    for img in srcs:                                               # batch 1 each, as infer_det.py:85-103 does
        t0 = time.perf_counter()
        d = resize({"image": np.ascontiguousarray(img[:, :, ::-1])})              # (BGR -> RGB as DecodeImage does)
        shape = d["shape"]
        x = norm(to_tensor(d))["image"]
        x = (x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x)))[None].float()
        sizes.add("%dx%d->%dx%d" % (img.shape[0], img.shape[1], x.shape[2], x.shape[3]))
        t1 = time.perf_counter()
        maps = fwd(sd, x)["maps"].numpy()
        t2 = time.perf_counter()
        for t in tags:                                             # the net's own map and / or the text-like stress map
            m = maps[0, 0] if t == "model" else stress
            nbox += len(dbpost.boxes_from_bitmap(m, dbpost.binarize(m, 0.3), 0.5, 1.7, int(shape[1]) if t == "model" else W,
                                                 int(shape[0]) if t == "model" else H))
        t3 = time.perf_counter()
        t_pre += t1 - t0
        t_model += t2 - t1
        t_post += t3 - t2
    total = t_pre + t_model + t_post
    gpu_pre = None
    if torch.cuda.is_available():                                  # the same sources through the fused GPU pre-process (u8 in HBM -> network input)
        from pytorchocr_amd.data.gpu_preprocess import det_preprocess_batch
        dev = torch.device("cuda", torch.cuda.current_device())
        gsrc = [torch.from_numpy(np.ascontiguousarray(i)).to(dev)[None] for i in srcs]
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for g in gsrc:
                det_preprocess_batch(g, resize.target_size(int(g.shape[1]), int(g.shape[2])), norm.mean, norm.std, swap_rb=True)
            e1.record()
            torch.cuda.synchronize()
            gpu_pre = e0.elapsed_time(e1) / len(gsrc)
    return {"value": round(n_img / total, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "gpu_preprocess_ms_per_image": None if gpu_pre is None else round(gpu_pre, 4),
            "sample": "%d source images (%s), batch 1 each: host DetResizeForTest + ToTensor + Normalize %.4f s/img + torch-CPU fp32 %s forward "
                      "%.3f s/img + C post-process of the %s map%s %.4f s/img (%d boxes/img; single thread, like the GIL-bound reference extension)"
                      % (n_img, ", ".join(sorted(sizes)), t_pre / n_img, det_model, t_model / n_img, " + ".join(tags), "s" if len(tags) > 1 else "",
                         t_post / n_img, nbox // max(n_img, 1))}


def crnn_cpu_baseline(n_lines):
    """The oracle (torch-CPU fp32 CRNN forward + the reference's CTCLabelDecode restatement) on a bounded sample of lines,
    one batch of 16 at a time."""
    import torch
    from oracle import ctc_oracle, model_oracle
    from pytorchocr_amd.utils.synth import synth_state_dict, synth_text_lines
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(load_contract("rec_vgg_bilstm_ctc")).items()}
    chars = ctc_oracle.load_characters(os.path.join(ROOT, "pytorchocr_amd", "utils", "char_dict_6623.txt"))
    x = torch.from_numpy(synth_text_lines(16, 32, 320, seed=2022))
    model_oracle.crnn_forward(sd, x[:2])                               # warm-up of the thread pool
    t0 = time.perf_counter()
    done = 0
    while done < n_lines:
        probs = model_oracle.crnn_forward(sd, x).numpy()
        ctc_oracle.ctc_label_decode(probs, chars)
        done += 16
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 3), "unit": "text-lines/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d lines 32x320 in batches of 16: torch-CPU fp32 forward + softmax + CTCLabelDecode restatement, %.1f ms/line"
                      % (done, dt / done * 1e3)}


def parallelism(what, world):
    """config.parallelism: how the work is split, the torch.distributed backend actually in use and its world size"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        be = "%s (%s), dist world %d" % (dist.get_backend(), "RCCL over xGMI" if dist.get_backend() == "nccl" else "rehearsal backend",
                                         dist.get_world_size())
    else:
        be = "single process, no process group"
    return "%s x%d; %s; weight broadcast from rank 0 is the only collective (%s)" % (
        what, world, be, "ms per broadcast on this rank, first one includes the communicator's start-up: %s" % BROADCAST_MS if BROADCAST_MS
        else "none in a single-process run")


def ocr_cpu_baseline(n_img):
    """configs[4] beside the CPU, image by image and box by box as run_ocr.py:167-231 does: host resize + normalise, the oracle's
    DBNet++ r18 forward, the C post-process, sort_boxes, per box perspective crop -> gray -> resize/pad -> batch-1 CRNN forward ->
    CTC decode.  (Resize and crops use the product's vectorised HOST operators -- the reference's are OpenCV calls; the oracle's
    per-pixel Python restatements of them are checkers, far too slow to stand for a CPU path.)"""
    import numpy as np
    import torch
    from oracle import ctc_oracle, dbpost, model_oracle
    from pytorchocr_amd.data.imaug import bgr_to_gray, resize_bilinear
    from pytorchocr_amd.utils.synth import synth_scene_images, synth_state_dict
    from pytorchocr_amd.utils.utility import sort_boxes
    from pytorchocr_amd.utils.warp import get_part_img
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    det_sd = {k: torch.from_numpy(v) for k, v in det_state_dict("detpp_r18_db", "detpp").items()}        # the checkpoint make_ocrer loads
    rec_sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(load_contract("rec_vgg_bilstm_ctc")).items()}
    chars = ctc_oracle.load_characters(os.path.join(ROOT, "pytorchocr_amd", "utils", "char_dict_6623.txt"))
    imgs = synth_scene_images(n_img, 960, 1280, seed=100)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    model_oracle.crnn_forward(rec_sd, torch.zeros(1, 1, 32, 320))
    lines = 0
    t0 = time.perf_counter()
    for img in imgs:
        rs = resize_bilinear(img[:, :, ::-1], (992, 736)).astype(np.float32) / 255.0
        x = torch.from_numpy(np.ascontiguousarray(((rs - mean) / std).transpose(2, 0, 1)[None], np.float32))
        pm = model_oracle.dbnet_r18_forward(det_sd, x)["maps"].numpy()[0, 0]
        boxes = dbpost.boxes_from_bitmap(pm, dbpost.binarize(pm, 0.3), 0.5, 1.7, 1280, 960)
        boxes = sort_boxes(boxes.astype(np.int16)) if len(boxes) else []
        for box in boxes:
            part = get_part_img(img, np.asarray(box))
            if part.shape[0] < 2 or part.shape[1] < 2:
                continue
            if part.shape[0] >= 1.5 * part.shape[1]:
                part = np.rot90(part, 1)
            g = bgr_to_gray(np.ascontiguousarray(part))
            rw = min(320, int(np.ceil(32 * g.shape[1] / float(g.shape[0]))))
            r = resize_bilinear(g, (max(rw, 1), 32)).astype(np.float32) / 255.0
            pad = np.zeros((1, 1, 32, 320), np.float32)
            pad[0, 0, :, :r.shape[1]] = (r - 0.5) / 0.5
            ctc_oracle.ctc_label_decode(model_oracle.crnn_forward(rec_sd, torch.from_numpy(pad)).numpy(), chars)
            lines += 1
    dt = time.perf_counter() - t0
    return {"value": round(n_img / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d source images 1280x960 (%d text lines), image by image, batch-1 CRNN per box as run_ocr.py:187-229: torch-CPU fp32 "
                      "DB++ r18 + C post-process + host crops + torch-CPU CRNN + CTC decode, %.2f s/img" % (n_img, lines, dt / max(n_img, 1))}


def lstm_stats():
    """(split-form LSTM calls, calls the on-stream repair pass had to recompute, calls whose exchange went through one XCD's L2) since the library was loaded"""
    import ctypes as C
    from pytorchocr_amd import _lib
    a, b = C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().ptocr_lstm_stats(C.byref(a), C.byref(b)), "ptocr_lstm_stats")
    c = C.c_int(0)
    _lib.check(_lib.lib().ptocr_lstm_same_xcd_calls(C.byref(c)), "ptocr_lstm_same_xcd_calls")
    return a.value, b.value, c.value


def _per_rank(value, world, device):
    """every rank's own figure, gathered to rank 0 (so that an N-rank line can be checked rank by rank)"""
    if world == 1:
        return [round(value, 3)]
    import torch
    import torch.distributed as dist
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [round(float(v), 3) for v in t.tolist()]


BROADCAST_MS = []            # wall time of every weight broadcast of this process (build_and_sync_weights)


def _sync_all(world):
    import torch
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()


def _max_over_ranks(dt, world, device):
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def _conv_profile(prof, labels):
    """HIP-event durations of the conv launches of the timed region, split into Winograd launches and the rest."""
    ms = [e0.elapsed_time(e1) for e0, e1 in prof]
    wino_ms, n_wino = 0.0, 0
    wino_flops = {"alg": 0.0, "exec": 0.0, "n4": 0, "ms4": 0.0}      # direct-convolution FLOPs and the FLOPs the MFMAs executed
    for lab, t in zip(labels, ms):
        if lab.startswith("wino3x3") or lab.startswith("wino43x3"):
            n_, h_, w_, ci = [int(v) for v in lab.split()[1].split("->")[0].split("x")]
            co = int(lab.split("->")[1].split()[0])
            four = lab.startswith("wino43x3")
            wino_ms += t
            n_wino += 1
            f = 2.0 * n_ * h_ * w_ * ci * co * 9
            wino_flops["alg"] += f
            wino_flops["exec"] += f / (WINOGRAD4_GAIN if four else WINOGRAD_GAIN)
            wino_flops["n4"] += int(four)
            wino_flops["ms4"] += t if four else 0.0
    return sum(ms), len(ms), wino_ms, wino_flops, n_wino


PROFILE_PASS_STEPS = 10                     # steps of the untimed per-launch-event pass behind `roofline`


def _profile_pass_note(prof_steps, dt_prof, timed_step_s):
    """where the per-launch durations of `roofline` come from, and what the event records cost: the same step with and without them"""
    return {"pass": "separate pass of %d steps of the same step right after the timed region (ops.PROFILE on: two hipEventRecords around "
                    "every conv launch); the timed region carries no per-launch events" % prof_steps,
            "ms_per_step_with_launch_events": round(dt_prof / prof_steps * 1e3, 3),
            "ms_per_step_timed_region": round(timed_step_s * 1e3, 3)}


def _profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return None


def _wino_roofline(wino_ms, wino_flops, n_wino, steps, what):
    """`achieved` / `frac` = what the matrix pipe EXECUTED (Winograd F(4x4,3x3) runs 1/4, F(2x2,3x3) 1/2.25 of the
    direct-convolution multiplies): a utilisation, <= 1.  The algorithmic (direct-convolution) rate of the same
    launches is reported beside it as `algorithmic_tflops` -- it can exceed the peak because of the algorithmic gain, and is
    not a fraction."""
    sec = wino_ms * 1e-3
    alg = wino_flops["alg"] / sec / 1e12 if wino_ms > 0 else 0.0
    ex = wino_flops["exec"] / sec / 1e12 if wino_ms > 0 else 0.0
    traffic = (_profile_json("conv_traffic.json") or {}).get("hbm_bytes_per_launch")
    n4 = wino_flops["n4"]
    return {"bound": "mfma", "achieved": round(ex, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ex / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
            "traffic_source": "profiles/conv_traffic.json / crnn_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                              "(2 * FETCH_SIZE + WRITE_SIZE per launch); copied from the committed profile, NOT measured in this run",
            "kernel": "conv_wino4r_kernel F(4x4,3x3; PTOCR_WINO4R=0: conv_wino4_kernel) + conv_wino_kernel F(2x2,3x3) (%s; %d + %d launches per step, %.3f ms avg launch, HIP "
                      "events on the launch stream); achieved = executed MFMA FLOPs (direct-convolution FLOPs / 4 resp. / 2.25) of "
                      "those launches / their time" % (what, n4 // max(steps, 1), (n_wino - n4) // max(steps, 1), wino_ms / max(n_wino, 1)),
            "algorithmic_tflops": round(alg, 2), "algorithmic_over_peak": round(alg / PEAK_F32_MFMA_TFLOPS, 4)}


def _ocr_roofline(prof, labels, steps):
    """run_ocr's dominant kernel is the same Winograd kernel (DB++ r18 and the CRNN's VGG stack): executed MFMA FLOPs of its launches
    over their HIP-event durations, like the det line"""
    conv_ms, n_launch, wino_ms, wino_flops, n_wino = _conv_profile(prof, labels)
    roof = _wino_roofline(wino_ms, wino_flops, n_wino, steps, "3x3/s1 layers of DBNet++ r18 and of the CRNN's VGG stack")
    roof["traffic"], roof["traffic_source"] = None, None
    roof["all_conv"] = {"launches_per_step": n_launch // max(steps, 1), "ms_per_step": round(conv_ms / max(steps, 1), 3)}
    return roof


def run_det(args, rank, local, world, device):
    import numpy as np
    import torch
    from pytorchocr_amd.modeling import ops
    from pytorchocr_amd.postprocess import build_post_process
    from pytorchocr_amd.utils.synth import synth_images, synth_prob_maps, synth_scene_inputs
    B, H, W = args.batch or 32, 736, 1280
    det_cfg, det_contract, gflop_img, scene_name = DET_VARIANTS[args.det_model]
    scene = args.weights == "scene"
    model = build_and_sync_weights(det_cfg, det_contract, device, rank, world, scene=scene_name if scene else None)
    bf16 = args.dtype == "bf16"
    if bf16:
        model.set_compute_dtype("bf16")                  # BASELINE configs[3]: bf16 activations / weights, fp32 accumulation and maps
    post = build_post_process(DET_POST, dict(use_gpu=True, seed=2022))
    nd = min(args.distinct_images, B)
    # nd distinct seeded images, tiled to the batch: text-like scenes for the scene checkpoint (its maps then hold ~140 boxes per
    # image and the step is the reference pipeline as it is: forward, post-process of the net's own maps), uniform noise otherwise
    base = synth_scene_inputs(nd, H, W, seed=2022 + rank) if scene else synth_images(nd, 3, H, W, seed=2022 + rank)
    x = torch.from_numpy(base).to(device).repeat(B // nd + 1, 1, 1, 1)[:B].contiguous()
    shape_list = np.array([[H, W, 1.0, 1.0]] * B)
    # --weights random gives speckle maps; so that the post-process does realistic work there, text-like stress maps are
    # post-processed inside the timed step as a SECOND pass (--post-input both, that mode's default)
    post_input = args.post_input or ("model" if scene else "both")
    tags = [t for t in ("model", "stress") if post_input in (t, "both")]
    stress = None
    if "stress" in tags:
        stress = torch.from_numpy(synth_prob_maps(nd, H, W, seed=7 + rank)).to(device).repeat(B // nd + 1, 1, 1)[:B, None].contiguous()

    def step(pending):
        """forward of this batch; its post-process is queued on the post-process stream and collected one step later,
        so it overlaps with the next batch's convolutions (every batch's boxes are on the host before the clock stops)"""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        with torch.no_grad():
            out = model(x)
        last["maps"] = out["maps"]
        if fwd_events is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            fwd_events.append((ev, e1))
        futs = []
        if "model" in tags:                              # the pipeline's own data flow
            futs.append(post.submit({"maps": out["maps"]}, shape_list))
        if "stress" in tags:                             # realistic ~140 boxes/image load
            futs.append(post.submit({"maps": stress}, shape_list))
        if not args.overlap:
            return [], [f.result() for f in futs], ev
        return futs, [f.result() for f in pending], ev

    fwd_events = None
    last = {}
    pending = []
    for _ in range(args.warmup):
        pending, _, _ = step(pending)
    for f in pending:
        f.result()
    pending = []
    _sync_all(world)
    fwd_bytes = None
    if bf16 and rank == 0:                               # bytes one forward moves, counted by the launch wrappers (one untimed pass)
        from pytorchocr_amd.modeling import bf16_path
        from pytorchocr_amd import _lib as _plib
        bf16_path.TRAFFIC = 0
        calls0 = _plib.CALLS
        with torch.no_grad():
            model(x)
        fwd_bytes, bf16_path.TRAFFIC = bf16_path.TRAFFIC, None
        fwd_launches = _plib.CALLS - calls0
        torch.cuda.synchronize()
        fwd_events = []
    # The timed region carries NO per-launch instrumentation (ops.PROFILE stays None: round 5 bracketed every conv launch of the timed
    # steps with two hipEventRecords, 58 per step).  The per-launch HIP events behind `roofline` come from a SEPARATE pass of the same
    # step, run right after the timed region (same batch, same overlap with the post-process stream) and reported with its own wall time.
    post.device_ms_log = []
    nbox = {t: 0 for t in tags}
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pending, res, ev = step(pending)
        events.append(ev)
        for t, r in zip(tags, res):
            nbox[t] += sum(len(i["points"]) for i in r)
    for t, f in zip(tags, pending):                      # drain: the last batch's boxes
        nbox[t] += sum(len(i["points"]) for i in f.result())
    ev_end = torch.cuda.Event(enable_timing=True)
    ev_end.record()
    _sync_all(world)
    dt = time.perf_counter() - t0
    post_ms = post.device_ms_log
    post.device_ms_log = None
    prof_steps = max(1, min(args.steps, PROFILE_PASS_STEPS))
    ops.PROFILE = [] if rank == 0 else None
    ops.PROFILE_LABELS = [] if rank == 0 else None
    pending = []
    tp0 = time.perf_counter()
    for _ in range(prof_steps):
        pending, _, _ = step(pending)
    for f in pending:
        f.result()
    torch.cuda.synchronize()
    dt_prof = time.perf_counter() - tp0
    prof, labels = ops.PROFILE, ops.PROFILE_LABELS
    ops.PROFILE = ops.PROFILE_LABELS = None
    _sync_all(world)
    per_rank = _per_rank(B * args.steps / dt, world, device)
    dt = _max_over_ranks(dt, world, device)
    if rank != 0:
        return None
    events.append(ev_end)
    step_ms = [events[i].elapsed_time(events[i + 1]) for i in range(len(events) - 1)]
    conv_ms, n_launch, wino_ms, wino_flops, n_wino = _conv_profile(prof, labels)
    conv_flops = (gflop_img - DET_TAIL_GFLOP_PER_IMG) * 1e9 * B * prof_steps
    # post-process stage: device time of every call (HIP events on the post-process stream, inside the timed region, i.e. while
    # the next batch's convolutions share the chip), and the same call alone on an idle chip
    by_tag = {t: post_ms[i::len(tags)] for i, t in enumerate(tags)} if tags else {}
    # ... alone: on the text-like stress maps (the input rounds 1-2 quoted this stage on: clean bars, ~136 boxes per image) and on the
    # last batch's own maps (the scene checkpoint's maps have ragged edges -- the read-out's residual -- and cost more)
    alone = {}
    if tags:
        if stress is None:
            stress = torch.from_numpy(synth_prob_maps(nd, H, W, seed=7 + rank)).to(device).repeat(B // nd + 1, 1, 1)[:B, None].contiguous()
        for t, src in (("stress", stress), ("model", last["maps"])):
            if t == "model" and "model" not in tags:
                continue
            post.device_ms_log = []
            for _ in range(20):                                      # the workspace keeps its labelling route for eight calls after a change of input
                post({"maps": src}, shape_list)
            alone[t] = median(post.device_ms_log[10:])
            post.device_ms_log = None
    post_bytes = float(POST_BYTES_PER_PIXEL) * H * W * B
    roofline_post = None
    if tags:
        # `frac` / `ms_per_call_alone` are quoted on the maps the TIMED STEP post-processes (its own maps with the scene checkpoint, the
        # default); the text-like stress maps -- the input rounds 1-3 quoted this stage on -- stand beside them as *_stress_maps
        own = "model" if "model" in alone else "stress"
        best = alone[own]
        fr = lambda ms: round(post_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
        roofline_post = {
            "bound": "hbm", "achieved": round(post_bytes / (best * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": fr(best),
            "input": "the timed step's own maps" if own == "model" else "text-like stress maps (the step's only post-process input)",
            "traffic": (_profile_json("post_traffic.json") or {}).get("hbm_bytes_per_call"),
            "traffic_source": "profiles/post_traffic.json (rocprofv3 --pmc passes of tools/bench_post.py; per_kernel too): copied from the "
                              "committed profile, NOT measured in this run",
            "stage": "DB post-process of %d maps %dx%d: %d B/pixel accounting (SURVEY 8d) = %.1f MB per call / median device time of the "
                     "call's kernels alone on the chip (HIP events on its stream)" % (B, H, W, POST_BYTES_PER_PIXEL, post_bytes / 1e6),
            "ms_per_call_alone": round(best, 4),
            "ms_per_call_alone_stress_maps": round(alone["stress"], 4),
            "frac_stress_maps": fr(alone["stress"]),
            "ms_per_call_overlapped": {t: round(median(v), 4) for t, v in by_tag.items()},
            "per_kernel": (_profile_json("post_traffic.json") or {}).get("per_kernel"),
        }
    # rank 0 only, after the timed region (the other ranks wait at the next barrier); with several ranks on the host the sample is halved
    n_cpu = args.cpu_images if world == 1 else max(1, args.cpu_images // 2)
    cpu = None
    if args.cpu_images > 0:
        cpu = det_cpu_baseline(n_cpu, H, W, args.det_model, scene, tuple(tags) or ("model",))
    if bf16:
        fms = median([a.elapsed_time(b) for a, b in fwd_events])
        gbps = fwd_bytes / (fms * 1e-3) / 1e9
        # the same forward with NOTHING else on the chip (a pass of its own after the timed region: no post-process stream beside it) --
        # `frac` above prices the forward as the timed step runs it, with the previous batch's post-process taking bandwidth from it
        alone_ev = []
        with torch.no_grad():
            for _ in range(30):
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                model(x)
                a1.record()
                alone_ev.append((a0, a1))
        torch.cuda.synchronize()
        fms_alone = median([a.elapsed_time(b) for a, b in alone_ev[10:]])
        roof = {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
                "traffic": (_profile_json("mbv3s_bf16_traffic.json") or {}).get("hbm_bytes_per_forward"),
                "traffic_source": "profiles/mbv3s_bf16_traffic.json (rocprofv3 --pmc passes of this command with --det-model mbv3s --dtype bf16): "
                                  "copied from the committed profile, NOT measured in this run",
                "launches_per_forward": fwd_launches,
                "forward_alone": {"ms": round(fms_alone, 4), "achieved": round(fwd_bytes / (fms_alone * 1e-3) / 1e9, 1),
                                  "frac": round(fwd_bytes / (fms_alone * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                                  "pass": "20 forwards back to back after the timed region, no post-process in flight (HIP events on the launch stream)"},
                "kernel": "whole bf16 forward (stem, 1x1 MFMA convs, expansion + depthwise, depthwise + SE pool, SE gate, 3x3 MFMA convs with the lateral / four-plane "
                          "gather inside, head tail): "
                          "bytes every launch reads + writes once (activations, weights; %.1f MB per forward of %d images) / median forward time "
                          "%.3f ms (HIP events on the launch stream, post-process of the previous batch overlapping)" % (fwd_bytes / 1e6, B, fms),
                # fusing two launches removes their intermediate from `achieved`'s numerator as well as its time from the denominator: the
                # same forward priced at the bytes of the round-3 launch graph (6.26 GB per 32 images: every layer's input + output + weights
                # once) stands beside it, so that rounds compare
                "frac_at_round3_launch_graph_bytes": round(6.26e9 * (B / 32.0) / (fms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                "mfma_tflops": round(gflop_img * 1e9 * B / (fms * 1e-3) / 1e12, 2),
                "mfma_frac_of_bf16_peak": round(gflop_img * 1e9 * B / (fms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
    else:
        roof = _wino_roofline(wino_ms, wino_flops, n_wino, prof_steps, "every 3x3/s1 layer")
        roof["measured_in"] = _profile_pass_note(prof_steps, dt_prof, dt / args.steps)
    if not bf16:
        roof["all_conv"] = {"launches_per_step": n_launch // max(prof_steps, 1), "ms_per_step": round(conv_ms / max(prof_steps, 1), 3),
                            "algorithmic_tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2) if conv_ms > 0 else 0.0,
                            "kernels": "conv_wino4r_kernel / conv_wino_kernel (3x3 s1) + stem_conv_kernel (7x7 s2) + conv_pw64_kernel (FPN in2) + conv_mfma_v2_kernel (3x3 s2, other 1x1)"}
    return {
        "metric": "images/sec end-to-end (DBNet-r18 det+post, 736x1280)" if args.det_model == "r18"
                  else "images/sec end-to-end (%s det+post, 736x1280; NOT the BASELINE metric)" % args.det_model,
        "value": round(world * B * args.steps / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_step_median": round(median(step_ms), 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": "DBNet %s %s, batch %d synthetic 736x1280 per GPU (%d distinct images tiled), HIP conv + HIP DBPostProcess "
                               "(BASELINE.json %s)" % (args.det_model, "bf16" if bf16 else "fp32", B, nd,
                                                       "configs[3]: batch 256 = 32 per GPU x 8" if bf16 else "configs[1]"),
                   "weights": ("scene checkpoint: seeded random-init weights in every backbone / neck layer and in %s of the head's channels; the other "
                               "%s carry a linear read-out of the neck features (%s) fitted to the synthetic scenes' text map and a logit gain of 14, so the "
                               "net's own maps are text-like and the step is the reference pipeline as it is (forward, post-process of its maps)"
                               % (("22 of 24", "2", "one estimate per 4x4 block") if args.det_model == "mbv3s" else
                                  ("32 of 64", "32", "16 estimates, one per pixel of the 4x4 output block")))
                              if scene else "seeded random-init weights in every layer (the maps are speckle)",
                   "global_batch": world * B, "post_input": post_input, "post_overlap": bool(args.overlap),
                   "boxes_per_image": {t: round(v / (B * args.steps), 1) for t, v in nbox.items()},
                   "parallelism": parallelism("image-sharded", world), "per_rank_images_per_sec": per_rank},
        "roofline": roof, "roofline_post": roofline_post, "cpu_baseline": cpu,
    }


def run_crnn(args, rank, local, world, device):
    import torch
    from pytorchocr_amd.modeling import ops
    from pytorchocr_amd.postprocess import build_post_process
    from pytorchocr_amd.utils.synth import synth_text_lines
    B = args.crnn_batch
    model = build_and_sync_weights(crnn_cfg(), "rec_vgg_bilstm_ctc", device, rank, world)
    post = build_post_process(dict(name="CTCLabelDecode"), dict(character_dict_path=os.path.join(
        ROOT, "pytorchocr_amd", "utils", "char_dict_6623.txt"), use_space_char=False))
    nd = min(max(args.distinct_images, 16), B)
    base = synth_text_lines(nd, 32, 320, seed=2022 + rank)
    x = torch.from_numpy(base).to(device).repeat(B // nd + 1, 1, 1, 1)[:B].contiguous()

    def step(pending):
        """forward + CTC arg-max of this batch on the GPU; its label ids / confidences travel to pinned host memory behind it
        and are decoded to text (host Python, as in the reference) one step later, i.e. while the next batch computes.  Every
        batch's texts are on the host before the clock stops."""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        with torch.no_grad():
            if not args.overlap:
                return None, post(model.forward_greedy(x)), ev
            fut = post.submit(model.forward_greedy(x))
            res = pending.result() if pending is not None else None
        return fut, res, ev

    pending = None
    for _ in range(args.warmup):
        pending, _, _ = step(pending)
    if pending is not None:
        pending.result()
    pending = None
    _sync_all(world)
    lstm0 = lstm_stats()
    nchar = 0
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pending, res, ev = step(pending)
        events.append(ev)
        nchar += sum(len(t) for t, _ in res) if res else 0
    if pending is not None:
        nchar += sum(len(t) for t, _ in pending.result())    # drain: the last batch's texts
    ev_end = torch.cuda.Event(enable_timing=True)
    ev_end.record()
    _sync_all(world)
    dt = time.perf_counter() - t0
    lstm1 = lstm_stats()
    # per-launch HIP events: a separate pass of the same step, outside the timed region (see run_det)
    prof_steps = max(1, min(args.steps, PROFILE_PASS_STEPS))
    ops.PROFILE = [] if rank == 0 else None
    ops.PROFILE_LABELS = [] if rank == 0 else None
    pending = None
    tp0 = time.perf_counter()
    for _ in range(prof_steps):
        pending, _, _ = step(pending)
    if pending is not None:
        pending.result()
    torch.cuda.synchronize()
    dt_prof = time.perf_counter() - tp0
    prof, labels = ops.PROFILE, ops.PROFILE_LABELS
    ops.PROFILE = ops.PROFILE_LABELS = None
    _sync_all(world)
    per_rank = _per_rank(B * args.steps / dt, world, device)
    dt = _max_over_ranks(dt, world, device)
    if rank != 0:
        return None
    events.append(ev_end)
    step_ms = [events[i].elapsed_time(events[i + 1]) for i in range(len(events) - 1)]
    conv_ms, n_launch, wino_ms, wino_flops, n_wino = _conv_profile(prof, labels)
    roof = _wino_roofline(wino_ms, wino_flops, n_wino, prof_steps, "VGG conv1..conv5")
    roof["measured_in"] = _profile_pass_note(prof_steps, dt_prof, dt / args.steps)
    roof["traffic"] = (_profile_json("crnn_traffic.json") or {}).get("hbm_bytes_per_launch")
    roof["whole_step_algorithmic_tflops"] = round(CRNN_GFLOP_PER_LINE * 1e9 * B * args.steps / dt / 1e12, 2)
    roof["all_conv"] = {"launches_per_step": n_launch // max(prof_steps, 1), "ms_per_step": round(conv_ms / max(prof_steps, 1), 3)}
    return {
        "metric": "CRNN text-lines/sec (vgg_v1_x1.0 + BiLSTM + CTC greedy, 32x320)",
        "value": round(world * B * args.steps / dt, 2), "unit": "text-lines/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_step_median": round(median(step_ms), 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "CRNN vgg_v1_x1.0 + CTC greedy, batch %d synthetic 32x320 gray crops per GPU (BASELINE.json configs[2])" % B,
                   "global_batch": world * B, "decode_overlap": bool(args.overlap), "chars_per_line": round(nchar / (B * args.steps), 1),
                   "parallelism": parallelism("line-sharded", world), "per_rank_lines_per_sec": per_rank},
        # the split-form LSTM needs its four workgroups per line group co-resident; a call that timed out is recomputed on the stream by the
        # repair launch (correct, but a whole second layer): a silent 2x would show here
        "lstm": {"split_calls": lstm1[0] - lstm0[0], "repaired": lstm1[1] - lstm0[1], "same_xcd_calls": lstm1[2] - lstm0[2]},
        "roofline": roof,
        "cpu_baseline": crnn_cpu_baseline(args.cpu_lines if world == 1 else max(16, args.cpu_lines // 2)) if args.cpu_lines > 0 else None,
    }


def run_dry(args, rank, world):
    """PTOCR_BENCH_DRY=1: control-flow rehearsal without a GPU (tests/test_bench_launch.py): rendezvous over gloo, barrier,
    K timed sleeps, max over ranks, one JSON line from rank 0.  Nothing is measured; the line says so."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", init_method="env://")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run (control flow only, nothing measured)", "value": 0.0, "unit": "images/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
                          "config": {"workload": "PTOCR_BENCH_DRY rehearsal"}, "roofline": {"frac": 0.0}, "cpu_baseline": None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (SURVEY 8d: >= 100 for the median)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="det", choices=["det", "crnn", "ocr"],
                    help="det = the headline line (with the CRNN configs[2] result embedded under \"crnn\"); crnn = configs[2] alone; "
                         "ocr = configs[4] (DB++ r18 -> crops -> CRNN over 64 images of 1280x960)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32 det / 512 crnn / 64 ocr)")
    ap.add_argument("--crnn-steps", type=int, default=-1, help="steps of the embedded CRNN measurement (default: --steps; 0 = skip it)")
    ap.add_argument("--distinct-images", type=int, default=32, help="distinct synthetic images / maps (tiled to the batch if fewer than it)")
    ap.add_argument("--post-input", default=None, choices=["both", "stress", "model", "none"],
                    help="maps post-processed inside the timed step: the model's own maps (the pipeline's data flow; default with "
                         "--weights scene, whose maps hold ~140 boxes per image), text-like stress maps with ~140 boxes per image, or both "
                         "(default with --weights random, whose own maps are speckle: strictly more work than the real pipeline); "
                         "none = forward only, a diagnostic that is NOT the metric")
    ap.add_argument("--weights", default="scene", choices=["scene", "random"],
                    help="detector checkpoint: scene = random-init weights plus a fitted read-out in two head channels, so the maps of the "
                         "synthetic scene images are text-like; random = random-init everywhere (speckle maps, rounds 1-2)")
    ap.add_argument("--det-model", default="r18", choices=sorted(DET_VARIANTS),
                    help="r18 = BASELINE configs[1] (the metric); r18pp (DB++ / ASF) and mbv3s (MobileNetV3-small) are the "
                         "other detectors of the hot path")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16 = BASELINE configs[3] (with --det-model mbv3s): bf16 activations and weights, fp32 accumulation, fp32 maps")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="run the post-process synchronously after each forward instead of overlapping it with the next batch")
    ap.add_argument("--no-embed", dest="embed", action="store_false",
                    help="leave the configs[3] (mbv3s_bf16) and configs[4] (ocr) sub-lines out of the default det line")
    ap.add_argument("--cpu-images", type=int, default=8, help="images in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-lines", type=int, default=256, help="text lines in the CRNN CPU-baseline sample (0 = skip)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(sys.argv[1:], args.gpus)               # never returns
    rank, local, world = dist_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("PTOCR_BENCH_DRY") == "1":
        return run_dry(args, rank, world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev:
        # more ranks than GPUs is only the one-GPU rehearsal of the launch path: the split LSTM needs every CU for one
        # process, and RCCL wants one device per rank
        os.environ.setdefault("PTOCR_LSTM_SPLIT", "0")
        os.environ.setdefault("PTOCR_DIST_BACKEND", "gloo")
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PTOCR_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; "gloo" only to rehearse the control flow
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
        else:
            dist.init_process_group(backend=backend, init_method="env://")
    if args.dtype == "bf16" and (args.workload != "det" or args.det_model != "mbv3s"):
        raise SystemExit("bench.py: --dtype bf16 is BASELINE configs[3], i.e. --workload det --det-model mbv3s")
    args.crnn_batch = (args.batch or 512) if args.workload == "crnn" else 512
    # The embedded lines of the other BASELINE configs run as CHILD processes (started after the headline is measured, one workload at a
    # time; with several ranks every rank starts its own child and the children of a workload form their own process group on a port
    # rank 0 picked): a fault in one of them cannot take the headline with it -- but it is NOT hidden either: the child's error stands in
    # its place, its full stderr is kept under profiles/incidents/, and bench.py exits non-zero after printing the line.
    # PTOCR_BENCH_INPROC=1 runs everything in this process instead (tools/guard/guard_run.py does, to put every workload behind guard pages).
    isolate = os.environ.get("PTOCR_BENCH_INPROC") != "1"
    ports = {}
    if isolate and world > 1 and args.workload == "det":
        import torch.distributed as dist
        box = [None]
        if rank == 0:
            socks = [socket.socket() for _ in range(3)]
            for k in socks:
                k.bind(("127.0.0.1", 0))
            box = [[k.getsockname()[1] for k in socks]]
            for k in socks:
                k.close()
        dist.broadcast_object_list(box, src=0)
        ports = dict(zip(("crnn", "mbv3s_bf16", "ocr"), box[0]))

    def child_line(name, extra):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--no-embed", "--crnn-steps", "0"] + [str(v) for v in extra]
        env = dict(os.environ, PTOCR_BENCH_INPROC="1")
        if world > 1:
            env["MASTER_PORT"] = str(ports[name])
            # under torch.distributed.run the rendezvous store lives in the launcher's agent (on the launcher's port); the children of one
            # workload form their own group on their own port, where rank 0's child must host the store itself
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        err_text, what = "", None
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, env=env)
            err_text = r.stderr.decode(errors="replace")
            rows = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            if r.returncode == 0 and (rows or rank != 0):
                return json.loads(rows[-1]) if rows else None
            what = "child exited with %d" % r.returncode
        except subprocess.TimeoutExpired as e:
            err_text = (e.stderr or b"").decode(errors="replace")
            what = "child timed out after %d s" % e.timeout
        except Exception as e:                                       # unparsable line
            what = "%s: %s" % (type(e).__name__, e)
        keep = os.path.join(ROOT, "profiles", "incidents")          # the faulting address and access kind are in there: never thrown away
        os.makedirs(keep, exist_ok=True)
        path = os.path.join(keep, "bench_%s_rank%d_%s.stderr.log" % (name, rank, time.strftime("%Y%m%d_%H%M%S")))
        with open(path, "w") as f:
            f.write("# %s\n# %s\n%s" % (" ".join(cmd), what, err_text))
        return {"error": what, "stderr_file": os.path.relpath(path, ROOT), "stderr_tail": err_text[-1500:]}

    if args.workload == "det" and isolate and args.det_model == "r18" and args.dtype == "f32":
        line = run_det(args, rank, local, world, device)
        crnn_steps = args.steps if args.crnn_steps < 0 else args.crnn_steps
        torch.cuda.empty_cache()
        subs = {}
        if crnn_steps > 0:
            subs["crnn"] = child_line("crnn", ["--workload", "crnn", "--steps", crnn_steps, "--warmup", min(args.warmup, 10), "--cpu-lines", args.cpu_lines])
        if args.embed:
            subs["mbv3s_bf16"] = child_line("mbv3s_bf16", ["--det-model", "mbv3s", "--dtype", "bf16", "--steps", args.steps, "--warmup", min(args.warmup, 10),
                                                           "--cpu-images", min(args.cpu_images, 4), "--distinct-images", args.distinct_images, "--weights", args.weights]
                                    + ([] if args.overlap else ["--no-overlap"]) + (["--post-input", args.post_input] if args.post_input else []))
            subs["ocr"] = child_line("ocr", ["--workload", "ocr", "--steps", max(2, min(5, args.steps)), "--warmup", 2, "--cpu-images", min(args.cpu_images, 1)])
        failed = [k for k, v in subs.items() if isinstance(v, dict) and "error" in v]
        if rank == 0:
            line.update(subs)
    elif args.workload == "det":
        failed = []
        line = run_det(args, rank, local, world, device)
        crnn_steps = args.steps if args.crnn_steps < 0 else args.crnn_steps
        if crnn_steps > 0 and args.det_model == "r18":
            import copy
            a2 = copy.copy(args)
            a2.steps = crnn_steps
            a2.warmup = min(args.warmup, 10)
            torch.cuda.empty_cache()
            crnn = run_crnn(a2, rank, local, world, device)
            if rank == 0:
                line["crnn"] = crnn
        if args.embed and args.det_model == "r18" and args.dtype == "f32":
            import copy
            a3 = copy.copy(args)                                     # BASELINE configs[3]: bf16 MobileNetV3-small detector, 32 images per GPU
            a3.det_model, a3.dtype, a3.warmup, a3.cpu_images = "mbv3s", "bf16", min(args.warmup, 10), min(args.cpu_images, 4)
            torch.cuda.empty_cache()
            sub = run_det(a3, rank, local, world, device)
            if rank == 0:
                line["mbv3s_bf16"] = sub
            a4 = copy.copy(args)                                     # BASELINE configs[4]: run_ocr over 64 source images, a few steps
            a4.batch, a4.steps, a4.warmup = 0, max(2, min(5, args.steps)), 2
            torch.cuda.empty_cache()
            from pytorchocr_amd.deploy.bench_ocr import run_ocr_bench
            sub = run_ocr_bench(a4, rank, local, world, device, roofline_fn=_ocr_roofline,
                                cpu_fn=(lambda: ocr_cpu_baseline(1)) if args.cpu_images > 0 else None, parallelism_fn=parallelism)
            if rank == 0:
                line["ocr"] = sub
    elif args.workload == "crnn":
        failed = []
        line = run_crnn(args, rank, local, world, device)
    else:
        failed = []
        from pytorchocr_amd.deploy.bench_ocr import run_ocr_bench
        line = run_ocr_bench(args, rank, local, world, device, roofline_fn=_ocr_roofline,
                             cpu_fn=(lambda: ocr_cpu_baseline(max(1, min(2, args.cpu_images)))) if args.cpu_images > 0 else None,
                             parallelism_fn=parallelism)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if args.workload == "det" and failed:
        # the headline is printed; a workload that died is an ERROR of this run, not a footnote of it
        sys.stderr.write("bench.py: embedded workload(s) %s failed (stderr kept under profiles/incidents/)\n" % ", ".join(failed))
        raise SystemExit(3)


if __name__ == "__main__":
    main()
