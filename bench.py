#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the detection hot path over one batch that is already resident in HBM:
DBNet-r18 fp32 forward (HIP MFMA convs) on f32[32,3,736,1280] -> probability maps -> DB post-process on the
device -> int16 boxes on the host (BASELINE.json configs[1]).  With N > 1 every rank (one process per GPU)
runs the same per-GPU batch on its own images (weak scaling, no data-path collective); the only collective is
the RCCL broadcast of the weights from rank 0 before the timed region.  Rank 0 prints ONE JSON line.

--workload crnn measures BASELINE.json configs[2] instead (CRNN text-lines/sec, batch 512 of 32x320 crops).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
DET_GFLOP_PER_IMG = 114.195        # SURVEY.md 8d / BASELINE.md: DBNet-r18 @ 3x736x1280 (2*MAC over conv/deconv)
DET_TAIL_GFLOP_PER_IMG = 2.05      # ConvT 64->64 (1.93) + ConvT 64->1 (0.12) run in db_head_tail_kernel, not in the conv kernels
CRNN_GFLOP_PER_LINE = 4.980

DET_R18 = dict(model_type="det", algorithm="DB", Transform=None,
               Backbone=dict(name="ResNet", layers=18, pretrained=False),
               Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False, attention_type="scale_channel_spatial"),
               Head=dict(name="DBHead", k=50))
# the other detectors of SURVEY 8a, for tools / DESIGN numbers only (--det-model); the metric is DBNet-r18
DET_VARIANTS = {
    "r18": (DET_R18, "det_r18_db", 114.195),
    "r18pp": (dict(DET_R18, Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=True, attention_type="scale_channel_spatial")),
              "detpp_r18_db", 131.59),
    "mbv3s": (dict(model_type="det", algorithm="DB", Transform=None,
                   Backbone=dict(name="MobileNetV3", model_name="small", scale=1.0, pretrained=False),
                   Neck=dict(name="FPN", out_channels=96, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50)),
              "det_mbv3s_db", 8.43),
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


def build_and_sync_weights(cfg, contract_name, device, rank, world):
    """Random-init weights of the named architecture: rank 0 makes them, RCCL broadcast puts them on every GPU."""
    import torch.distributed as dist
    from pytorchocr_amd.modeling.architectures import build_model
    from pytorchocr_amd.utils.synth import synth_state_dict
    model = build_model(cfg)
    if rank == 0:
        sd = synth_state_dict(load_contract(contract_name))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(device).eval()
    if world > 1:
        from pytorchocr_amd.parallel import broadcast_model_
        broadcast_model_(model, src=0)
    return model


def det_cpu_baseline(n_img, H, W):
    """The oracle (CPU restatement of the reference path: torch-CPU fp32 forward + C post-process with
    cpp_speedup=True semantics) on a bounded sample of the same workload."""
    from oracle import dbpost, model_oracle
    from pytorchocr_amd.utils.synth import synth_images, synth_prob_maps, synth_state_dict
    torch.set_num_threads(min(16, os.cpu_count() or 1))            # the GPU box's CPU share for one GPU is 16 cores
    sd = synth_state_dict(load_contract("det_r18_db"))
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    x = torch.from_numpy(synth_images(1, 3, H, W, seed=2022))
    stress = synth_prob_maps(1, H, W, seed=2022)[0]
    model_oracle.dbnet_r18_forward(sd, x[:, :, :64, :64])          # warm-up of the thread pool
    t_model = t_post = 0.0
    for _ in range(n_img):                                         # batch 1 each, as infer_det.py:85-103 does
        t0 = time.perf_counter()
        maps = model_oracle.dbnet_r18_forward(sd, x)["maps"].numpy()
        t1 = time.perf_counter()
        for m in (maps[0, 0], stress):                             # the net's own map + the text-like stress map
            bm = dbpost.binarize(m, 0.3)
            dbpost.boxes_from_bitmap(m, bm, 0.5, 1.7, W, H)
        t2 = time.perf_counter()
        t_model += t1 - t0
        t_post += (t2 - t1) / 2
    total = t_model + t_post
    return {"value": round(n_img / total, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d images %dx%d, batch 1 each: torch-CPU fp32 forward %.3f s/img + C post-process %.4f s/img "
                      "(single thread, like the GIL-bound reference extension)" % (n_img, H, W, t_model / n_img, t_post / n_img)}


def crnn_cpu_baseline(n_lines):
    """The oracle (torch-CPU fp32 CRNN forward + the reference's CTCLabelDecode restatement) on a bounded sample of lines,
    one batch of 16 at a time."""
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


def run_det(args, rank, local, world, device):
    from pytorchocr_amd.modeling import ops
    from pytorchocr_amd.postprocess import build_post_process
    from pytorchocr_amd.utils.synth import synth_images, synth_prob_maps
    B, H, W = args.batch, 736, 1280
    det_cfg, det_contract, _ = DET_VARIANTS[args.det_model]
    model = build_and_sync_weights(det_cfg, det_contract, device, rank, world)
    post = build_post_process(DET_POST, dict(use_gpu=True, seed=2022))
    # synthetic images: a few distinct seeded images tiled to the batch (the generator is slow on the host)
    base = synth_images(4, 3, H, W, seed=2022 + rank)
    x = torch.from_numpy(base).to(device).repeat(B // 4 + 1, 1, 1, 1)[:B].contiguous()
    shape_list = np.array([[H, W, 1.0, 1.0]] * B)
    # Random weights give noise-like maps; so that the post-process does realistic work (~130 text boxes per
    # image) the text-like stress maps are blended into the timed path's post-process input as a SECOND pass
    stress = torch.from_numpy(synth_prob_maps(4, H, W, seed=7 + rank)).to(device).repeat(B // 4 + 1, 1, 1)[:B, None].contiguous()

    def step(pending):
        """forward of this batch; its post-process is queued on the post-process stream and collected one step later,
        so it overlaps with the next batch's convolutions (every batch's boxes are on the host before the clock stops)"""
        with torch.no_grad():
            out = model(x)
        futs = []
        if args.post_input in ("model", "both"):        # the pipeline's own data flow
            futs.append(post.submit({"maps": out["maps"]}, shape_list))
        if args.post_input in ("stress", "both"):       # realistic ~140 boxes/image load (random weights give noise maps)
            futs.append(post.submit({"maps": stress}, shape_list))
        res = [f.result() for f in pending] if args.overlap else []
        if not args.overlap:
            res = [f.result() for f in futs]
            futs = []
        return futs, (res[-1] if res else [])

    pending = []
    for _ in range(args.warmup):
        pending, _ = step(pending)
    for f in pending:
        f.result()
    pending = []
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()
    ops.PROFILE = [] if rank == 0 else None
    ops.PROFILE_LABELS = [] if rank == 0 else None
    t0 = time.perf_counter()
    nbox = 0
    for _ in range(args.steps):
        pending, res = step(pending)
        nbox += sum(len(r["points"]) for r in res)
    for f in pending:                                    # drain: the last batch's boxes
        res = f.result()
    nbox += sum(len(r["points"]) for r in res) if pending else 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, labels = ops.PROFILE, ops.PROFILE_LABELS
    ops.PROFILE = ops.PROFILE_LABELS = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    # Dominant kernel = conv_wino_kernel (Winograd F(2x2,3x3) on fp32 MFMA: every 3x3/s1 layer).  `achieved` counts the
    # ALGORITHMIC (direct-convolution) FLOPs of those launches, as SURVEY 8d defines the work; Winograd executes 2.25x fewer
    # multiplies, so `executed_*` gives what the matrix pipe really ran (the honest utilisation of the 157.3 TFLOP/s peak).
    ms = [e0.elapsed_time(e1) for e0, e1 in prof]
    conv_ms = sum(ms)
    n_launch = len(prof)
    wino_ms, wino_flops, n_wino = 0.0, 0.0, 0
    for lab, t in zip(labels, ms):
        if lab.startswith("wino3x3"):
            n_, h_, w_, ci = [int(v) for v in lab.split()[1].split("->")[0].split("x")]
            co = int(lab.split("->")[1].split()[0])
            wino_ms += t; n_wino += 1
            wino_flops += 2.0 * n_ * h_ * w_ * ci * co * 9
    conv_flops = (DET_VARIANTS[args.det_model][2] - DET_TAIL_GFLOP_PER_IMG) * 1e9 * B * args.steps
    conv_all = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    achieved = wino_flops / (wino_ms * 1e-3) / 1e12 if wino_ms > 0 else 0.0
    cpu = det_cpu_baseline(args.cpu_images, H, W) if world == 1 and args.cpu_images > 0 else None
    traffic = None
    tp = os.path.join(ROOT, "profiles", "conv_traffic.json")       # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    if os.path.exists(tp):
        with open(tp) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    line = {
        "metric": "images/sec end-to-end (DBNet-r18 det+post, 736x1280)" if args.det_model == "r18"
                  else "images/sec end-to-end (%s det+post, 736x1280; NOT the BASELINE metric)" % args.det_model,
        "value": round(world * B * args.steps / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "DBNet r18 fp32, batch %d synthetic 736x1280 per GPU, HIP conv + HIP DBPostProcess "
                               "(BASELINE.json configs[1])" % B,
                   "global_batch": world * B, "post_input": args.post_input, "post_overlap": bool(args.overlap), "boxes_per_image": round(nbox / (B * args.steps), 1),
                   "parallelism": "image-sharded x%d, RCCL weight broadcast only" % world},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                     "kernel": "conv_wino_kernel (%d launches per step, %.3f ms avg launch, HIP events on the launch stream); "
                               "achieved = direct-convolution FLOPs of those launches / their time"
                               % (n_wino // max(args.steps, 1), wino_ms / max(n_wino, 1)),
                     "executed_tflops": round(achieved / 2.25, 2), "executed_frac": round(achieved / 2.25 / PEAK_F32_MFMA_TFLOPS, 4),
                     "all_conv": {"launches_per_step": n_launch // max(args.steps, 1), "ms_per_step": round(conv_ms / max(args.steps, 1), 3),
                                  "algorithmic_tflops": round(conv_all, 2),
                                  "kernels": "conv_wino_kernel (3x3 s1) + stem_conv_kernel (7x7 s2) + conv_pw64_kernel (FPN in2) + conv_mfma_v2_kernel (3x3 s2, other 1x1)"}},
        "cpu_baseline": cpu,
    }
    return line


def run_crnn(args, rank, local, world, device):
    from pytorchocr_amd.postprocess import build_post_process
    from pytorchocr_amd.utils.synth import synth_text_lines
    B = args.batch
    model = build_and_sync_weights(crnn_cfg(), "rec_vgg_bilstm_ctc", device, rank, world)
    post = build_post_process(dict(name="CTCLabelDecode"), dict(character_dict_path=os.path.join(
        ROOT, "pytorchocr_amd", "utils", "char_dict_6623.txt"), use_space_char=False))
    base = synth_text_lines(16, 32, 320, seed=2022 + rank)
    x = torch.from_numpy(base).to(device).repeat(B // 16 + 1, 1, 1, 1)[:B].contiguous()

    def step(pending):
        """forward + CTC arg-max of this batch on the GPU; its label ids / confidences travel to pinned host memory behind it
        and are decoded to text (host Python, as in the reference) one step later, i.e. while the next batch computes.  Every
        batch's texts are on the host before the clock stops."""
        with torch.no_grad():
            fut = post.submit(model.forward_greedy(x)) if args.overlap else None
            res = pending.result() if pending is not None else None
            if not args.overlap:
                res = post(model.forward_greedy(x))
        return fut, res

    pending = None
    for _ in range(args.warmup):
        pending, _ = step(pending)
    if pending is not None:
        pending.result()
    pending = None
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    nchar = 0
    for _ in range(args.steps):
        pending, res = step(pending)
        nchar += sum(len(t) for t, _ in res) if res else 0
    if pending is not None:
        res = pending.result()                               # drain: the last batch's texts
        nchar += sum(len(t) for t, _ in res)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    achieved = CRNN_GFLOP_PER_LINE * 1e9 * B * args.steps / dt / 1e12
    return {
        "metric": "CRNN text-lines/sec (vgg_v1_x1.0 + BiLSTM + CTC greedy, 32x320)",
        "value": round(world * B * args.steps / dt, 2), "unit": "text-lines/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "CRNN vgg_v1_x1.0 + CTC greedy, batch %d synthetic 32x320 gray crops per GPU (BASELINE.json configs[2])" % B,
                   "global_batch": world * B, "decode_overlap": bool(args.overlap), "chars_per_line": round(nchar / (B * args.steps), 1),
                   "parallelism": "line-sharded x%d" % world},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                     "kernel": "whole step (end-to-end FLOP rate; per-kernel split in profiles/)"},
        "cpu_baseline": crnn_cpu_baseline(args.cpu_lines) if world == 1 and args.cpu_lines > 0 else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="det", choices=["det", "crnn"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32 det / 512 crnn)")
    ap.add_argument("--post-input", default="both", choices=["both", "stress", "model", "none"],
                    help="maps post-processed inside the timed step: the model's own maps (true data flow; random weights "
                         "give noise-like maps), text-like stress maps with ~130 boxes per image, or both (default: "
                         "strictly more work than the real pipeline); none = forward only, a diagnostic that is NOT the metric")
    ap.add_argument("--det-model", default="r18", choices=sorted(DET_VARIANTS),
                    help="r18 = BASELINE configs[1] (the metric); r18pp (DB++ / ASF) and mbv3s (MobileNetV3-small, fp32) are the "
                         "other detectors of the hot path, timed for DESIGN.md only")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="run the post-process synchronously after each forward instead of overlapping it with the next batch")
    ap.add_argument("--cpu-images", type=int, default=8, help="images in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-lines", type=int, default=256, help="text lines in the CRNN CPU-baseline sample (0 = skip)")
    args = ap.parse_args()
    if args.batch == 0:
        args.batch = 32 if args.workload == "det" else 512
    rank, local, world = dist_env()
    if world != args.gpus:
        if args.gpus == 1 and world == 1:
            pass
        else:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # one process per GPU; the modulo only matters for the 2-ranks-on-1-GPU rehearsal
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
    line = (run_det if args.workload == "det" else run_crnn)(args, rank, local, world, device)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
