"""Multi-GPU glue for the inference hot path: one process per GPU, images sharded statically, and ONE collective --
the broadcast of the weights from rank 0 (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference has no multi-GPU inference (single --gpu_id, deploy/pytorch/infer_det.py:36-40); its NCCL use is
training-only (tools/program.py:505-508).  Images are independent end to end (eval-mode BN), so steady state has
no inter-GPU traffic: results are tiny int16 boxes / label ids returned to the host per rank.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous static split of n_items over ranks: returns (start, stop)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_model_(model, src=0):
    """In-place broadcast of every parameter and buffer of `model` from rank `src`, as ONE flat buffer per dtype
    (47 MB fp32 for DBNet-r18: a single per-link-bound fan-out instead of ~200 small messages)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    tensors = [t for t in list(model.parameters()) + list(model.buffers())]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype, ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    return model


def gather_results(local_results, rank, world):
    """Concatenate per-rank python result lists in rank order on every rank (host-side objects only)."""
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return list(local_results)
    out = [None] * world
    dist.all_gather_object(out, list(local_results))
    return [r for part in out for r in part]
