"""N > 1 path on CPU: two gloo ranks, weights broadcast from rank 0, static sharding, ordered gather."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorchocr_amd.modeling.architectures import build_model
    from pytorchocr_amd.parallel import broadcast_model_, gather_results, shard_range
    cfg = dict(model_type="det", algorithm="DB", Transform=None,
               Backbone=dict(name="ResNet", layers=18, pretrained=False),
               Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50))
    torch.manual_seed(100 + rank)                 # different random init per rank
    m = build_model(cfg)
    for b in m.buffers():
        if b.dtype == torch.int64:
            b.fill_(rank + 5)
    before = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).double().sum().item()
    broadcast_model_(m, src=0)
    after = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).double().sum().item()
    nbt = int(m.backbone.bn1.num_batches_tracked)
    s, e = shard_range(11, rank, world)
    res = gather_results([("img%d" % i, rank) for i in range(s, e)], rank, world)
    q.put((rank, before, after, nbt, (s, e), res))
    dist.destroy_process_group()


def test_broadcast_and_sharding_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0, n0, r0, g0), (_, b1, a1, n1, r1, g1) = out
    assert b0 != b1 and a0 == b0 and a1 == b0            # rank 1 now holds rank 0's weights
    assert n0 == n1 == 5                                  # integer buffers travel too
    assert r0 == (0, 6) and r1 == (6, 11)
    assert g0 == g1 and [x[0] for x in g0] == ["img%d" % i for i in range(11)]


def test_shard_range_covers_everything():
    from pytorchocr_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 64, 257):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
