"""Packed-weight files (pytorchocr_amd/utils/packed_weights.py): the codec on the CPU, and on the GPU that a
model running from a packed file -- or from rank 0's broadcast bytes -- computes exactly what the model that packed them does."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from pytorchocr_amd.utils import packed_weights as pw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DET = dict(model_type="det", algorithm="DB", Transform=None, Backbone=dict(name="ResNet", layers=18, pretrained=False),
           Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50))


def test_codec_roundtrip_without_a_gpu():
    from pytorchocr_amd.modeling import ops
    pc = ops.PackedConv.__new__(ops.PackedConv)
    pc.__dict__.update(dict(w=torch.arange(12, dtype=torch.float32).reshape(3, 4), b=torch.zeros(3), cin=4, relu=1, wino_u=None,
                            half=torch.tensor([1.5, -2.0]).to(torch.bfloat16), empty=torch.zeros(0, 4), name="x", flag=True))
    s = {"backbone": {"stem": pc, "blocks": [("ir", {"dw": pc, "res": False}), ("cba", pc)], 7: "int key"},
         "head": (1, 2.5, None, torch.tensor([1, 2, 3], dtype=torch.int32), np.int64(9))}
    data = pw.dumps(s, {"state_sha256": "abc"})
    assert data[:8] == pw.MAGIC
    s2, h = pw.loads(data)
    assert h["state_sha256"] == "abc" and all(m["offset"] % 64 == 0 for m in h["tensors"])
    q = s2["backbone"]["stem"]
    assert type(q) is ops.PackedConv and torch.equal(q.w, pc.w) and q.half.dtype == torch.bfloat16 and torch.equal(q.half, pc.half)
    assert q.empty.shape == (0, 4) and q.wino_u is None and q.flag is True and q.name == "x" and q.cin == 4
    assert s2["backbone"]["blocks"][0][0] == "ir" and s2["backbone"]["blocks"][0][1]["res"] is False and s2["backbone"][7] == "int key"
    assert isinstance(s2["head"], tuple) and s2["head"][1] == 2.5 and s2["head"][2] is None and s2["head"][4] == 9
    assert s2["head"][3].dtype == torch.int32 and s2["head"][3].tolist() == [1, 2, 3]
    with pytest.raises(ValueError):
        pw.loads(b"NOTAFILE" + data[8:])
    with pytest.raises(TypeError):
        pw.dumps({"x": object()})
    bad = data.replace(b'"PackedConv"', b'"PackedEvil"')
    with pytest.raises(ValueError):
        pw.loads(bad)


def _load_synth(m, contract, key, seed=2022):
    from pytorchocr_amd.utils.synth import synth_state_dict
    sd = synth_state_dict(contract[key], seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m


@pytest.mark.gpu
def test_model_runs_from_a_packed_file(tmp_path, contract):
    from pytorchocr_amd.modeling.architectures import build_model
    from pytorchocr_amd.utils.synth import synth_images
    import copy
    dev = torch.device("cuda:0")
    a = _load_synth(build_model(copy.deepcopy(DET)), contract, "det_r18_db").to(dev).eval()
    x = torch.from_numpy(synth_images(2, 3, 96, 160, seed=3)).to(dev)
    with torch.no_grad():
        ya = a(x)["maps"].cpu()
    path = str(tmp_path / "det_r18.ptocrw")
    n = pw.save_packed(a, path)
    assert n == os.path.getsize(path) and n > 50e6                  # direct + both Winograd forms of 12.3 M parameters
    # a model holding OTHER weights: the file is refused unless the caller opts out of the check, then it computes a's maps
    torch.manual_seed(7)
    b = build_model(copy.deepcopy(DET)).to(dev).eval()
    with pytest.raises(ValueError):
        pw.load_packed(b, path)
    pw.load_packed(b, path, check=False)
    with torch.no_grad():
        assert torch.equal(b(x)["maps"].cpu(), ya)
    # with the matching checkpoint loaded the check passes, and nothing is re-packed at the next forward
    c = _load_synth(build_model(copy.deepcopy(DET)), contract, "det_r18_db").to(dev).eval()
    pw.load_packed(c, path)
    marker = c.backbone._packed
    with torch.no_grad():
        assert torch.equal(c(x)["maps"].cpu(), ya)
    assert c.backbone._packed is marker
    # a later load_state_dict invalidates the installed weights (version counters), as for self-packed ones
    _load_synth(c, contract, "det_r18_db", seed=5)
    with torch.no_grad():
        assert not torch.equal(c(x)["maps"].cpu(), ya)


def _rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import copy
    import json
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorchocr_amd.modeling.architectures import build_model
    from pytorchocr_amd.utils.synth import synth_images
    dev = torch.device("cuda:0")
    torch.manual_seed(50 + rank)
    m = build_model(copy.deepcopy(DET))
    if rank == 0:
        with open(os.path.join(ROOT, "tests", "golden", "state_dict_contract.json")) as f:
            contract = {k: (tuple(s), d) for k, (s, d) in json.load(f)["det_r18_db"].items()}
        _load_synth(m, {"det_r18_db": contract}, "det_r18_db")
    m = m.to(dev).eval()
    nbytes = pw.broadcast_packed_(m, src=0)
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=11)).to(dev)
    with torch.no_grad():
        y = m(x)["maps"].cpu().numpy()
    q.put((rank, nbytes, y))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_broadcast_packed_two_ranks(gold_dir):
    """rank 0 holds the checkpoint and packs once; rank 1 (random init) installs the broadcast bytes: identical maps, equal to the
    reference's own output for these weights"""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] == out[1][1] > 50e6
    assert np.array_equal(out[0][2], out[1][2])
    g = np.load(os.path.join(gold_dir, "det_r18_db_1x3x64x96.npz"))
    assert np.abs(out[1][2] - g["maps"]).max() <= 1e-4
