"""Packed-weight files: the kernel-layout weights of a model -- BatchNorm folded in fp64, Winograd-transformed,
channel-padded, fragment-ordered: everything `PackedModule.packed()` derives from a checkpoint -- in ONE flat file, so that a
serving process (or every rank of a node) starts from bytes it can hand to the kernels instead of re-deriving them.

Layout: b"PTOCRW1\\n" | uint64 little-endian header length | header (JSON, utf-8) | tensor data, each tensor 64-byte aligned.
The header lists the tensors (dtype, shape, offset, nbytes), the packed structure of every `PackedModule` of the model (nested
dicts / lists / `ops.Packed*` objects with tensors replaced by indices) and a SHA-256 over the model's state_dict, so that
`load_packed(..., check=True)` refuses a file that was packed from other weights.  No pickle: only the `ops.Packed*` classes
named in the header are instantiated, and only by name from `pytorchocr_amd.modeling.ops`.

    save_packed(model, path)                  # model on a ROCm device, in eval mode
    load_packed(model, path)                  # installs the packed weights; the next forward does not re-pack
    broadcast_packed_(model, src=0)           # rank src packs once, every rank installs the same bytes (RCCL / gloo)
"""
import hashlib
import io
import json
import struct

import numpy as np
import torch

MAGIC = b"PTOCRW1\n"
_DTYPES = {"torch.float32": torch.float32, "torch.float64": torch.float64, "torch.int32": torch.int32, "torch.int64": torch.int64,
           "torch.bfloat16": torch.bfloat16, "torch.float16": torch.float16, "torch.uint8": torch.uint8, "torch.int16": torch.int16,
           "torch.bool": torch.bool}


def _packed_classes():
    from ..modeling import ops
    return {n: getattr(ops, n) for n in dir(ops) if n.startswith("Packed") and isinstance(getattr(ops, n), type)}


def state_digest(model):
    """SHA-256 over the names, shapes, dtypes and bytes of the model's state_dict"""
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        t = v.detach().cpu().contiguous()
        h.update(k.encode()); h.update(str(tuple(t.shape)).encode()); h.update(str(t.dtype).encode())
        h.update(t.reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    return h.hexdigest()


def _encode(obj, tensors):
    if isinstance(obj, torch.Tensor):
        tensors.append(obj)
        return {"__t__": len(tensors) - 1}
    if isinstance(obj, (bool, int, float, str)) or obj is None:
        return obj
    if isinstance(obj, (np.integer, np.floating)):
        return obj.item()
    if isinstance(obj, dict):
        return {"__d__": [[_encode(k, tensors), _encode(v, tensors)] for k, v in obj.items()]}
    if isinstance(obj, (list, tuple)):
        return {"__l__" if isinstance(obj, list) else "__u__": [_encode(v, tensors) for v in obj]}
    cls = type(obj).__name__
    if cls in _packed_classes() and hasattr(obj, "__dict__"):
        return {"__o__": cls, "a": _encode(dict(vars(obj)), tensors)}
    raise TypeError("packed_weights: cannot serialise %r inside a packed structure" % type(obj))


def _decode(node, tensors, classes):
    if isinstance(node, dict):
        if "__t__" in node:
            return tensors[node["__t__"]]
        if "__d__" in node:
            return {_decode(k, tensors, classes): _decode(v, tensors, classes) for k, v in node["__d__"]}
        if "__l__" in node:
            return [_decode(v, tensors, classes) for v in node["__l__"]]
        if "__u__" in node:
            return tuple(_decode(v, tensors, classes) for v in node["__u__"])
        if "__o__" in node:
            cls = classes.get(node["__o__"])
            if cls is None:
                raise ValueError("packed_weights: unknown packed class %r" % node["__o__"])
            o = cls.__new__(cls)
            o.__dict__.update(_decode(node["a"], tensors, classes))
            return o
        raise ValueError("packed_weights: malformed header node")
    return node


def dumps(structure, extra=None):
    """structure: {module path: packed object} -> bytes of the file"""
    tensors = []
    enc = {k: _encode(v, tensors) for k, v in structure.items()}
    metas, off, blobs = [], 0, []
    for t in tensors:
        c = t.detach().cpu().contiguous()
        raw = c.reshape(-1).view(torch.uint8).numpy().tobytes() if c.numel() else b""
        pad = (-off) % 64
        blobs.append(b"\0" * pad)
        off += pad
        metas.append({"dtype": str(c.dtype), "shape": list(c.shape), "offset": off, "nbytes": len(raw)})
        blobs.append(raw)
        off += len(raw)
    header = json.dumps(dict(extra or {}, tensors=metas, modules=enc)).encode("utf-8")
    return MAGIC + struct.pack("<Q", len(header)) + header + b"".join(blobs)


def loads(data, device="cpu"):
    """bytes (or a uint8 tensor / memoryview) -> ({module path: packed object}, header dict)"""
    if isinstance(data, torch.Tensor):
        data = data.cpu().numpy().tobytes()
    data = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    if data[:len(MAGIC)] != MAGIC:
        raise ValueError("packed_weights: not a PTOCRW1 file")
    (hlen,) = struct.unpack("<Q", data[len(MAGIC):len(MAGIC) + 8])
    base = len(MAGIC) + 8
    header = json.loads(data[base:base + hlen].decode("utf-8"))
    body = memoryview(data)[base + hlen:]
    tensors = []
    for m in header["tensors"]:
        dt = _DTYPES[m["dtype"]]
        raw = np.frombuffer(body[m["offset"]:m["offset"] + m["nbytes"]], dtype=np.uint8).copy()
        t = torch.from_numpy(raw).view(dt).reshape(m["shape"]) if m["nbytes"] else torch.empty(m["shape"], dtype=dt)
        tensors.append(t.to(device))
    classes = _packed_classes()
    return {k: _decode(v, tensors, classes) for k, v in header["modules"].items()}, header


def _packed_modules(model):
    from ..modeling import ops
    return [(name, m) for name, m in model.named_modules() if isinstance(m, ops.PackedModule)]


def pack_bytes(model):
    if model.training:
        raise NotImplementedError("pytorchocr_amd implements the inference (eval) hot path only; call .eval()")
    mods = _packed_modules(model)
    if not mods:
        raise ValueError("packed_weights: the model holds no packed modules")
    return dumps({name: m.packed() for name, m in mods}, {"state_sha256": state_digest(model), "format": 1})


def save_packed(model, path):
    data = pack_bytes(model)
    with open(path, "wb") as f:
        f.write(data)
    return len(data)


def install(model, data, check=True):
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("pytorchocr_amd: model is on %s; move it to a cuda (ROCm) device -- no CPU fallback" % dev)
    structure, header = loads(data, dev)
    if check and header.get("state_sha256") != state_digest(model):
        raise ValueError("packed_weights: the file was packed from other weights than the model holds (state_dict digest differs); "
                         "load the matching checkpoint first or pass check=False to run from the packed weights alone")
    mods = dict(_packed_modules(model))
    if set(mods) != set(structure):
        raise ValueError("packed_weights: packed modules %s do not match the model's %s" % (sorted(structure), sorted(mods)))
    for name, m in mods.items():
        m._packed = structure[name]
        m._packed_sig = m._sig()
    return header


def load_packed(model, path, check=True):
    with open(path, "rb") as f:
        return install(model, f.read(), check=check)


def broadcast_packed_(model, src=0):
    """rank `src` packs (or has loaded) its weights; every rank installs the same bytes: one size + one flat-buffer broadcast.
    The model's state_dict is NOT touched (check=False on the receivers): the kernels read packed weights only."""
    import torch.distributed as dist
    dev = next(model.parameters()).device
    rank = dist.get_rank()
    buf_dev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    if rank == src:
        data = pack_bytes(model)
        n = torch.tensor([len(data)], dtype=torch.int64, device=buf_dev)
    else:
        data, n = None, torch.zeros(1, dtype=torch.int64, device=buf_dev)
    dist.broadcast(n, src)
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(buf_dev) if rank == src else torch.empty(int(n.item()), dtype=torch.uint8, device=buf_dev)
    dist.broadcast(buf, src)
    if rank != src:
        install(model, buf.cpu().numpy().tobytes(), check=False)
    return int(n.item())
