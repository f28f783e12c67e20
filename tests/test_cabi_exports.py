"""The C-ABI library builds for gfx950, loads without a GPU, and exports every symbol include/ptocr_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ptocr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptocr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pytorchocr_amd import _lib, build
    build.build(verbose=False)
    names = _header_symbols()
    assert len(names) >= 15
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), "libptocr_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names
    assert _lib.missing_exports() == []
    assert L.ptocr_version() >= 1


def test_calls_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pytorchocr_amd.modeling import ops
    with pytest.raises(RuntimeError):
        ops.nchw_to_nhwc(torch.zeros(1, 3, 8, 8), 4)
    from pytorchocr_amd.postprocess.db_postprocess import device_boxes
    with pytest.raises(RuntimeError):
        device_boxes(torch.zeros(1, 8, 8), [[8, 8]], 0.3, 0.5, 1.7)


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "pytorchocr_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "dbpost_oracle" in txt:
                    bad.append(fn)
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
    """no CPU fallback: without the built .so the very first call raises and names the build command"""
    from pytorchocr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "pytorchocr_amd", "no_such_libptocr_hip.so"))
    with pytest.raises(RuntimeError, match="pytorchocr_amd.build"):
        _lib.lib()
