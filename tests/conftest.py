import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing them.  On a GPU box
    nothing is skipped: a missing libptocr_hip.so still fails loudly (pytorchocr_amd._lib.lib() raises)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no ROCm device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def contract():
    import json
    with open(os.path.join(GOLD, "state_dict_contract.json")) as f:
        c = json.load(f)
    return {m: {k: (tuple(s), d) for k, (s, d) in v.items()} for m, v in c.items()}
