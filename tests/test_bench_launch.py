"""bench.py as the driver invokes it: `python bench.py --gpus N` with no WORLD_SIZE must start its own N ranks (a child
torch.distributed.run), relay rank 0's JSON line and exit 0.  On CPU the ranks run the PTOCR_BENCH_DRY control-flow rehearsal
(gloo rendezvous, barrier, max over ranks); the -m gpu variant runs the real workload with two ranks sharing the one GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra, timeout):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_self_launch_two_ranks_dry():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"PTOCR_BENCH_DRY": "1"}, 300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["roofline"]["frac"] <= 1


def test_world_size_mismatch_is_an_error():
    p, lines = _run(["--gpus", "1"], {"PTOCR_BENCH_DRY": "1", "WORLD_SIZE": "2", "RANK": "0"}, 120)
    assert p.returncode != 0 and not lines


@pytest.mark.gpu
def test_self_launch_two_ranks_on_the_gpu_box():
    """the real workload, two ranks (gloo: both share the box's one GPU), tiny batch"""
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-images", "0", "--cpu-lines", "0",
                     "--crnn-steps", "1"], {}, 900)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and 0 < line["roofline"]["frac"] <= 1
    assert line["roofline_post"]["frac"] <= 1 and line["crnn"]["value"] > 0 and line["crnn"]["roofline"]["frac"] <= 1
