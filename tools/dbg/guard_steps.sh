#!/bin/bash
# GPU box: the guard-page runs of the round's final state -- every GPU test with each allocation ending at an unmapped page, then the whole
# default bench in ONE process with each allocation STARTING at one and the post-process workspace poisoned
source tools/gpu_steps.sh guard
TAILN=6 step tests_end_mode 1100 python tools/guard/guard_run.py pytest tests -m gpu -x -q
export PTOCR_GUARD_MODE=start PTOCR_DBPOST_POISON=1
TAILN=3 step bench_start_mode_poison 900 python tools/guard/guard_run.py bench --steps 3 --warmup 1
