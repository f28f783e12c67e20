#!/bin/bash
# GPU box: the round's closing validation -- the whole GPU suite, then the seeded dbpost sweep (3 943 cases), then the other sweeps
source tools/gpu_steps.sh final1
step suite 1000 python -m pytest tests -m gpu -q -x
export PTOCR_DBPOST_FUZZ=3000 PTOCR_DBPOST_FUZZ_BIG=150
step dbpost_fuzz 1000 python -m pytest tests/test_gpu_dbpost.py -m gpu -q -x
unset PTOCR_DBPOST_FUZZ PTOCR_DBPOST_FUZZ_BIG
