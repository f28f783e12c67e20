#!/bin/bash
# GPU box: the headline line with the Winograd kernel's MFMA-issue priority off / on, alternating (is +1 % above the run-to-run noise?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for v in "-DW4_PRIO=0" "-DW4_PRIO=3" "-DW4_PRIO=0" "-DW4_PRIO=3" "-DW4_PRIO=2"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 bench.py --no-embed --cpu-images 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms; wino frac', d['roofline']['frac'], 'post overlapped', d['roofline_post']['ms_per_call_overlapped'])"
done
PTOCR_EXTRA_HIPCC_FLAGS="-DW4_PRIO=3" python3 -m pytorchocr_amd.build > /dev/null 2>&1
PTOCR_EXTRA_HIPCC_FLAGS="-DW4_PRIO=3" python3 bench.py --workload crnn --steps 40 --warmup 10 --cpu-lines 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('crnn prio 3', d['value'], d['ms_per_step'])"
PTOCR_EXTRA_HIPCC_FLAGS="-DW4_PRIO=0" python3 -m pytorchocr_amd.build > /dev/null 2>&1
PTOCR_EXTRA_HIPCC_FLAGS="-DW4_PRIO=0" python3 bench.py --workload crnn --steps 40 --warmup 10 --cpu-lines 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('crnn prio 0', d['value'], d['ms_per_step'])"
