#!/bin/bash
# GPU box: the bf16 detector's depthwise kernels under compile-time variants (VARIANTS="flags1|flags2|..."): bf16 tests, then the kernel
# times of one forward.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
IFS='|' read -ra VS <<< "${VARIANTS:-|-DDW_PF=0}"
for v in "${VS[@]}"; do
  export PTOCR_EXTRA_HIPCC_FLAGS="$v"
  python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"
  python3 -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -1
  bash tools/dbg/bf16_trace.sh | grep "dwconv\|expand_dw\|sum\|stem\|head_tail"
  python3 bench.py --det-model mbv3s --dtype bf16 --steps 40 --warmup 10 --crnn-steps 0 --cpu-images 0 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('img/s', j['value'], 'ms', j['ms_per_step'])"
done
