#!/bin/bash
# GPU box: F(4x4) Winograd kernel under compile-time variants: s_memtime phases (tools/wino4_timing.py) and the detector's bench line.
# usage: VARIANTS="flags1|flags2" w4_variants.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
IFS='|' read -ra VS <<< "${VARIANTS:--DW4_WARM=0|-DW4_WARM=256}"
for v in "${VS[@]}"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 tools/wino4_timing.py 2>&1 | grep -v amdgpu.ids
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 bench.py --no-embed --cpu-images 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms; wino frac', d['roofline']['frac'], 'all_conv ms', d['roofline']['all_conv']['ms_per_step'])"
done
