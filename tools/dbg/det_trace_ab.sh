#!/bin/bash
# GPU box: det forward per-launch trace + unprofiled forward-only step time for a list of build variants.  usage: det_trace_ab.sh "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for FLAGS in "$@"; do
  echo "== variant: '${FLAGS}'"
  PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS" python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  bash tools/dbg/det_trace.sh > gpurun_out/det_trace_ab.tmp 2>&1 || { tail -5 gpurun_out/det_trace_ab.tmp; exit 1; }
  grep -E "wino4r|forward:" gpurun_out/det_trace_ab.tmp | awk '{ if ($0 ~ /forward/) print; else { g += $4; d += $6; n++ } } END { printf("wino launches %d: sum dur %.1f us, sum gaps in front %.1f us\n", n, d, g) }'
  timeout -k 10 300 python bench.py --no-embed --crnn-steps 0 --cpu-images 0 --steps 60 --warmup 10 --post-input none 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forward only, no profiler: %.3f ms/step' % d['ms_per_step'])"
done
