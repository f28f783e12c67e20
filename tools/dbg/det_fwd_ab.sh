#!/bin/bash
# GPU box: forward-only step time of the det bench (no profiler) for a list of build variants, each measured twice in alternation.
# usage: det_fwd_ab.sh "<flags>" "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
for FLAGS in "$@"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS" python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  timeout -k 10 300 python bench.py --no-embed --crnn-steps 0 --cpu-images 0 --steps 80 --warmup 10 --post-input none 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s forward only: %.3f ms/step (median %.3f)' % (sys.argv[1], d['ms_per_step'], d['ms_per_step_median']))" "'$FLAGS'"
done
done
