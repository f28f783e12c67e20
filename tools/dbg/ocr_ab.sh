#!/bin/bash
# GPU box: the run_ocr bench line for a list of build variants, two alternating rounds.  usage: ocr_ab.sh "<flags>" "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
for FLAGS in "$@"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS" python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  timeout -k 10 300 python bench.py --workload ocr --cpu-images 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s ocr: %.1f images/s  %.2f ms/step' % (sys.argv[1], d['value'], d['ms_per_step']))" "'$FLAGS'"
  timeout -k 10 300 python bench.py --workload crnn --cpu-lines 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s crnn: %.0f lines/s  %.3f ms/step' % (sys.argv[1], d['value'], d['ms_per_step']))" "'$FLAGS'"
done
done
