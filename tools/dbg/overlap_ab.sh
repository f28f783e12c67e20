#!/bin/bash
# gpurun helper: what the post-process costs the det step -- forward only / post-process overlapped on its own stream / synchronous
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "--post-input none" "" "--no-overlap" ${EXTRA_VARIANT:+"$EXTRA_VARIANT"}; do
  timeout -k 10 300 python bench.py --no-embed --crnn-steps 0 --cpu-images 0 --steps 60 --warmup 10 $v > gpurun_out/overlap_ab.tmp 2>gpurun_out/overlap_ab.err || { tail -5 gpurun_out/overlap_ab.err; exit 1; }
  python3 - "$v" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/overlap_ab.tmp").read().strip().splitlines()[-1])
print("%-22s %8.3f ms/step  %8.1f images/s  post alone %s overlapped %s" % (sys.argv[1] or "(default: overlap)", d["ms_per_step"], d["value"],
      (d.get("roofline_post") or {}).get("ms_per_call_alone"), (d.get("roofline_post") or {}).get("ms_per_call_overlapped")))
PY
done
