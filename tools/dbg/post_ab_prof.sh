#!/bin/bash
# GPU box: A/B of dbpost.hip compile-time variants: device time of the stand-alone call (no profiler) AND the per-kernel table (rocprofv3).
# usage: post_ab_prof.sh "<flags of variant 1>" "<flags of variant 2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  echo "== variant: '${FLAGS}'"
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  touch pytorchocr_amd/csrc/dbpost.hip
  python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  bash tools/prof_post.sh ab$i | grep -v "calls    [12] avg" | grep -v "post-process:"
  for k in 1 2; do timeout -k 10 200 python3 tools/dbg/post_device_ms.py 2>&1 | grep "stress maps" || exit 1; done
done
