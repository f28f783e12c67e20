#!/bin/bash
# GPU box: the stem + pool kernel under compile-time variants, alternating on ONE box: usage stem_ab.sh "<flags A>" "<flags B>" -- per
# variant only conv_stem_pool.hip is rebuilt into a library of its own copy; prints the stem's time from tools/dbg/stem_times.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for v in "$1" "$2"; do
    PTOCR_EXTRA_HIPCC_FLAGS="$v" python -m pytorchocr_amd.build > gpurun_out/stem_ab_build.log 2>&1 || { tail -3 gpurun_out/stem_ab_build.log; exit 1; }
    echo "== flags [$v]"
    PTOCR_EXTRA_HIPCC_FLAGS="$v" python tools/dbg/stem_times.py 2>&1 | grep "stem+pool"
  done
done
