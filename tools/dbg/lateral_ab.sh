#!/bin/bash
# GPU box: the FPN lateral kernels (conv_pw64 / conv_pw128) under compile-time variants, alternating on ONE box.  usage: lateral_ab.sh "<flags A>" "<flags B>"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for v in "$1" "$2"; do
    PTOCR_EXTRA_HIPCC_FLAGS="$v" python -m pytorchocr_amd.build > gpurun_out/lateral_ab_build.log 2>&1 || { tail -3 gpurun_out/lateral_ab_build.log; exit 1; }
    echo "== flags [$v]"
    PTOCR_EXTRA_HIPCC_FLAGS="$v" python tools/dbg/lateral_times.py 2>&1 | grep "x128->256\|x64->256"
  done
done
