#!/bin/bash
# GPU box: the FPN lateral in2 (conv_pw64_kernel) under knock-out builds (-DPW64_DBG bits: 1 no stores, 2 no top-down row loads)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for v in "" "-DPW64_DBG=1" "-DPW64_DBG=2" "-DPW64_DBG=3"; do
  export PTOCR_EXTRA_HIPCC_FLAGS="$v"
  python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"
  bash tools/dbg/det_trace.sh | grep "pw64\|pw128"
done
