#!/bin/bash
# GPU box: the bf16 1x1 kernels under knock-out builds (-DPW_DBG bits: 1 weight loads from one address, 2 pixel loads, 4 no SE scaling)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for v in "" "-DPW_DBG=1" "-DPW_DBG=2" "-DPW_DBG=4" "-DPW_DBG=7"; do
  export PTOCR_EXTRA_HIPCC_FLAGS="$v"
  python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"
  bash tools/dbg/bf16_trace.sh | grep "pw_bf16" | awk '{printf "%s ", $(NF-1)} END{print ""}'
done
