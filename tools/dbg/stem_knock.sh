#!/bin/bash
# GPU box: the bf16 stem under knock-out builds (-DSTEM_DBG bits: 1 every block reads image row 0, 2 no loads, 4 no stores)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for v in "" "-DSTEM_DBG=1" "-DSTEM_DBG=2" "-DSTEM_DBG=4" "-DSTEM_DBG=6"; do
  export PTOCR_EXTRA_HIPCC_FLAGS="$v"
  python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"
  bash tools/dbg/bf16_trace.sh | grep "stem"
done
