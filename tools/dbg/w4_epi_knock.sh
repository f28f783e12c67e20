#!/bin/bash
# GPU box: phases of the F(4x4) kernel's epilogue by knock-out (W4_DBG: 1 no global stores, 2 no consumer section, 4 no exchange writes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for v in "" "-DW4_DBG=1" "-DW4_DBG=2" "-DW4_DBG=4" "-DW4_DBG=6"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 tools/wino4_timing.py 2>&1 | grep "head" | head -1
done
