#!/bin/bash
# GPU box: timeline of border_wave_kernel under compile-time variants. usage: VARIANTS="flags1|flags2|..." sc_stamps.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
IFS='|' read -ra VS <<< "${VARIANTS:--DSC_DBG=16|-DSC_DBG=8}"
for v in "${VS[@]}"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== $v"; PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 tools/dbg/post_stamps.py 32 2>&1 | grep "kernel span\|ends\|masked sum\|state scan"
done
