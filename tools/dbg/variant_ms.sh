#!/bin/bash
# GPU box: device time of the post-process (scene checkpoint's own maps and the text-like stress maps) under compile-time variants.
# usage: VARIANTS="flags1|flags2|..." variant_ms.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
IFS='|' read -ra VS <<< "${VARIANTS:-|-DPT_SLAB_ROWS=16}"
for v in "${VS[@]}"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"; PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 tools/dbg/scene_post.py r18 2>&1 | grep "device ms"
done
