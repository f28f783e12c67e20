#!/bin/bash
# GPU box: border_wave_kernel with phases of the score role compiled out (-DSC_DBG: 1 no masked sum, 2 no state scan, 4 no prefix pass); results wrong by design
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for v in ${VARIANTS:-0 1 2 4 7}; do
  PTOCR_EXTRA_HIPCC_FLAGS="-DSC_DBG=$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sck
  PTOCR_EXTRA_HIPCC_FLAGS="-DSC_DBG=$v" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sck -- python3 $R/tools/bench_post.py 10 > /tmp/sck.log 2>&1
  cd $R
  echo "SC_DBG=$v: $(grep border_stage /tmp/sck/*/*kernel_stats.csv | awk -F, '{print $4}') ns"
done
