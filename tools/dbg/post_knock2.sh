#!/bin/bash
# GPU box: device time of the two stage kernels of the stand-alone post-process with phases knocked out (PTOCR_DBPOST_DBG_SKIP; wrong
# results by design).  usage: post_knock2.sh <batch> <skip values...>
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-32}; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pk2
  PTOCR_DBPOST_DBG_SKIP=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk2 -- python3 $R/tools/bench_post.py 10 $B > /tmp/pk2.log 2>&1
  echo "skip=$v: $(grep 'border_quad\|border_wave' $(ls /tmp/pk2/*/*kernel_stats.csv | head -1) | awk -F, '{printf "%s %.1f us  ", substr($1,9,18), $4/1000}')"
done
