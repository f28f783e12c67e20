#!/bin/bash
# GPU box: device time of named kernels of the stand-alone post-process under PTOCR_DBPOST_DBG_SKIP values. usage: post_knock3.sh <batch> <kernel substring> <skip values...>
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-32}; K=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pk3
  PTOCR_DBPOST_DBG_SKIP=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk3 -- python3 $R/tools/bench_post.py 10 $B > /tmp/pk3.log 2>&1
  python3 - "$K" "$v" $(ls /tmp/pk3/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
k, v, f = sys.argv[1:4]
print("skip=%s:" % v, "  ".join("%s calls %s avg %.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f)) if k in r["Name"]))
PY
done
