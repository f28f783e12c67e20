#!/bin/bash
# GPU box: per-kernel times of the stand-alone DB post-process at several batch sizes (is a stage bound by per-border latency or by
# throughput?).  usage: prof_post_batches.sh <out dir under gpurun_out> [batches...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-postprof}; shift
BS=${@:-32 2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in $BS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -- python3 $R/tools/bench_post.py 20 $B > $O/b${B}_stdout.log 2>&1 || exit 1
  cp $(ls $O/b$B/*/*kernel_stats.csv | head -1) $O/b${B}_kernel_stats.csv
  rm -rf $O/b$B
done
