#!/bin/bash
# gpurun helper: dbpost parity tests, then a rocprofv3 kernel-trace summary of the stand-alone post-process (tools/bench_post.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-post}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_dbpost.py -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/tools/bench_post.py 20 > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R
grep post-process gpurun_out/${TAG}_prof.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/${TAG}_prof/*/*kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0][-40:]
    if "ptocr" in n:
        per_call=float(r["TotalDurationNs"])/22/1e3
        tot+=per_call
        print("%-40s calls %4s avg %9.1f us  per call %8.1f us" % (n, r["Calls"], float(r["AverageNs"])/1e3, per_call))
print("sum of ptocr kernels per call: %.1f us" % tot)
PY
