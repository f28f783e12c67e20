#!/bin/bash
# round-4 iteration: dbpost parity tests, then per-kernel stats on the stress maps and on the scene checkpoint's own maps, stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_dbpost.py -m gpu -x -q > gpurun_out/r4_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4_tests.log
[ $rc -ne 0 ] && exit $rc
python3 tools/dbg/post_stamps.py 32 > gpurun_out/r4_stamps32.log 2>&1; tail -32 gpurun_out/r4_stamps32.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4_post_prof $R/gpurun_out/scene_post
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_post_prof -- python3 $R/tools/bench_post.py 20 > $R/gpurun_out/r4_post_prof.log 2>&1
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r4_post_prof/*/*kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0][-40:]
    if "ptocr" in n:
        per_call=float(r["TotalDurationNs"])/22/1e3
        tot+=per_call
        print("%-40s calls %4s avg %9.1f us  per call %8.1f us" % (n, r["Calls"], float(r["AverageNs"])/1e3, per_call))
print("sum of ptocr kernels per call: %.1f us" % tot)
PY
bash tools/dbg/scene_post.sh 2>&1 | grep -v "^[EW]2026"
