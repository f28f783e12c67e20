#!/bin/bash
# gpurun helper: bf16 parity tests, then a rocprofv3 kernel-trace summary of the bf16 mbv3s forward (no post-process)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-bf16}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --det-model mbv3s --dtype bf16 --steps 5 --warmup 2 --post-input none > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R
python3 - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/${TAG}_prof/*/*kernel_stats.csv")[0]
tot=0
for r in list(csv.DictReader(open(f))):
    n=r["Name"][:58]
    if "ptocr" in n:
        per=float(r["TotalDurationNs"])/8/1e3; tot+=per
        print("%-58s calls %4s avg %8.1f us  per fwd %8.1f us" % (n, r["Calls"], float(r["AverageNs"])/1e3, per))
print("sum per forward: %.1f us" % tot)
l=json.loads([x for x in open("gpurun_out/${TAG}_prof.log") if x.startswith("{")][-1])
print(l["value"], l["ms_per_step"], l["roofline"]["achieved"], l["roofline"]["frac"])
PY
