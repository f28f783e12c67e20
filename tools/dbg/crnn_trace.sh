#!/bin/bash
# GPU box: per-LAUNCH durations of one det forward (forward only: --post-input none), in launch order, from a rocprofv3 kernel trace:
# which layer each conv launch is and what it costs.  Output: gpurun_out/crnn_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/crnn_trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/crnn_trace -- python3 $R/bench.py --steps 6 --warmup 3 --cpu-images 0 --cpu-lines 0 --workload crnn > $R/gpurun_out/crnn_trace.log 2>&1 || { tail -5 $R/gpurun_out/crnn_trace.log; exit 1; }
cd $R
python3 - <<'PY' > gpurun_out/crnn_trace.txt
import csv, glob
f = glob.glob("gpurun_out/crnn_trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "ptocr" in r["Kernel_Name"]]
# one forward = from a stem_pool launch to the next
starts = [i for i, r in enumerate(rows) if "conv3x3_small_pool16" in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?")
    print("%9.1f us  +gap %6.1f  dur %8.1f us  grid %-8s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, g, r["Kernel_Name"].split("(")[0][-60:]))
    prev_end = e
    tot += e - s
print("forward: %.1f us of kernels, %.1f us first start to last end" % (tot / 1e3, (prev_end - t0) / 1e3))
PY
cat gpurun_out/crnn_trace.txt
rm -rf gpurun_out/crnn_trace
