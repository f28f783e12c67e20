# GPU box: kernel trace of ONE CRNN step (512 lines 32x320): duration, workgroups, workgroup size
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ct
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ct -- python3 $GRAFT_REPO_ROOT/bench.py --workload crnn --steps 2 --warmup 1 --cpu-lines 0 > $GRAFT_REPO_ROOT/gpurun_out/ct.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ct/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "conv3x3_small_pool" in r["Kernel_Name"]]
start = idx[-1]
tot = 0
for r in rows[start:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ptocr::", "").replace("ptocr::", "")[:46]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nb = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    print("%-48s %8.1f us  wgs %7d x %4d thr  lds %6s" % (n, d, nb, wg, r["LDS_Block_Size"]))
    if "ctc_combine" in n: break
print("sum %.1f us" % tot)
PY
