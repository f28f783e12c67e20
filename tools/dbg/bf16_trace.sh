cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/bt
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bt -- python3 $GRAFT_REPO_ROOT/bench.py --det-model mbv3s --dtype bf16 --steps 2 --warmup 1 --post-input none --crnn-steps 0 --cpu-images 0 > $GRAFT_REPO_ROOT/gpurun_out/bt.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/bt/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: find last stem kernel
idx = [i for i, r in enumerate(rows) if "stem3x3s2_bf16" in r["Kernel_Name"] or "stem_dw_bf16" in r["Kernel_Name"]]
start = idx[-1]
tot = 0
for r in rows[start:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ptocr::", "").replace("ptocr::", "")[:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%-42s grid %-10s %8.1f us" % (n, r.get("Grid_Size", r.get("Grid_Size_X", "")), d))
    if "head_tail" in n: break
print("sum %.1f us" % tot)
PY
