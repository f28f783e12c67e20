# GPU box: kernel trace of ONE fp32 DBNet-r18 forward (32 x 736 x 1280): duration, workgroups, workgroup size, LDS, registers
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dt
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --post-input none --crnn-steps 0 --cpu-images 0 --no-embed > $GRAFT_REPO_ROOT/gpurun_out/dt.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/dt/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_pool_kernel" in r["Kernel_Name"]]
start = idx[-1]
tot = 0
for r in rows[start:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ptocr::", "").replace("ptocr::", "")[:44]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nb = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    print("%-46s %8.1f us  wgs %7d x %4d thr  lds %6s  vgpr %4s agpr %4s" % (n, d, nb, wg, r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"]))
    if "head_tail" in n: break
print("sum %.1f us" % tot)
PY
