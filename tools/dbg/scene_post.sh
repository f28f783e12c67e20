cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY=model rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scene_post -- python3 $R/tools/dbg/scene_post.py r18 > $R/gpurun_out/scene_post.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/scene_post/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r["Name"]
    if "ptocr" in n and not any(s in n for s in ("conv", "stem", "head_tail", "pw", "nchw")):
        print("%-70s %4s avg %8.1f us max %8.1f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
tail -3 $R/gpurun_out/scene_post.log
