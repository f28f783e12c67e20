# GPU box: per-kernel averages of the stand-alone post-process (text-like maps).  usage: post_kstats.sh [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rm -rf $R/gpurun_out/pk
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pk -- python3 $R/tools/bench_post.py 30 > $R/gpurun_out/pk.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/pk/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ptocr" in r["Name"]:
        print("%-64s %4s avg %7.1f us  min %7.1f max %7.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
tail -1 $R/gpurun_out/pk.log
