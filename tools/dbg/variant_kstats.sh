#!/bin/bash
# GPU box: per-kernel times of the post-process on the scene checkpoint's own maps under compile-time variants.
# usage: VARIANTS="flags1|flags2" KERNELS="border_states|scatter" variant_kstats.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
IFS='|' read -ra VS <<< "${VARIANTS:-|}"
for v in "${VS[@]}"; do
  PTOCR_EXTRA_HIPCC_FLAGS="$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/vk
  for which in model text-like; do
    ONLY=$which PTOCR_EXTRA_HIPCC_FLAGS="$v" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vk_$which -- python3 $R/tools/dbg/scene_post.py r18 > /tmp/vk.log 2>&1
  done
  cd $R
  python3 - <<PY
import csv,glob,re
pat = re.compile("${KERNELS:-border_states|scatter_states|border_stage}")
for which in ("model", "text-like"):
    out=[]
    for r in csv.DictReader(open(glob.glob("/tmp/vk_%s/*/*kernel_stats.csv" % which)[0])):
        if pat.search(r["Name"]) and "conv" not in r["Name"]: out.append("%s %.1f" % (r["Name"].split("(")[0].split("::")[-1][:24], float(r["AverageNs"])/1e3))
    print("[$v] %-9s %s" % (which, "; ".join(out)))
PY
  rm -rf /tmp/vk_model /tmp/vk_text-like
done
