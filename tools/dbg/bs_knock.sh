#!/bin/bash
# GPU box: border_states_kernel with parts compiled out (-DBS_DBG: 1 no straight stretches, 2 no generic pixels);
# results wrong by design (the later kernels see other borders: only this kernel's time means anything).  Stress maps (tools/bench_post.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in ${VARIANTS:-0 1 2 3}; do
  PTOCR_EXTRA_HIPCC_FLAGS="-DBS_DBG=$v" python3 -m pytorchocr_amd.build > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sck
  PTOCR_EXTRA_HIPCC_FLAGS="-DBS_DBG=$v" timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sck -- python3 $R/tools/bench_post.py 10 > /tmp/sck.log 2>&1
  cd $R
  python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/sck/*/*kernel_stats.csv")[0])):
    if "border_states" in r["Name"]: print("BS_DBG=$v %-30s %8.1f us" % (r["Name"].split("(")[0][-30:], float(r["AverageNs"])/1e3))
PY
done
