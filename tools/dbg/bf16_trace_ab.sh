#!/bin/bash
# GPU box: bf16 mbv3s forward per-launch trace for a list of build variants; prints per-kernel-family sums.  usage: bf16_trace_ab.sh "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  echo "== variant: '${FLAGS}'"
  PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS" python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  bash tools/dbg/bf16_trace.sh > gpurun_out/bf16_trace_ab.tmp 2>&1 || { tail -5 gpurun_out/bf16_trace_ab.tmp; exit 1; }
  cp gpurun_out/bf16_trace.txt gpurun_out/bf16_trace_v$i.txt
  python3 - <<'PY'
import re, collections
fam = collections.OrderedDict()
for l in open("gpurun_out/bf16_trace.txt"):
    m = re.search(r"dur\s+([\d.]+) us\s+grid \S+\s+(.*)$", l)
    if not m: print(l.strip()); continue
    name = re.sub(r"<.*", "", m.group(2).replace("void ", "").strip())
    fam[name] = fam.get(name, 0.0) + float(m.group(1))
print("  ".join("%s %.0f" % (k.replace("ptocr::", ""), v) for k, v in fam.items()))
PY
done
