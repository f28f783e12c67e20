#!/bin/bash
# GPU box: det forward per-launch trace for a list of build variants; prints every non-Winograd launch and the Winograd sum.  usage: det_trace_fam.sh "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for FLAGS in "$@"; do
  echo "== variant: '${FLAGS}'"
  PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS" python -m pytorchocr_amd.build > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  export PTOCR_EXTRA_HIPCC_FLAGS="$FLAGS"
  bash tools/dbg/det_trace.sh > gpurun_out/det_trace_ab.tmp 2>&1 || { tail -5 gpurun_out/det_trace_ab.tmp; exit 1; }
  python3 - <<'PY'
import re
w = 0.0; out = []
for l in open("gpurun_out/det_trace.txt"):
    m = re.search(r"dur\s+([\d.]+) us\s+grid \S+\s+(.*)$", l)
    if not m: print(l.strip()); continue
    if "wino4r" in m.group(2): w += float(m.group(1))
    else: out.append("%s %.0f" % (re.sub(r"<.*|ptocr::|void ", "", m.group(2)).strip()[:14], float(m.group(1))))
print("wino %.0f | " % w + "  ".join(out))
PY
done
