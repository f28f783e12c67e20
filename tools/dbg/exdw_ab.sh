#!/bin/bash
# GPU box: bf16 tests, then the bf16 detector line with the expansion+depthwise fusion off / on, then the per-kernel trace of one forward.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -m pytest tests/test_gpu_bf16.py -x -q > gpurun_out/exdw_tests.log 2>&1 || { tail -30 gpurun_out/exdw_tests.log; exit 1; }
tail -2 gpurun_out/exdw_tests.log
for f in 0 1; do
  PTOCR_BF16_EXDW_FUSE=$f python3 bench.py --det-model mbv3s --dtype bf16 --steps 60 --warmup 10 --crnn-steps 0 --cpu-images 0 > gpurun_out/exdw_$f.log 2>&1
  python3 - <<PY
import json
l = [x for x in open("gpurun_out/exdw_$f.log") if x.startswith("{")][-1]
j = json.loads(l)
print("fuse=$f", j["value"], j["ms_per_step"], json.dumps(j.get("roofline", {}))[:400])
PY
done
bash tools/dbg/bf16_trace.sh
