#!/bin/bash
# GPU box: bf16 tests, then the bf16 detector line with the FPN output as one concat buffer / as four planes, then the kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -m pytest tests/test_gpu_bf16.py -x -q > gpurun_out/planes_tests.log 2>&1 || { tail -30 gpurun_out/planes_tests.log; exit 1; }
tail -2 gpurun_out/planes_tests.log
for f in 0 1; do
  PTOCR_BF16_FUSE_PLANES=$f python3 bench.py --det-model mbv3s --dtype bf16 --steps 60 --warmup 10 --crnn-steps 0 --cpu-images 0 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('planes=$f img/s', j['value'], 'ms', j['ms_per_step'])"
done
bash tools/dbg/bf16_trace.sh | grep "pers8\|lat_bf16\|head_tail\|sum\|stem"
