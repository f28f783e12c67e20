#!/bin/bash
# GPU box: CRNN tests + line with the LSTM cell's libm activations (-DLSTM_FAST_ACT=0) and the v_exp / v_rcp forms (default)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
trap 'PTOCR_EXTRA_HIPCC_FLAGS= python3 -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for v in "-DLSTM_FAST_ACT=0" ""; do
  export PTOCR_EXTRA_HIPCC_FLAGS="$v"
  python3 -m pytorchocr_amd.build > /dev/null 2>&1
  echo "== [$v]"
  PTOCR_LSTM_FUZZ=40 python3 -m pytest tests/test_gpu_crnn.py tests/test_gpu_ocr_batch.py -x -q 2>&1 | tail -1
  python3 bench.py --workload crnn --steps 40 --warmup 10 --cpu-lines 0 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lines/s', j['value'], 'ms', j['ms_per_step'])"
  bash tools/dbg/crnn_trace.sh | grep "lstm_bidir_split"
done
