R=$GRAFT_REPO_ROOT
PTOCR_WINO_SPLIT=1 python -m pytest tests/test_gpu_det_model.py tests/test_gpu_conv.py -x -q > gpurun_out/wsplit_tests.log 2>&1
tail -15 gpurun_out/wsplit_tests.log
for m in 0 1 0 1; do
  PTOCR_WINO_SPLIT=$m python bench.py --steps 20 --warmup 5 --no-embed --cpu-images 0 --crnn-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('split $m', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['all_conv']['ms_per_step'], d['roofline']['kernel'][:160])
" >> gpurun_out/wsplit_ab.log
done
cat gpurun_out/wsplit_ab.log
