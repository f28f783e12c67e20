# (the library carries the digest of its compile flags: the flags stay exported for the runs, and the default build is restored on exit)
trap 'unset PTOCR_EXTRA_HIPCC_FLAGS; python -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
R=$GRAFT_REPO_ROOT
for dbg in 16 8 40; do
  export PTOCR_EXTRA_HIPCC_FLAGS=-DW4_DBG=$dbg; python -m pytorchocr_amd.build > gpurun_out/wk_build_$dbg.log 2>&1 || { tail -5 gpurun_out/wk_build_$dbg.log; exit 1; }
  for m in 0 1; do
  PTOCR_WINO_SPLIT=$m python bench.py --steps 10 --warmup 3 --no-embed --cpu-images 0 --crnn-steps 0 --post-input none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('dbg $dbg split $m', d['value'], d['ms_per_step'], d['roofline']['all_conv']['ms_per_step'], d['roofline']['kernel'][105:140])
" >> gpurun_out/wsplit_knock.log
  done
done
cat gpurun_out/wsplit_knock.log
