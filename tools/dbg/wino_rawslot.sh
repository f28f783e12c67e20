# (the library carries the digest of its compile flags: the flags stay exported for the runs, and the default build is restored on exit)
trap 'unset PTOCR_EXTRA_HIPCC_FLAGS; python -m pytorchocr_amd.build > /dev/null 2>&1' EXIT
for slot in 0 7 5; do
  export PTOCR_EXTRA_HIPCC_FLAGS=-DW4_RAW_SLOT=$slot; python -m pytorchocr_amd.build > gpurun_out/rs_build_$slot.log 2>&1 || { tail -5 gpurun_out/rs_build_$slot.log; exit 1; }
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-embed --cpu-images 0 --crnn-steps 0 --post-input none 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('raw slot $slot', d['value'], d['ms_per_step'], d['roofline']['all_conv']['ms_per_step'], d['roofline']['kernel'][105:140])
" >> gpurun_out/rawslot.log
  done
done
cat gpurun_out/rawslot.log
