set -e
for m in both stress none both stress none; do
  python bench.py --steps 20 --warmup 5 --no-embed --cpu-images 0 --post-input $m 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$m', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), (d.get('roofline_post') or {}).get('ms_per_call_overlapped'))
" >> gpurun_out/ab_post_input.log
done
cat gpurun_out/ab_post_input.log
