for rep in 1 2; do
for cfg in "f16 0" "f16 3" "bf16 0" "bf16 3" "bf16 1"; do set -- $cfg
python bench.py --scale-10m off --no-cpu-baseline --image-dtype $1 --async-tail $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'async_tail', $2, 'q/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'launch %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'clock %.0f' % r['in_kernel_clock_mhz'], 'cand %.0f' % d['config']['candidates_per_query'])
"; done; done
