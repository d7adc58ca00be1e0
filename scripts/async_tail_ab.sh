# same box, alternating: the synchronous tail (0) against the asynchronous placements (1: beside the next batch's bootstrap
# and scoring launch, 2: beside its bootstrap only, 3: deferred -- beside its scoring launch only)
for rep in 1 2; do for m in ${MODES:-0 3 1 2}; do python bench.py --scale-10m off --no-cpu-baseline --async-tail $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('async_tail', $m, 'q/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'launch %.4f' % r['avg_launch_ms'], 'clock %.0f' % r['in_kernel_clock_mhz'])
"; done; done
