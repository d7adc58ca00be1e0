# lookahead A/B on one box, alternating, deferred tail: off / on
for rep in 1 2 3; do for cfg in "off:--option ladder=1" "on:--lookahead --option ladder=1"; do
name=${cfg%%:*}; a=${cfg#*:}
python bench.py --scale-10m off --no-cpu-baseline --async-tail 3 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lookahead $name: %.0f q/s  ms/step %.4f  launch %.4f  share %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['kernel_share_of_step']))
"; done; done
