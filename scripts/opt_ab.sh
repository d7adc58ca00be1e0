# A/B of runtime options on ONE box: bash scripts/opt_ab.sh "" "chunk0_tiles=48" ...
for rep in 1 2; do for o in "$@"; do
  if [ -z "$o" ]; then extra=""; else extra="--option $o"; fi
  timeout -k 10 100 python bench.py --no-cpu-baseline $extra | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('[$o] ms/step=%.3f launch ms=%.3f TF=%.0f surv/q=%.0f'%(j['ms_per_step'], r['avg_launch_ms'], r['achieved'], j['config']['survivors_per_query']))
"
done; done
