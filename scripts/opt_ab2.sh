# A/B of mi_set_option values with bench.py on ONE box, alternating: bash scripts/opt_ab2.sh name v1 v2 [rounds]
name=$1; a=$2; b=$3; rounds=${4:-2}
for r in $(seq $rounds); do for v in $a $b; do
  timeout -k 10 200 python bench.py --scale-10m off --no-cpu-baseline --option $name=$v | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('$name=$v ms/step=%.3f launch ms=%.3f TF=%.0f clock=%s share=%.3f'%(j['ms_per_step'], r['avg_launch_ms'], r['achieved'], r.get('in_kernel_clock_mhz'), r['kernel_share_of_step']))
"
done; done
