for dbg in ${DBGS:-0 4}; do timeout -k 10 120 python bench.py --scale-10m off --steps 5 --warmup 1 --no-cpu-baseline --diagnostic --option debug=$dbg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('debug=$dbg', 'ms/step=%.3f'%j['ms_per_step'], 'gemm TF=%.0f'%j['roofline']['achieved'], 'avg_launch_ms=%.3f'%j['roofline']['avg_launch_ms'])
    elif 'invalid' in l or 'Error' in l: print(l.strip()[:200])
"; done
