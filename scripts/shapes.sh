for rows in 1005994 502997 251499 125750; do timeout -k 10 200 python bench.py --scale-10m off --steps 10 --warmup 2 --no-cpu-baseline --rows $rows 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('rows=$rows', 'ms/step=%.3f'%j['ms_per_step'], 'gemm ms=%.3f'%(r['avg_launch_ms']), 'share=%.2f'%r['kernel_share_of_step'], 'cand/q=%.0f'%j['config']['candidates_per_query'])
"; done
