for k in 10 100 1000; do timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --topk $k 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('K=$k', 'ms/step=%.3f'%j['ms_per_step'], 'q/s=%.0f'%j['value'], 'gemm TF=%.0f'%r['achieved'], 'cand/q=%.0f'%(j['config']['candidates_per_query']))
    elif 'rror' in l or 'invalid' in l: print(l.strip()[:300])
"; done
