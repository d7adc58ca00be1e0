timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for dt in f16 bf16; do timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --image-dtype $dt 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('$dt', 'ms/step=%.3f'%j['ms_per_step'], 'q/s=%.0f'%j['value'], 'gemm TF=%.0f'%r['achieved'], 'share=%.2f'%r['kernel_share_of_step'], 'cand/q=%.0f surv/q=%.0f'%(j['config']['candidates_per_query'], j['config']['survivors_per_query']))
    elif 'rror' in l or 'invalid' in l: print(l.strip()[:300])
"; done
