timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for cfg in "chunk0_tiles=32" "chunk0_tiles=64 --option survivor_cap=16384"; do for rows in 0 125750; do
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rows $rows --option $cfg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('$cfg rows=$rows', 'ms/step=%.3f'%j['ms_per_step'], 'q/s=%.0f'%j['value'], 'gemm TF=%.0f'%r['achieved'], 'surv/q=%.0f'%(j['config']['survivors_per_query']))
    elif 'rror' in l or 'invalid' in l: print(l.strip()[:300])
"; done; done
