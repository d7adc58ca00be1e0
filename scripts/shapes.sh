timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --with-aqe 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('with alpha-QE', 'ms/step=%.3f'%j['ms_per_step'], 'q/s=%.0f'%j['value'], 'gemm TF=%.0f'%r['achieved'], 'launches', r['launches'])
    elif 'rror' in l or 'invalid' in l: print(l.strip()[:300])
"
