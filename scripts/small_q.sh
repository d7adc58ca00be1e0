# per-step time and scoring-launch time for small query batches (online: Q=1, rOxford test set: Q=70)
for q in ${QS:-1 16 70 128}; do timeout -k 10 200 python bench.py --scale-10m off --steps 20 --warmup 3 --no-cpu-baseline --queries $q 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('Q=$q', 'ms/step=%.3f'%j['ms_per_step'], 'score launch ms=%.3f'%(r['avg_launch_ms']), 'share=%.2f'%r['kernel_share_of_step'], 'image GB/s=%.0f'%(1005994*2048*2/r['avg_launch_ms']/1e6))
"; done
