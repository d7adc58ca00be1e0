# K-split bootstrap launch for small batches (option boot_ksplit) on / off, same box, alternating
for rep in 1 2 3; do for q in ${QS:-1 70 128 300}; do for ks in 0 1; do
python bench.py --scale-10m off --no-cpu-baseline --queries $q --steps 400 --option boot_ksplit=$ks 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('queries', $q, 'boot_ksplit', $ks, 'ms/step %.4f' % d['ms_per_step'], 'launch %.4f' % r['avg_launch_ms'], 'q/s %.0f' % d['value'])
"; done; done; done
