# bootstrap sample size for small batches (streaming kernel), ratio limit lifted: bash scripts/smallq_sample_sweep.sh
for q in 1 70; do for c in 32 16 8; do
  timeout -k 10 120 python bench.py --scale-10m off --no-cpu-baseline --queries $q --steps 100 --option chunk0_tiles=$c --option spec_max_ratio=2048 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']; print('Q=$q sample tiles=$c ms/step=%.4f launch=%.4f launches=%d survivors/q=%.0f' % (j['ms_per_step'], r['avg_launch_ms'], r['launches'], j['config']['survivors_per_query']))"
done; done
