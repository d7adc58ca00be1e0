# step time against the size of the threshold sample (chunk0_tiles x 256 rows), ratio limit lifted so that every size takes
# the single-launch schedule with the ladder: bash scripts/sample_size_sweep.sh [rows ...]
for rows in ${@:-1005994}; do
  for rep in 1 2; do for c in 4 8 16 32; do
    python bench.py --scale-10m off --steps 100 --warmup 5 --no-cpu-baseline --rows $rows --option spec_max_ratio=2048 --option chunk0_tiles=$c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows', $rows, 'sample', $c * 256, 'ms_per_step %.4f' % d['ms_per_step'], 'survivors %.0f' % d['config']['survivors_per_query'], 'launch_ms %.4f' % d['roofline']['avg_launch_ms'], 'rest %.4f' % (d['ms_per_step'] - d['roofline']['avg_launch_ms']))
"
  done; done
done
