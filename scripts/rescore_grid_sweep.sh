# re-score grid width (workgroups per query and sweep) against step time, single shard and a 125 750-row shard
for gx in 0 8 16 32; do
  for rows in 1005994; do
    python bench.py --scale-10m off --steps 60 --warmup 5 --no-cpu-baseline --rows $rows --option rescore_grid_x=$gx 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows', $rows, 'rescore_grid_x', $gx, 'ms_per_step %.4f' % d['ms_per_step'], 'launch_ms %.4f' % d['roofline']['avg_launch_ms'], 'rest %.4f' % (d['ms_per_step']-d['roofline']['avg_launch_ms']))
"
  done
done
