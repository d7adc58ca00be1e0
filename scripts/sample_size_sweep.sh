# step time against the size of the threshold sample (chunk0_tiles x 256 rows): bash scripts/sample_size_sweep.sh
for rows in 125750 251500 1005994; do
  for c in 4 8 16 32 64; do
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --rows $rows --option chunk0_tiles=$c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows', $rows, 'chunk0_tiles', $c, 'ms_per_step %.4f' % d['ms_per_step'], 'survivors %.0f' % d['config']['survivors_per_query'], 'launch_ms %.4f' % d['roofline']['avg_launch_ms'])
"
  done
done
