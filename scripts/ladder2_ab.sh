# one vs two ladder levels, same box, alternating (synchronous tail: --option switches the headline to it)
for rep in 1 2 3; do for lad in 1 2; do python bench.py --scale-10m off --no-cpu-baseline --option ladder=$lad 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ladder $lad: ms/step %.4f launch %.4f TF %.0f survivors/query %.0f candidates %.1f' % (d['ms_per_step'], r['avg_launch_ms'], r['achieved'], d['config']['survivors_per_query'], d['config']['candidates_per_query']))
"; done; done
