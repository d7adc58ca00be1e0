# What the ragged last round of gallery tiles costs the scoring launch (VERDICT r04 #7a): 1 005 994 rows = 3930 tiles x 4 query
# tiles = 61.4 rounds of the 256 workgroups; 999 424 rows = 3904 tiles = 61 rounds exactly.  The XCD shares are dealt in whole
# rounds, so the 0.4 round is the quantum the balance between XCDs works with.  Same box, alternating; TFLOP/s of the undisturbed
# (synchronous) launches is per row, so the two sizes compare directly: the difference is the most a finer quantum could buy.
for rep in 1 2 3; do for rows in 999424 1005994; do
python bench.py --rows $rows --steps 100 --warmup 20 --scale-10m off --no-cpu-baseline --extra-blocks off --async-tail 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('rows $rows: launch %.4f ms  %.1f TFLOP/s  frac %.4f  clock %.0f MHz  frac_at_clock %.4f' % (r['avg_launch_ms'], r['achieved'], r['frac'], r['in_kernel_clock_mhz'], r['frac_at_clock']))"
done; done
