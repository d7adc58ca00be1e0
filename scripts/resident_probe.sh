# What slows the scoring launch when the re-score runs beside it (async_tail 3)?  MI_RESIDENT_DEBUG (diagnostics in
# launch_rescore_resident): 1 = loads without the f64 arithmetic, 2 = arithmetic without the gather (64 cache-resident rows),
# 3 = both, 4 = a quarter of the CUs host the tail.  Results are wrong by construction for 1-3: --diagnostic.
# The switch exists only in a probe build of the library: this script rebuilds select.hip with -DMI_RESIDENT_PROBE and
# restores the product build when it is done.
pkg=image-search-engine-for-historical-research_amd
HIPCC_EXTRA="-DMI_RESIDENT_PROBE" python -c "import sys; sys.path.insert(0, '$pkg'); import build; build.build(force=True)"
trap 'python -c "import sys; sys.path.insert(0, \"$pkg\"); import build; build.build(force=True)"' EXIT
for rep in 1 2; do
for cfg in "0 0" "3 0" "3 1" "3 2" "3 3" "3 4"; do set -- $cfg
MI_RESIDENT_DEBUG=$2 python bench.py --scale-10m off --no-cpu-baseline --diagnostic --async-tail $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('async_tail', $1, 'resident_debug', $2, 'ms/step %.4f' % d['ms_per_step'], 'launch %.4f' % r['avg_launch_ms'], 'clock %.0f' % r['in_kernel_clock_mhz'])
"; done; done
