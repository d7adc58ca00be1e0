set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/r04j_gputests.txt 2>&1 || (tail -40 $o/r04j_gputests.txt; exit 1)
tail -2 $o/r04j_gputests.txt
python bench.py --steps 20 --warmup 5 > $o/r04j_bench.json 2> $o/r04j_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04j_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.0f | frac %.4f first5 %.4f last5 %.4f | sync %.0f | map diff %s | aqe baseline %.0f' % (d['value'], r['frac'], r['launch_ms_first5'], r['launch_ms_last5'], d['synchronous']['value'], [m['max_abs_map_difference'] for m in d['map']], d['cpu_baseline']['aqe']['value']))
PY
