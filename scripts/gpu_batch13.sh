set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $o/r04i_driver_style.json 2> $o/r04i_driver_style.err

echo "wall seconds: $SECONDS"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04i_driver_style.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.0f ms %.4f | roofline frac %.4f launch %.4f first5 %.4f last5 %.4f | pipelined launch %.4f share %.3f | sync %.0f' % (d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r.get('launch_ms_first5',0), r.get('launch_ms_last5',0), r['pipelined']['avg_launch_ms'], r['pipelined']['kernel_share_of_step'], d['synchronous']['value']))
print('10m', d['scale_10m']['value'], 'map', [(m['map'], m['max_abs_map_difference']) for m in d['map']])
PY
