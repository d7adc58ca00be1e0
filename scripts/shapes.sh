cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rows in 125750 251499 502997; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s$rows -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --rows $rows > gpurun_out/prof_s$rows.log 2>&1; tail -1 gpurun_out/prof_s$rows.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('rows=$rows ms/step=%.3f'%j['ms_per_step'])
"; done
