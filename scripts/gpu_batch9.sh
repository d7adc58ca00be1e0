set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/la_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/la_tl -- python3 bench.py --scale-10m off --steps 6 --warmup 2 --no-cpu-baseline --diagnostic --async-tail 3 > gpurun_out/la_tl.log 2>&1
python scripts/trace_tail.py gpurun_out/la_tl 45
