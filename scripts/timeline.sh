# per-dispatch timeline of one query batch: bash scripts/timeline.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/${tag}_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_tl -- python3 bench.py --scale-10m off --steps 3 --warmup 1 --no-cpu-baseline --diagnostic "$@" > gpurun_out/${tag}_tl.log 2>&1
python scripts/timeline.py gpurun_out/${tag}_tl
