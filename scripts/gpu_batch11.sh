set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bench_cli.py -m gpu -x -q > $o/r04g_tests.txt 2>&1 || (tail -40 $o/r04g_tests.txt; exit 1)
tail -2 $o/r04g_tests.txt
(ISEHR_DIST_BACKEND=gloo ISEHR_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 4 --rows 400000 --steps 6 --warmup 2 --no-cpu-baseline --multi-gpu-blocks on --scale-10m on --scale-10m-rows 1200000 --scale-10m-steps 4 2> /dev/null | tail -1 | cut -c1-8000) > $o/r04g_bare_gpus4.txt
python -c "
import json
d=json.loads(open('gpurun_out/r04g_bare_gpus4.txt').read())
print(d['batch_replicas'])
"
