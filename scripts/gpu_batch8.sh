set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tile_kernel_parity.py tests/test_gpu_knn_parity.py tests/test_gpu_rerank_and_shards.py tests/test_gpu_sharded_ranks.py -m gpu -x -q > $o/r04e_tests.txt 2>&1 || (tail -40 $o/r04e_tests.txt; exit 1)
tail -2 $o/r04e_tests.txt
for a in "" "--lookahead" "" "--lookahead"; do python bench.py --scale-10m off --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; sy=d['synchronous']
print('lookahead' if d['config']['lookahead'] else 'no lookahead', 'pipelined %.0f q/s %.4f ms launch %.4f | sync %.0f q/s %.4f ms launch %.4f' % (d['value'], d['ms_per_step'], r['pipelined']['avg_launch_ms'], sy['value'], sy['ms_per_step'], r['avg_launch_ms']))
"; done | tee $o/r04e_lookahead_ab.txt
