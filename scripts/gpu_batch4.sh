set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bench_cli.py tests/test_gpu_knn_parity.py tests/test_gpu_rerank_and_shards.py tests/test_gpu_sharded_ranks.py tests/test_gpu_full_size.py -m gpu -x -q > $o/r04b_tests.txt 2>&1 || (tail -40 $o/r04b_tests.txt; exit 1)
echo tests done
python bench.py > $o/r04b_bench.json 2> $o/r04b_bench.err || (tail -30 $o/r04b_bench.err; exit 1)
echo bench done
