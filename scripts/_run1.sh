mkdir -p gpurun_out/r06b && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rerank_and_shards.py -q -x -m gpu -k "whiten" > gpurun_out/r06b/t_whiten.txt 2>&1; echo whiten-tests rc=$?
python -m pytest tests/test_gpu_secondary_sweep.py -q -x -m gpu -k "whiten" >> gpurun_out/r06b/t_whiten.txt 2>&1; echo sweep rc=$?
python scripts/whiten_ab.py image-search-engine-for-historical-research_amd/libmi355_retrieval.so 65536 > gpurun_out/r06b/whiten_ab.txt 2>&1
python scripts/whiten_ab.py ab/lib_r05.so 65536 >> gpurun_out/r06b/whiten_ab.txt 2>&1
python bench.py --steps 20 --warmup 5 --blocks whiten --no-cpu-baseline > gpurun_out/r06b/bench_whiten.json 2> gpurun_out/r06b/bench_whiten.err; echo bench rc=$?
timeout -k 10 300 python scripts/hard_data_probe.py clustered > gpurun_out/r06b/hard_probe.txt 2>&1
tail -5 gpurun_out/r06b/t_whiten.txt; cat gpurun_out/r06b/whiten_ab.txt; tail -3 gpurun_out/r06b/bench_whiten.err; grep -A1 q1024 gpurun_out/r06b/hard_probe.txt
