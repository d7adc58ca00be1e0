mkdir -p gpurun_out/r06c && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_tile_kernel_parity.py tests/test_gpu_knn_parity.py tests/test_gpu_shape_sweep.py -q -x -m gpu > gpurun_out/r06c/tests.txt 2>&1; echo tests rc=$?
timeout -k 10 300 python scripts/hard_data_probe.py clustered > gpurun_out/r06c/hard_probe.txt 2>&1
bash scripts/ab.sh ab/lib_nospill.so ab/lib_spill.so --extra-blocks off --steps 100 > gpurun_out/r06c/ab_spill.txt 2>&1
bash scripts/ab.sh ab/lib_nospill.so ab/lib_spill.so --extra-blocks off --steps 100 --queries 70 >> gpurun_out/r06c/ab_spill.txt 2>&1
tail -4 gpurun_out/r06c/tests.txt; cat gpurun_out/r06c/hard_probe.txt | tail -12; cat gpurun_out/r06c/ab_spill.txt
