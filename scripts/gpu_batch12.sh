set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_tile_kernel_parity.py -m gpu -x -q > $o/r04h_tests.txt 2>&1 || (tail -40 $o/r04h_tests.txt; exit 1)
tail -2 $o/r04h_tests.txt
(for sd in 201 202 203 204 205 206 207 208 209 210 211 212; do ISEHR_SWEEP_SEED=$sd timeout -k 10 300 python -m pytest tests/test_gpu_shape_sweep.py tests/test_gpu_entry_sweep.py tests/test_gpu_secondary_sweep.py -q 2>&1 | tail -1; done) > $o/r04h_sweep_seeds.txt
cat $o/r04h_sweep_seeds.txt
