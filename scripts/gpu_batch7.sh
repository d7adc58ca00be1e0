set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tile_kernel_parity.py -m gpu -x -q > $o/r04d_tests.txt 2>&1 || (tail -40 $o/r04d_tests.txt; exit 1)
tail -2 $o/r04d_tests.txt
bash scripts/profile_record.sh r04c "7"
cat $o/r04c_sweep_seeds.txt
