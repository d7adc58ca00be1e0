# Usage (on the GPU box): bash scripts/profile_round.sh <tag>   -> writes gpurun_out/<tag>_*
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_l2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_l2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_sq -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_sq.log 2>&1 || true
tail -1 gpurun_out/${tag}_bench.json | cut -c1-600
