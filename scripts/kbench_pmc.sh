# Fabric traffic and L2 hit rate PER kbench VARIANT (every variant is its own template instantiation, so the counter rows of
# rocprofv3 carry its name).  Separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
# Usage on the GPU box: bash scripts/kbench_pmc.sh <tag> variant...      (kbench built beforehand: scripts/kbench_build.sh)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out
cd image-search-engine-for-historical-research_amd
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/${tag}_kb_fetch -- ./build/kbench --rounds 1 --reps 3 "$@" > $o/${tag}_kb_fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $o/${tag}_kb_l2 -- ./build/kbench --rounds 1 --reps 3 "$@" > $o/${tag}_kb_l2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/${tag}_kb_write -- ./build/kbench --rounds 1 --reps 3 "$@" > $o/${tag}_kb_write.log 2>&1
cd ..
python scripts/kbench_pmc_report.py $tag > $o/${tag}_kbench_pmc.txt
cat $o/${tag}_kbench_pmc.txt
