# Supplement of scripts/profile_record.sh for the small-batch path only (the reference's own batch sizes: 1 and 70 queries):
# bench lines with rocprofv3 kernel statistics, per-dispatch timelines and the same-box A/B against the previous round's
# library.  Usage on the GPU box: bash scripts/profile_small_batches.sh <tag>
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
for v in "q1:--queries 1" "q70:--queries 70" "q128:--queries 128"; do
  name=${v%%:*}; args=${v#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_${name}_trace -- python3 bench.py --scale-10m off --no-cpu-baseline $args > $o/${tag}_${name}_bench.json 2> $o/${tag}_${name}.err || echo "$name failed"
done
bash scripts/timeline.sh ${tag}_q1 --queries 1 > $o/${tag}_timeline_q1.txt 2>&1 || true
bash scripts/timeline.sh ${tag}_q70 --queries 70 > $o/${tag}_timeline_q70.txt 2>&1 || true
if [ -f ab/lib_r02.so ]; then
  cp image-search-engine-for-historical-research_amd/libmi355_retrieval.so ab/lib_now.so
  (for a in "--queries 70" "--queries 1" ""; do echo "# bench.py $a"; bash scripts/ab.sh ab/lib_r02.so ab/lib_now.so $a 2> /dev/null; done) > $o/${tag}_ab_r02.txt || true
fi
echo "small batches done"
