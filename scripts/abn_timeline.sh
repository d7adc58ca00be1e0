# per-kernel durations for several builds of the library on ONE box: bash scripts/abn_timeline.sh <kernel-substring> lib1.so lib2.so ...
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
dst=image-search-engine-for-historical-research_amd/libmi355_retrieval.so
cp $dst /tmp/lib_keep.so
for lib in "$@"; do
  cp $lib $dst
  rm -rf gpurun_out/abtl
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abtl -- python3 bench.py --scale-10m off --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/abtl.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/abtl/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "$pat" in r["Name"]: print("$lib", r["Name"].split("(")[0][-40:], "avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
done
cp /tmp/lib_keep.so $dst
