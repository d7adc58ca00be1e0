# A/B/... of several builds of the library on ONE box: bash scripts/abn.sh lib1.so lib2.so ... (2 rounds each)
dst=image-search-engine-for-historical-research_amd/libmi355_retrieval.so
cp $dst /tmp/lib_keep.so
for rep in 1 2; do for lib in "$@"; do
  cp $lib $dst
  timeout -k 10 100 python bench.py --scale-10m off --no-cpu-baseline | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('$lib ms/step=%.3f launch ms=%.3f TF=%.0f'%(j['ms_per_step'], r['avg_launch_ms'], r['achieved']))
"
done; done
cp /tmp/lib_keep.so $dst
