# The whitening kernel of commit ea3c756 (one LDS buffer per operand, two barriers per K chunk, the means fetched where they are
# subtracted) against the current one (two buffers, one barrier, means fetched with the chunk), same box, alternating, kernel
# durations from rocprofv3 --kernel-trace --stats.  Record: profiles/r06v_whiten_kernel_ab.txt.
#   bash scripts/whiten_kernel_ab.sh [build]     (`build`: only builds ab/lib_whiten_v1.so -- here, where the git history is)
set -e
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
pkg=$root/image-search-engine-for-historical-research_amd
if [ ! -f $root/ab/lib_whiten_v1.so ]; then
tmp=$(mktemp -d)
mkdir -p $tmp/pkg $tmp/include
cp -r $pkg/csrc $tmp/pkg/csrc; cp $root/include/mi355_retrieval.h $tmp/include/
git -C $root show ea3c756:image-search-engine-for-historical-research_amd/csrc/whiten.hip > $tmp/pkg/csrc/whiten.hip
cd $tmp/pkg/csrc
for f in *.hip; do echo "/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -c $f -o ${f%.hip}.o"; done | xargs -P 8 -I{} sh -c '{}'
mkdir -p $root/ab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/ab/lib_whiten_v1.so *.o
fi
[ "$1" = "build" ] && exit 0
cd $root
export TMPDIR=/tmp
out=$root/gpurun_out/whiten_kernel_ab.txt; mkdir -p $root/gpurun_out; : > $out
for v in v1 v2 v1 v2; do
  lib=$root/ab/lib_whiten_v1.so
  [ $v = v2 ] && lib=$pkg/libmi355_retrieval.so
  rm -rf /tmp/prof_$v
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $root/scripts/whiten_ab.py $lib 131072 > /tmp/whiten_$v.txt 2>&1)
  echo "== $v ($lib)" >> $out
  head -4 "$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)" >> $out
done
cat $out
