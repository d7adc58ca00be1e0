# The use-after-recycle scenario of tests/_hazard.py against (a) a build of the CURRENT sources minus the device synchronisation
# in ws_free (what the library was before commit 87f05b2) and (b) the product library: (a) must report disturbed rounds, (b) none.
#   bash scripts/recycle_hazard_ab.sh [build]     (`build`: only builds ab/lib_nosync.so -- hipcc cross-compiles without a GPU)
set -e
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(pwd)
pkg=$root/image-search-engine-for-historical-research_amd
if [ ! -f $root/ab/lib_nosync.so ] || [ $pkg/csrc/api_state.hip -nt $root/ab/lib_nosync.so ]; then
tmp=$(mktemp -d)
mkdir -p $tmp/pkg $tmp/include
cp -r $pkg/csrc $tmp/pkg/csrc; cp $root/include/mi355_retrieval.h $tmp/include/      # api_internal.h includes ../../include/
cd $tmp/pkg/csrc
python3 - <<'PY'
s = open("api_state.hip").read()
a = s.index("int ws_free(Workspace& ws) {")
b = s.index("(void)hipDeviceSynchronize();", a)
assert b - a < 700, "the synchronisation is not where ws_free has it"
s = s[:b] + "/* demo build: no synchronisation */" + s[b + len("(void)hipDeviceSynchronize();"):]
open("api_state.hip", "w").write(s)
PY
for f in *.hip; do echo "/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -c $f -o ${f%.hip}.o"; done | xargs -P 8 -I{} sh -c '{}'
mkdir -p $root/ab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/ab/lib_nosync.so *.o
fi
[ "$1" = "build" ] && exit 0
cd $root
dst=$pkg/libmi355_retrieval.so
cp $dst /tmp/lib_keep_hazard.so
for lib in ab/lib_nosync.so /tmp/lib_keep_hazard.so; do
  cp $lib $dst
  echo "== $lib"
  timeout -k 10 200 python tests/_hazard.py || true
done
cp /tmp/lib_keep_hazard.so $dst
