# same-PROCESS A/B of the gallery ingest kernels on the same buffers: ab/ingest_old.hip (a kept copy of csrc/ingest.hip, compiled
# with -Dmi=mi_old) against the tree's, launches interleaved.  (Between two processes the same kernel differs by 10-15 %: where
# the driver put the buffers.)  Usage on the GPU box: bash scripts/ingest_ab.sh [process runs]
set -e
cd $GRAFT_REPO_ROOT
# the baseline is a file you keep yourself (ab/ is git-ignored and travels with gpurun): e.g.
#   git show <commit>:image-search-engine-for-historical-research_amd/csrc/ingest.hip > ab/ingest_old.hip
[ -f ab/ingest_old.hip ] || { echo "ab/ingest_old.hip is missing (see the comment in this script)"; exit 2; }
P=image-search-engine-for-historical-research_amd/build
mkdir -p $P
C=image-search-engine-for-historical-research_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=128 -Dmi=mi_old -I $C -c ab/ingest_old.hip -o $P/ingest_old.o 2> /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=128 -DMI_INGEST_AB -c scripts/ingestbench.hip -o $P/ingestbench_ab.o 2> /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 $P/ingestbench_ab.o $P/ingest_old.o -o $P/ingestbench_ab
for r in $(seq 1 ${1:-3}); do
  for l in rows cols; do
    timeout -k 10 60 $P/ingestbench_ab 1005994 2048 $l 2>&1 | grep -E 'probe|copy' | awk '{print $3, $8}' | sort | awk '{a[$1] = a[$1] " " $2} END {for (k in a) print k, a[k]}'
  done
done
