# builds the standalone kernel A/B driver (its own -DMI_KBENCH build of the scoring kernels) and the probes: bash scripts/kbench_build.sh
set -e
cd "$(dirname "$0")/.."
pkg=image-search-engine-for-historical-research_amd
python -c "import sys; sys.path.insert(0, '.'); import __graft_entry__ as g; g.build()"
# kbench is a program of its own: the tile kernel with every diagnostic / A-B instantiation (-DMI_KBENCH; the product library
# holds six instantiations and none of these) compiled INTO it -- not linked against libmi355_retrieval.so, whose kernels of
# the same mangled names would otherwise be the ones that run
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_KBENCH -Wno-unused-value -Wno-unused-result scripts/kbench.hip \
  $pkg/csrc/gemm_select.hip $pkg/csrc/stream_select.hip $pkg/csrc/select.hip -o $pkg/build/kbench
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 scripts/mfma_probe.hip -o $pkg/build/mfma_probe
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -std=c++17 scripts/denorm_probe.hip -o $pkg/build/denorm_probe 2> /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 scripts/tile4_probe.hip -o $pkg/build/tile4_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 scripts/tailbench.hip -L$pkg -lmi355_retrieval \
  -Wl,-rpath,'$ORIGIN/..' -o $pkg/build/tailbench
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=128 scripts/ingestbench.hip -o $pkg/build/ingestbench 2> /dev/null
echo built $pkg/build/ingestbench $pkg/build/tailbench $pkg/build/kbench $pkg/build/mfma_probe $pkg/build/denorm_probe $pkg/build/tile4_probe
