# gallery ingest kernel: consecutive rows per workgroup turn (MI_INGEST_RUN), binaries built by
#   for r in 2 4 8 16 32; do hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=128 -DMI_INGEST_RUN=$r scripts/ingestbench.hip -o <pkg>/build/ingestbench_run$r; done
cd image-search-engine-for-historical-research_amd
for rep in 1 2; do for r in 2 4 8 16 32; do echo -n "RUN=$r: "; ./build/ingestbench_run$r 2>/dev/null | tail -2 | head -1; done; done
