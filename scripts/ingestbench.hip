// Standalone timing of the gallery ingest kernel (csrc/ingest.hip compiled INTO this program, so that -DMI_INGEST_PROBE=n
// builds diagnostic variants: 1 = no rounding statistics, 2 = no f32 row store, 4 = no image store; 0 = the product kernel).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=0 scripts/ingestbench.hip -o ingestbench
//   ingestbench [rows] [dim]
#ifdef MI_INGEST_SRC          // A/B against a kept copy of the kernels: -DMI_INGEST_SRC='"../ab/ingest_old.hip"' -I <csrc>
#include MI_INGEST_SRC
#else
#include "../image-search-engine-for-historical-research_amd/csrc/ingest.hip"
#endif

#include <cstdio>
#include <cstdlib>
#include <vector>

// the one symbol ingest.hip takes from another translation unit of the library (select.hip).  The program is NOT linked
// against libmi355_retrieval.so: with it loaded, the kernel of the same mangled name registered by the library is the one
// that runs, whatever this file was compiled with.
namespace mi {
int ensure_dynamic_lds(const void* kernel, int bytes) {
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return 0;
}
int current_device_cus() {
  hipDeviceProp_t p;
  int dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipGetDeviceProperties(&p, dev);
  return p.multiProcessorCount;
}
}  // namespace mi
#ifdef MI_INGEST_AB
// same-process A/B on the SAME buffers (launch-to-launch spread between two processes is 10-15 %: where the driver put the
// buffers): a kept copy of ingest.hip compiled as an object of its own with -Dmi=mi_old (scripts/ingest_ab.sh)
namespace mi_old {
struct RowStat;
int ensure_dynamic_lds(const void* kernel, int bytes) { return mi::ensure_dynamic_lds(kernel, bytes); }
int current_device_cus() { return mi::current_device_cus(); }
void launch_ingest(const void* src, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, int norm_mode, float* out_f32,
                   void* out_img, int img_f16, RowStat* rowstat, int32_t dp, int64_t npad, hipStream_t stream, int64_t row_base);
}  // namespace mi_old
#endif

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__global__ void fill(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = (float)((int)(x & 0xFFFF) - 32768) * (1.0f / 32768.0f);
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1005994;
  const int d = argc > 2 ? atoi(argv[2]) : 2048;
  // layout of the source: "rows" = row-major [n][d] (default); "cols" = the reference's [d][n] (row stride 1, column stride n);
  // "cols:<stride>" = [d][stride] with stride >= n (e.g. a multiple of 32: every 64-byte column segment of a panel aligned)
  const char* layout = argc > 3 ? argv[3] : "rows";
  const bool cols = layout[0] == 'c';
  const int64_t cstride = (cols && layout[4] == ':') ? atoll(layout + 5) : n;
  const int64_t npad = (n + 255) / 256 * 256;
  float *src, *out;
  uint16_t* img;
  mi::RowStat* rs;
  CK(hipMalloc(&src, (size_t)(cols ? cstride : n) * d * 4 + 256));
  CK(hipMalloc(&out, (size_t)n * d * 4 + 256));
  CK(hipMalloc(&img, (size_t)npad * d * 2 + 256));
  CK(hipMalloc(&rs, (size_t)npad * sizeof(mi::RowStat)));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, src, (size_t)(cols ? cstride : n) * d);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (getenv("MI_INGEST_PADS")) {
    // placement probe (row-major source): source | f32 rows | image inside ONE allocation, the two destinations shifted by the
    // listed paddings -- does the launch time follow the distances between the three streams?  MI_INGEST_PADS="0,4096,65536,..."
    const size_t sb = (size_t)n * d * 4, ib = (size_t)npad * d * 2, room = (size_t)64 << 20;
    char* arena;
    CK(hipMalloc((void**)&arena, 2 * sb + ib + 2 * room));
    printf("arena %p (offset in its GiB: %zu MiB)\n", (void*)arena, ((size_t)arena & (((size_t)1 << 30) - 1)) >> 20);
    float* asrc = (float*)arena;
    CK(hipMemcpy(asrc, src, sb, hipMemcpyDeviceToDevice));
    std::vector<size_t> pads;
    for (const char* p = getenv("MI_INGEST_PADS"); *p;) {
      pads.push_back((size_t)atoll(p));
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
    for (size_t p1 : pads)
      for (size_t p2 : pads) {
        float* aout = (float*)(arena + sb + p1);
        uint16_t* aimg = (uint16_t*)(arena + sb + room + sb + p2);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(e0, 0));
          mi::launch_ingest(asrc, 0, n, d, d, 1, 1, aout, aimg, 1, rs, d, npad, 0, 0);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) best = std::min(best, ms);
        }
        printf("pads f32 rows +%zu image +%zu: %.3f ms\n", p1, p2, best);
      }
    return 0;
  }
#ifdef MI_INGEST_AB
  const int nrep = 13;
#else
  const int nrep = 6;
#endif
  for (int rep = 0; rep < nrep; ++rep) {
    CK(hipEventRecord(e0, 0));
#ifdef MI_INGEST_AB
    if (rep % 2 == 0) {
      if (cols) mi_old::launch_ingest(src, 0, n, d, 1, cstride, 1, out, img, 1, (mi_old::RowStat*)rs, d, npad, 0, 0);
      else mi_old::launch_ingest(src, 0, n, d, d, 1, 1, out, img, 1, (mi_old::RowStat*)rs, d, npad, 0, 0);
      layout = cols ? "cols(old)" : "rows(old)";
    } else
#endif
    {
    if (cols) mi::launch_ingest(src, 0, n, d, 1, cstride, 1, out, img, 1, rs, d, npad, 0, 0);
    else mi::launch_ingest(src, 0, n, d, d, 1, 1, out, img, 1, rs, d, npad, 0, 0);
#ifdef MI_INGEST_AB
    layout = cols ? "cols(new)" : "rows(new)";
#endif
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gb = ((double)n * d * 4 * ((MI_INGEST_PROBE & 2) ? 1 : 2) + ((MI_INGEST_PROBE & 4) ? 0.0 : (double)npad * d * 2)) / 1e9;
    if (rep) printf("probe %d %s rows %lld dim %d: %.3f ms, %.2f TB/s of %.1f GB moved\n", MI_INGEST_PROBE, layout, (long long)n, d, ms, gb / ms, gb);
  }
  // the runtime's device-to-device copy of the same source into the same f32 destination (8.2 GB read + 8.2 GB written): what a
  // copy achieves on THESE buffers, wherever the driver put them
  for (int rep = 0; rep < 3 && !cols; ++rep) {
    CK(hipEventRecord(e0, 0));
    CK(hipMemcpyAsync(out, src, (size_t)n * d * 4, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("copy  %d d2d rows %lld dim %d: %.3f ms, %.2f TB/s of %.1f GB moved\n", MI_INGEST_PROBE, (long long)n, d, ms, 2.0 * n * d * 4 / 1e9 / ms, 2.0 * n * d * 4 / 1e9);
  }
  mi::RowStat h;
  CK(hipMemcpy(&h, rs + 5, sizeof(h), hipMemcpyDeviceToHost));
  float o5[2];
  CK(hipMemcpy(o5, out + 5 * (size_t)d, 8, hipMemcpyDeviceToHost));
  printf("row 5: norms %g %g %g, out %g %g\n", h.norm_f32, h.norm_img, h.norm_diff, o5[0], o5[1]);
  return 0;
}
