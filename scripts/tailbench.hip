// Standalone timing of the per-batch "fixed cost" kernels of libmi355_retrieval.so (no Python, no torch, no profiler):
// each kernel is launched back to back on synthetic state shaped like a 1024-query batch on the 1 M-row gallery (8192
// sample scores, ~900 survivors, ~127 candidates per query) and timed with one event pair around the whole train, next to
// an empty kernel of the same grid (the launch + dispatch floor).  Diagnostics only.
//   tailbench [--reps N]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../image-search-engine-for-historical-research_amd/csrc/kernels.h"

using namespace mi;

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float gauss(uint64_t h) {          // ~N(0, 1)
  float u = 0.f;
  for (int j = 0; j < 4; ++j) u += (float)((h >> (16 * j)) & 0xFFFF) * (1.0f / 65536.0f);
  return (u - 2.0f) * 1.7320508f;
}
__global__ void fill_scores_f32(float* p, size_t n, uint64_t seed, float sigma) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = gauss(mix(seed ^ i)) * sigma;
}
// survivors: per query `cnt` entries (score in [lo, lo + span), distinct rows)
__global__ void fill_surv(uint64_t* surv, uint32_t cap, uint32_t cnt, uint32_t nrows, float lo, float span, uint64_t seed) {
  const uint32_t q = blockIdx.x;
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
    const uint64_t h = mix(seed ^ ((uint64_t)q << 32) ^ i);
    const float u = (float)(h & 0xFFFFFF) / 16777216.0f;
    const float sc = lo + span * u * u * u;                  // dense near the threshold, sparse above
    surv[(size_t)q * cap + i] = ((uint64_t)__float_as_uint(sc) << 32) | (uint32_t)((h >> 24) % nrows);
  }
}
__global__ void fill_u32(uint32_t* p, size_t n, uint32_t v, uint32_t stride) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i * stride] = v;
}
__global__ void fill_f32(float* p, size_t n, float v) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void fill_cand(uint32_t* rows, double* sc, uint32_t rcap, uint32_t cnt, uint32_t nrows, uint64_t seed) {
  const uint32_t q = blockIdx.x;
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
    const uint64_t h = mix(seed ^ ((uint64_t)q << 32) ^ i);
    rows[(size_t)q * rcap + i] = (uint32_t)(h % nrows);
    sc[(size_t)q * rcap + i] = 0.08 + 1e-9 * (double)(h >> 40);
  }
}
__global__ void empty_kernel(int* p) {
  if (p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFFFFu) *p = 1;
}

// emit dissected (diagnostics): PHASE 0 = loads of the count and the first candidate + the padding stores only, 1 = + staging in
// LDS and the two barriers, 2 = + the rank-counting loop, 3 = + the result stores (= the product kernel at <= 256 candidates)
template <int PHASE>
__global__ __launch_bounds__(256) void emit_probe(const uint32_t* __restrict__ cand_rows, const uint32_t* __restrict__ cand_cnt,
                                                  const double* __restrict__ cand_score, uint32_t rcap, int32_t k,
                                                  int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
  __shared__ __attribute__((aligned(16))) double ts[512];
  __shared__ __attribute__((aligned(16))) uint32_t ti[512];
  const uint32_t q = blockIdx.x;
  const double* qs = cand_score + (uint64_t)q * rcap;
  const uint32_t* qi = cand_rows + (uint64_t)q * rcap;
  const uint32_t nc_raw = cand_cnt[q];
  const double a = qs[threadIdx.x];
  const uint32_t ia = qi[threadIdx.x];
  const uint32_t nc = min(nc_raw, 256u);
  const bool mine = threadIdx.x < nc;
  uint32_t place = threadIdx.x;
  if (PHASE >= 1) {
    const uint32_t tn4 = (nc + 3u) & ~3u;
    if (threadIdx.x < tn4) {
      ts[threadIdx.x] = mine ? a : __longlong_as_double(0x7FF8000000000000ll);
      ti[threadIdx.x] = mine ? ia : 0xFFFFFFFFu;
    }
    __syncthreads();
    if (PHASE >= 2 && mine) {
      place = 0;
#pragma unroll 2
      for (uint32_t e = 0; e < tn4; e += 4) {
        const double2 b01 = *reinterpret_cast<const double2*>(ts + e), b23 = *reinterpret_cast<const double2*>(ts + e + 2);
        const uint4 id4 = *reinterpret_cast<const uint4*>(ti + e);
        const double bs[4] = {b01.x, b01.y, b23.x, b23.y};
        const uint32_t is[4] = {id4.x, id4.y, id4.z, id4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) place += ((bs[u] > a) || (bs[u] == a && is[u] < ia)) ? 1u : 0u;
      }
    }
  }
  if (PHASE >= 3 || PHASE == 0) {
    if (mine && place < (uint32_t)k) {
      out_idx[(uint64_t)q * k + place] = (int64_t)ia;
      out_score[(uint64_t)q * k + place] = (float)a;
    }
  } else if (mine && place == 0xFFFFFFFFu) out_idx[0] = 0;
}

template <typename F>
static double time_train(const char* name, int reps, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("%-46s %8.2f us per launch (train of %d)\n", name, us, reps);
  return us;
}

int main(int argc, char** argv) {
  int reps = 50;
  for (int i = 1; i < argc; ++i)
    if (std::string(argv[i]) == "--reps" && i + 1 < argc) reps = atoi(argv[++i]);
  const int nq = 1024, k = 100, d = 2048, dp = 2048;
  const uint32_t cap = 12288, rcap = 2048, nrows = 1005994;
  QueryState st{};
  CK(hipMalloc((void**)&st.thr, nq * 4));
  CK(hipMalloc((void**)&st.margin, nq * 4));
  CK(hipMalloc((void**)&st.thr2, nq * 4));
  CK(hipMalloc((void**)&st.qflag, nq * 4));
  CK(hipMalloc((void**)&st.lad_tc, nq * 4));
  CK(hipMalloc((void**)&st.lad_pack, nq * 4));
  CK(hipMalloc((void**)&st.lad_cnt, nq * 4));
  CK(hipMalloc((void**)&st.cnt, (size_t)nq * CNT_STRIDE * 4));
  CK(hipMalloc((void**)&st.surv, (size_t)nq * cap * 8));
  CK(hipMalloc((void**)&st.flags, 16));
  CK(hipMemset(st.flags, 0, 16));
  st.repair = st.flags + 1;
  CK(hipMemset(st.qflag, 0, nq * 4));
  CK(hipMemset(st.lad_cnt, 0, nq * 4));
  st.cap = cap;
  float *topvals, *L;
  uint64_t* stats2;
  uint32_t *cand_rows, *cand_cnt;
  double* cand_score;
  int64_t* out_idx;
  float* out_score;
  CK(hipMalloc((void**)&topvals, (size_t)nq * k * 4));
  CK(hipMalloc((void**)&L, nq * 4));
  CK(hipMalloc((void**)&stats2, 2 * 1024 * 8));
  CK(hipMemset(stats2, 0, 2 * 1024 * 8));
  CK(hipMalloc((void**)&cand_rows, (size_t)nq * rcap * 4));
  CK(hipMalloc((void**)&cand_cnt, nq * 4));
  CK(hipMalloc((void**)&cand_score, (size_t)nq * rcap * 8));
  CK(hipMalloc((void**)&out_idx, (size_t)nq * k * 8));
  CK(hipMalloc((void**)&out_score, (size_t)nq * k * 4));
  float *gal = nullptr, *qry = nullptr;
  CK(hipMalloc((void**)&gal, (size_t)nrows * dp * 4));
  CK(hipMalloc((void**)&qry, (size_t)nq * dp * 4));
  hipLaunchKernelGGL(fill_scores_f32, dim3(4096), dim3(256), 0, 0, gal, (size_t)nrows * dp, 1ull, 0.0221f);
  hipLaunchKernelGGL(fill_scores_f32, dim3(256), dim3(256), 0, 0, qry, (size_t)nq * dp, 2ull, 0.0221f);
  hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, 0, st.margin, (size_t)nq, 1.4e-3f);
  CK(hipDeviceSynchronize());
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("# tailbench: %d queries, K = %d, %d CUs\n", nq, k, cus);

  // launch floors
  time_train("empty kernel 1024 x 512 threads", reps, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(512), 0, 0, (int*)nullptr); });
  time_train("empty kernel 1024 x 256 threads", reps, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, 0, (int*)nullptr); });
  time_train("empty kernel 65536 x 64 threads", reps, [&] { hipLaunchKernelGGL(empty_kernel, dim3(64, 1024), dim3(64), 0, 0, (int*)nullptr); });

  // ---- sample_threshold: 8192 f32 sample scores per query (sigma = 1 / sqrt(D)), r = 10, ladder rank 3
  auto prep_sample = [&] {
    hipLaunchKernelGGL(fill_u32, dim3(4), dim3(256), 0, 0, st.cnt, (size_t)nq, 8192u, (uint32_t)CNT_STRIDE);
    hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, 0, st.thr, (size_t)nq, -INFINITY);
  };
  for (int qv = 0; qv < nq; ++qv) {}
  {
    // scores of query q at ((float*)(surv + q * cap))[i]
    hipLaunchKernelGGL(fill_scores_f32, dim3(2048), dim3(256), 0, 0, (float*)st.surv, (size_t)nq * cap * 2, 3ull, 0.0221f);
    prep_sample();
    CK(hipDeviceSynchronize());
    time_train("sample_threshold<16> (f32 scores)", reps, [&] {
      prep_sample();                                           // the kernel zeroes cnt: re-arm (two tiny launches, timed along)
      launch_sample_threshold(st, nq, k, 10, 8192u, nullptr, 3, 1);
    });
    time_train("  (the two re-arm launches alone)", reps, [&] { prep_sample(); });
    for (int ph = 1; ph <= 2; ++ph) {
      set_tail_debug_phase(ph);
      time_train(ph == 1 ? "  sample_threshold up to: loads + maxima" : "  sample_threshold up to: W-th of the 512 maxima", reps, [&] {
        prep_sample();
        launch_sample_threshold(st, nq, k, 10, 8192u, nullptr, 3, 1);
      });
    }
    set_tail_debug_phase(0);
  }
  // ---- maintain MODE 1 on ~900 survivors per query (speculative verification on, candidate lists fused)
  auto prep_maint = [&] {
    hipLaunchKernelGGL(fill_surv, dim3(nq), dim3(256), 0, 0, st.surv, cap, 900u, nrows, 0.0663f, 0.2f, 4ull);
    hipLaunchKernelGGL(fill_u32, dim3(4), dim3(256), 0, 0, st.cnt, (size_t)nq, 900u, (uint32_t)CNT_STRIDE);
    hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, 0, st.thr, (size_t)nq, 0.0663f);
    hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, 0, st.thr2, (size_t)nq, 0.06f);
  };
  prep_maint();
  CK(hipDeviceSynchronize());
  time_train("select_maintain<1> (900 survivors, fused cands)", reps, [&] {
    prep_maint();                                              // the kernel compacts in place: re-fill (timed along)
    launch_select_maintain(st, nq, k, 1, topvals, L, stats2, 0, 1, 0, nullptr, nullptr, cand_rows, cand_cnt, rcap);
  });
  time_train("  (the four re-fill launches alone)", reps, [&] { prep_maint(); });
  for (int ph = 1; ph <= 3; ++ph) {
    set_tail_debug_phase(ph);
    time_train(ph == 1 ? "  maintain up to: entries + keys loaded" : ph == 2 ? "  maintain up to: K-th largest" : "  maintain up to: compaction + topvals", reps, [&] {
      prep_maint();
      launch_select_maintain(st, nq, k, 1, topvals, L, stats2, 0, 1, 0, nullptr, nullptr, cand_rows, cand_cnt, rcap);
    });
  }
  set_tail_debug_phase(0);
  // ---- exact re-score of 127 candidates per query (1 M-row f32 gallery) and emit
  hipLaunchKernelGGL(fill_cand, dim3(nq), dim3(256), 0, 0, cand_rows, cand_score, rcap, 127u, nrows, 5ull);
  hipLaunchKernelGGL(fill_u32, dim3(4), dim3(256), 0, 0, cand_cnt, (size_t)nq, 127u, 1u);
  CK(hipDeviceSynchronize());
  time_train("rescore (127 rows x 8 KiB per query)", reps, [&] {
    launch_rescore(gal, qry, dp, nq, cand_rows, cand_cnt, rcap, cand_score, nullptr, 0, nrows - 1);
  });
  time_train("emit (127 candidates -> top 100)", reps, [&] {
    launch_emit(cand_rows, cand_cnt, cand_score, rcap, nq, k, 0, out_idx, out_score, nullptr, nullptr);
  });
  time_train("  emit_probe<0> loads + stores", reps, [&] { hipLaunchKernelGGL(emit_probe<0>, dim3(nq), dim3(256), 0, 0, cand_rows, cand_cnt, cand_score, rcap, k, out_idx, out_score); });
  time_train("  emit_probe<1> + LDS staging, barrier", reps, [&] { hipLaunchKernelGGL(emit_probe<1>, dim3(nq), dim3(256), 0, 0, cand_rows, cand_cnt, cand_score, rcap, k, out_idx, out_score); });
  time_train("  emit_probe<2> + rank loop", reps, [&] { hipLaunchKernelGGL(emit_probe<2>, dim3(nq), dim3(256), 0, 0, cand_rows, cand_cnt, cand_score, rcap, k, out_idx, out_score); });
  time_train("  emit_probe<3> + result stores", reps, [&] { hipLaunchKernelGGL(emit_probe<3>, dim3(nq), dim3(256), 0, 0, cand_rows, cand_cnt, cand_score, rcap, k, out_idx, out_score); });
  time_train("rescore + emit", reps, [&] {
    launch_rescore(gal, qry, dp, nq, cand_rows, cand_cnt, rcap, cand_score, nullptr, 0, nrows - 1);
    launch_emit(cand_rows, cand_cnt, cand_score, rcap, nq, k, 0, out_idx, out_score, nullptr, nullptr);
  });
  (void)d;
  return 0;
}
