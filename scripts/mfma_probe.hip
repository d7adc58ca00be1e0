// Probe: what the matrix pipe of one SIMD sustains for v_mfma_f32_16x16x32_f16 streams on random operands, by the
// number of waves per SIMD and the synchronisation between them (diagnostics for the tile kernel's structure).
//   mode 0: free-running waves        mode 1: ping-pong (two wave groups alternate 32-MFMA segments between barriers)
//   mode 2: ping-pong, 64-MFMA segments
// WAVES = waves per workgroup (4 = one per SIMD, 8 = two per SIMD); one workgroup per CU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int WAVES, int MODE, int ORDER>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe(const f16x8* __restrict__ src, float* __restrict__ sink,
                                                               unsigned long long* __restrict__ clk, int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = (WAVES == 8) ? (w >> 2) : 0;
  f16x8 af[8], bf[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) af[i] = src[(blockIdx.x * 12 + i) * 64 + lane];
#pragma unroll
  for (int i = 0; i < 4; ++i) bf[i] = src[(blockIdx.x * 12 + 8 + i) * 64 + lane];
  f32x4 acc[8][4];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (MODE != 0 && grp == 1) __builtin_amdgcn_s_barrier();
  const unsigned long long c0 = stamp(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE != 0) {
      // the "load" half of the ping-pong: nothing to do here, the partner group computes
      __builtin_amdgcn_s_barrier();
    }
    constexpr int REP = (MODE == 2) ? 2 : 1;
#pragma unroll
    for (int rep = 0; rep < REP; ++rep) {
      if (ORDER == 0) {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb], bf[nb], acc[mb][nb], 0, 0, 0);
      } else {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int mb = 0; mb < 8; ++mb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb], bf[nb], acc[mb][nb], 0, 0, 0);
      }
    }
    if (MODE != 0) __builtin_amdgcn_s_barrier();
    // keep the operands opaque so that nothing is hoisted or folded
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(bf[i]));
  }
  const unsigned long long c1 = stamp(), r1 = __builtin_amdgcn_s_memrealtime();
  if (MODE != 0 && grp == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) s += acc[mb][nb][0] + acc[mb][nb][1] + acc[mb][nb][2] + acc[mb][nb][3];
  sink[blockIdx.x * WAVES * 64 + tid] = s;
  if (lane == 0) {
    clk[(blockIdx.x * WAVES + w) * 2] = c1 - c0;
    clk[(blockIdx.x * WAVES + w) * 2 + 1] = r1 - r0;
  }
}


// the same ping-pong / free-running stream with v_mfma_f32_32x32x16_f16: wave tile 128 x 64 = 4 x 2 blocks, two k halves per
// 32-deep slice -> 16 MFMAs of 8 passes per slice instead of 32 of 4 passes; half the operand-register reads per flop
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe32(const f16x8* __restrict__ src, float* __restrict__ sink,
                                                                 unsigned long long* __restrict__ clk, int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = (WAVES == 8) ? (w >> 2) : 0;
  f16x8 af[4][2], bf[2][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) af[i >> 1][i & 1] = src[(blockIdx.x * 12 + i) * 64 + lane];
#pragma unroll
  for (int i = 0; i < 4; ++i) bf[i >> 1][i & 1] = src[(blockIdx.x * 12 + 8 + i) * 64 + lane];
  f32x16 acc[4][2];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;
  __syncthreads();
  if (MODE != 0 && grp == 1) __builtin_amdgcn_s_barrier();
  const unsigned long long c0 = stamp(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE != 0) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mb][kh], bf[nb][kh], acc[mb][nb], 0, 0, 0);
    if (MODE != 0) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(af[i >> 1][i & 1]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(bf[i >> 1][i & 1]));
  }
  const unsigned long long c1 = stamp(), r1 = __builtin_amdgcn_s_memrealtime();
  if (MODE != 0 && grp == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[mb][nb][e];
  sink[blockIdx.x * WAVES * 64 + tid] = s;
  if (lane == 0) {
    clk[(blockIdx.x * WAVES + w) * 2] = c1 - c0;
    clk[(blockIdx.x * WAVES + w) * 2 + 1] = r1 - r0;
  }
}

__global__ void fill(uint16_t* p, size_t n, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    float u = 0.f;
    for (int j = 0; j < 4; ++j) u += (float)((x >> (16 * j)) & 0xFFFF) * (1.0f / 65536.0f);
    const _Float16 h = (_Float16)((u - 2.0f) * 1.7320508f * scale);
    p[i] = *reinterpret_cast<const uint16_t*>(&h);
  }
}

template <int WAVES, int MODE, int ORDER, int SHAPE = 0>
static void run(const char* name, const f16x8* src, float* sink, unsigned long long* clk, int iters, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0, nullptr));
    if (SHAPE == 0)
      hipLaunchKernelGGL((probe<WAVES, MODE, ORDER>), dim3(grid), dim3(WAVES * 64), 0, nullptr, src, sink, clk, iters);
    else
      hipLaunchKernelGGL((probe32<WAVES, MODE>), dim3(grid), dim3(WAVES * 64), 0, nullptr, src, sink, clk, iters);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (r) ms.push_back(t);
  }
  CK(hipGetLastError());
  std::vector<unsigned long long> c((size_t)grid * WAVES * 2);
  CK(hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, mhz;
  for (size_t i = 0; i < c.size(); i += 2) {
    cyc.push_back((double)c[i]);
    mhz.push_back((double)c[i] / (double)c[i + 1] * 100.0);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(mhz.begin(), mhz.end());
  std::sort(ms.begin(), ms.end());
  const int per_it = (MODE == 2 ? 64 : 32);
  const double mfma_per_simd = (double)iters * per_it * (WAVES / 4);
  const double flops = (double)grid * WAVES * iters * per_it * 16384.0;
  printf("%-34s waves/SIMD=%d  %.3f ms  %7.1f TF  clock %.0f MHz  cycles per 16 Kflop on the SIMD %.2f\n", name, WAVES / 4,
         ms[ms.size() / 2], flops / ms[ms.size() / 2] / 1e9, mhz[mhz.size() / 2], cyc[cyc.size() / 2] / mfma_per_simd);
}

int main() {
  const int grid = 256;
  f16x8* src;
  float* sink;
  unsigned long long* clk;
  CK(hipMalloc((void**)&src, (size_t)grid * 12 * 64 * 16));
  CK(hipMalloc((void**)&sink, (size_t)grid * 512 * 4));
  CK(hipMalloc((void**)&clk, (size_t)grid * 8 * 16));
  hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, nullptr, (uint16_t*)src, (size_t)grid * 12 * 64 * 8, 0.0221f);
  CK(hipDeviceSynchronize());
  const int iters = 60000;
  run<4, 0, 0>("1 wave/SIMD free (A reused x4)", src, sink, clk, iters * 2, grid);
  run<4, 0, 1>("1 wave/SIMD free (B reused x8)", src, sink, clk, iters * 2, grid);
  run<8, 0, 0>("2 waves/SIMD free (A reused x4)", src, sink, clk, iters, grid);
  run<8, 0, 1>("2 waves/SIMD free (B reused x8)", src, sink, clk, iters, grid);
  run<8, 1, 0>("2 waves/SIMD ping-pong 32", src, sink, clk, iters, grid);
  run<8, 2, 0>("2 waves/SIMD ping-pong 64", src, sink, clk, iters / 2, grid);
  run<8, 2, 1>("2 waves/SIMD ping-pong 64 (B x8)", src, sink, clk, iters / 2, grid);
  run<8, 0, 0, 1>("32x32x16: 2 waves/SIMD free", src, sink, clk, iters, grid);
  run<8, 1, 0, 1>("32x32x16: 2 waves/SIMD ping-pong 32", src, sink, clk, iters, grid);
  run<8, 1, 0>("2 waves/SIMD ping-pong 32 (again)", src, sink, clk, iters, grid);
  run<8, 1, 0, 1>("32x32x16: ping-pong 32 (again)", src, sink, clk, iters, grid);
  return 0;
}
