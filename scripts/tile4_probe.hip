// Probe: the "4-wave" structure of the scoring kernel -- ONE wave per SIMD with 128 x 128 outputs (256 accumulator
// VGPRs), 16 ds_read_b128 per 64 MFMAs instead of the product's 12 per 32 (a third fewer LDS bytes per flop) -- on the
// product's data path: the same tile-blocked fp16 images, 256 x 256 workgroup tile, XCD-aware tile order, K-slices of 32
// moved by global_load_lds_dwordx4 into LDS rings that never drain, counted vmcnt, raw s_barrier.  No survivor filter: at
// the end of a tile every accumulator is added into a checksum (the cost of the filter's "decide" step, roughly), so the
// row to compare with is the product kernel WITHOUT its filter (kbench `nofilter:4`).  Diagnostics only; not part of the
// library.  Build: scripts/kbench_build.sh.  Usage: tile4_probe [--rows N] [--reps K] [--bf16]
//
// What differs from the product (csrc/gemm_select.hip, gemm_tile_kernel), by necessity:
//  * no partner wave on the SIMD: fragment reads of slice S+1 and the DMA issue of slice S+3 are interleaved with the
//    MFMAs of slice S by the wave itself (fragments double-buffered in registers: 2 x 64 VGPRs);
//  * one vmcnt stream per wave carries BOTH operands (a wave issues 4 pieces of A and 4 of B per slice), so both rings
//    have the same lead: 4 + 4 slots, three slices in flight;
//  * one s_barrier per slice (in the middle of the MFMA stream) instead of two.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../image-search-engine-for-historical-research_amd/csrc/common.h"

using namespace mi;

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

constexpr int R4 = 4;                                   // ring slots per operand; R4 - 1 slices in flight
constexpr int A4_RING = 0, B4_RING = R4 * SLICE_BYTES;  // 64 KiB + 64 KiB
constexpr int LDS4_BYTES = 2 * R4 * SLICE_BYTES;

__device__ __forceinline__ unsigned long long stamp4() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// VAR: 0 = full, 1 = no DMA (the rings keep whatever they hold), 2 = no DMA and fragments read once (MFMA + barriers only)
template <bool F16, int VAR>
__global__ __launch_bounds__(256, 1) void tile4_kernel(const char* __restrict__ gal, const char* __restrict__ qry,
                                                       uint32_t ntiles, uint32_t nqt, uint32_t KSL,
                                                       double* __restrict__ out_sum, unsigned long long* __restrict__ dbg) {
  using frag_t = typename std::conditional<F16, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t b = blockIdx.x, nwg = gridDim.x >> 3, xcd = b & 7u, j = b >> 3;
  const uint32_t start_x = (uint32_t)(((uint64_t)ntiles * xcd) >> 3), cnt_x = (uint32_t)(((uint64_t)ntiles * (xcd + 1)) >> 3) - start_x;
  const uint32_t nvirt = cnt_x * nqt;
  if (j >= nvirt) return;
  const uint32_t my_tiles = (nvirt - j + nwg - 1) / nwg;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  auto tile_of = [&](uint32_t i, uint32_t& gt, uint32_t& qt) {
    const uint32_t v = j + (i < my_tiles ? i : my_tiles - 1) * nwg;   // past the end: harmless re-load of the last tile
    qt = v % nqt;
    gt = start_x + v / nqt;
  };
  // ---- DMA stream: slice s of my i-th tile, continuous over tile boundaries; 4 pieces of A and 4 of B per wave and slice
  uint32_t pf_i = 0, pf_sl = 0, wr_slot = 0;
  const char *pfa, *pfb;
  auto pf_set = [&](uint32_t i) {
    uint32_t gt, qt;
    tile_of(i, gt, qt);
    pfa = gal + (int64_t)gt * KSL * SLICE_BYTES + w * 4096;
    pfb = qry + (int64_t)qt * KSL * SLICE_BYTES + w * 4096;
  };
  pf_set(0);
  const uint32_t pf_lane = (uint32_t)lane * 16u;
  // the eight pieces of one slice (p = 0..3: A, 4..7: B) are issued one at a time, between MFMAs
  uint32_t pf_off = 0;
  auto issue_piece = [&](int p) {
    if (VAR != 0) return;
    if (p == 0) {
      pf_off = pf_lane;
      asm volatile("" : "+v"(pf_off));
    }
    const GLOBAL_AS void* src = (const GLOBAL_AS void*)((p < 4 ? pfa : pfb) + pf_off);
    LDS_AS void* dst = (LDS_AS void*)(smem + (p < 4 ? A4_RING : B4_RING) + wr_slot * SLICE_BYTES + w * 4096);
    switch (p & 3) {
      case 0: __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0); break;
      case 1: __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0); break;
      case 2: __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0); break;
      default: __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0); break;
    }
  };
  auto issue_advance = [&]() {
    pfa += SLICE_BYTES;
    pfb += SLICE_BYTES;
    if (++pf_sl == KSL) {
      pf_sl = 0;
      pf_set(++pf_i);
    }
    if (++wr_slot == R4) wr_slot = 0;
  };
  auto issue = [&]() {
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_piece(p);
    issue_advance();
  };

  f32x4 acc[8][8];
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[mb][nb] = zero4;
  const uint32_t fsw = (0u - (uint32_t)(l15 >> 2)) & 3u;
  const uint32_t a_off = (uint32_t)A4_RING + (uint32_t)(wr * 128 + l15) * 64u + ((((uint32_t)lq) ^ fsw) << 4);
  uint32_t b_off = (uint32_t)B4_RING + (uint32_t)(wc * 128 + l15) * 64u + ((((uint32_t)lq) ^ fsw) << 4);
  asm volatile("" : "+v"(b_off));
  uint32_t rd_slot = 0;                                  // slot of the slice whose fragments are read NEXT

  auto read_frags = [&](frag_t (&af)[8], frag_t (&bf)[8]) {
    const char* base = smem + rd_slot * SLICE_BYTES;
    if (++rd_slot == R4) rd_slot = 0;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) bf[nb] = *reinterpret_cast<const frag_t*>(base + b_off + nb * 1024);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) af[mb] = *reinterpret_cast<const frag_t*>(base + a_off + mb * 1024);
  };
  auto mfma = [&](frag_t a, frag_t bb, f32x4 c) -> f32x4 {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bb, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, c, 0, 0, 0);
  };
  // One K-slice: 64 MFMAs on (ca, cb); part 1 carries the DMA issue of the slice three ahead, then the mid-stream barrier
  // publishes slice S+1, part 2 carries its 16 fragment reads into (na, nb_).  Snake order over the 8 x 8 blocks.
  auto slice = [&](frag_t (&ca)[8], frag_t (&cb)[8], frag_t (&na)[8], frag_t (&nbf)[8], auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int m2 = 0; m2 < 8; ++m2) {
        const int mb = (nb & 1) ? 7 - m2 : m2;
        acc[mb][nb] = mfma(ca[mb], cb[nb], FIRST ? zero4 : acc[mb][nb]);
        if ((m2 & 3) == 3) {                               // 4 MFMAs, then one DMA piece (fenced: the order is the point)
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(nb * 2 + (m2 >> 2));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    issue_advance();
    __builtin_amdgcn_sched_barrier(0);
    // my pieces of slice S+1 have landed when only the slices S+2, S+3 (and nothing older) are outstanding
    if (VAR == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (VAR != 2) read_frags(na, nbf);
#pragma unroll
    for (int nb = 4; nb < 8; ++nb)
#pragma unroll
      for (int m2 = 0; m2 < 8; ++m2) {
        const int mb = (nb & 1) ? 7 - m2 : m2;
        acc[mb][nb] = mfma(ca[mb], cb[nb], FIRST ? zero4 : acc[mb][nb]);
      }
    if (VAR != 2) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {                       // 2 MFMAs, one fragment read
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next slice's fragments are in registers
    if (VAR == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { asm volatile("" : "+v"(na[e])); asm volatile("" : "+v"(nbf[e])); }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: slices 0, 1, 2 in flight; slice 0 landed for everybody; its fragments in registers
  issue();
  issue();
  issue();
  if (VAR == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  frag_t fa0[8], fb0[8], fa1[8], fb1[8];
  read_frags(fa0, fb0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (VAR == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa1[e] = fa0[e]; fb1[e] = fb0[e]; }
  }
  const unsigned long long clk0 = stamp4(), rt0 = __builtin_amdgcn_s_memrealtime();
  double part = 0.0;
  for (uint32_t i = 0; i < my_tiles; ++i) {
    slice(fa0, fb0, fa1, fb1, std::true_type{});                          // slice 0: accumulate onto the constant 0
#pragma unroll 1
    for (uint32_t sp = 0; sp < (KSL - 2) / 2; ++sp) {                     // KSL even: 1 + 2 * (KSL - 2) / 2 + 1 slices
      slice(fa1, fb1, fa0, fb0, std::false_type{});
      slice(fa0, fb0, fa1, fb1, std::false_type{});
    }
    slice(fa1, fb1, fa0, fb0, std::false_type{});
    // ---- tile done: checksum of the 256 accumulators of this lane (stands in for the filter's decide step)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        s0 += acc[mb][nb][0]; s1 += acc[mb][nb][1]; s2 += acc[mb][nb][2]; s3 += acc[mb][nb][3];
      }
    part += (double)((s0 + s1) + (s2 + s3));
  }
  const unsigned long long clk1 = stamp4(), rt1 = __builtin_amdgcn_s_memrealtime();
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) {
    atomicAdd(out_sum, part);
    unsigned long long* d = dbg + (uint64_t)(b * 4 + w) * 4;
    d[0] = clk1 - clk0;
    d[1] = rt1 - rt0;
    d[2] = (unsigned long long)my_tiles * KSL;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trailing (unused) DMA pieces land before the LDS is released
}

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// the images of scripts/kbench.hip: every element approximately N(0, 1 / D), fp16 or bf16
__global__ void fill_image(uint16_t* img, size_t count, uint64_t seed, float scale, int f16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix(seed ^ (i * 0x2545F4914F6CDD1Dull));
    float u = 0.f;
    for (int jj = 0; jj < 4; ++jj) u += (float)((h >> (16 * jj)) & 0xFFFF) * (1.0f / 65536.0f);
    const float v = (u - 2.0f) * 1.7320508f * scale;
    uint16_t bits;
    if (f16) {
      const _Float16 hv = (_Float16)v;
      bits = *reinterpret_cast<const uint16_t*>(&hv);
    } else {
      const uint32_t ub = __float_as_uint(v);
      bits = (uint16_t)((ub + 0x7FFFu + ((ub >> 16) & 1u)) >> 16);
    }
    img[i] = bits;
  }
}
// column sums of a tile-blocked image [tiles][slices][256 rows][32 k, chunks swizzled]: sum over all rows for every k
// (the swizzle permutes the four 8-element chunks of a row inside its 32-k slice, so it is undone here)
__global__ void image_colsum(const uint16_t* img, uint32_t ntile, uint32_t nsl, int f16, double* out) {
  const uint32_t sl = blockIdx.x;                       // one block per slice; thread = (k in slice)
  const uint32_t kk = threadIdx.x & 31u, part = threadIdx.x >> 5, nparts = blockDim.x >> 5;
  double s = 0.0;
  for (uint32_t t = 0; t < ntile; ++t) {
    const uint16_t* blk = img + ((size_t)t * nsl + sl) * SLICE_ELEMS;
    for (uint32_t r = part; r < TILE; r += nparts) {
      const uint32_t ch = kk >> 3, e = kk & 7u;
      const uint16_t bits = blk[(size_t)r * SLICE_K + (swz_chunk(r, ch) << 3) + e];
      float v;
      if (f16) v = (float)*reinterpret_cast<const _Float16*>(&bits);
      else v = __uint_as_float((uint32_t)bits << 16);
      s += (double)v;
    }
  }
  atomicAdd(&out[sl * 32 + kk], s);
}

int main(int argc, char** argv) {
  int64_t rows = 1005994;
  int d = 2048, nq = 1024, reps = 6, rounds = 3, f16 = 1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : "0"); };
    if (a == "--rows") rows = atoll(next().c_str());
    else if (a == "--reps") reps = atoi(next().c_str());
    else if (a == "--rounds") rounds = atoi(next().c_str());
    else if (a == "--bf16") f16 = 0;
  }
  const int dp = (int)round_up(d, BK);
  const int64_t npad = round_up(rows, TILE), ntiles = npad / TILE;
  const int qpad = (int)round_up(nq, TILE), nqt = qpad / TILE, nsl = dp / SLICE_K;
  if (nsl < 4 || (nsl & 1)) { fprintf(stderr, "needs an even number of K-slices >= 4\n"); return 2; }
  const size_t gal_elems = (size_t)npad * dp, q_elems = (size_t)qpad * dp;
  uint16_t *gal = nullptr, *qry = nullptr;
  CK(hipMalloc((void**)&gal, gal_elems * 2 + 256));
  CK(hipMalloc((void**)&qry, q_elems * 2 + 256));
  const float scale = 1.0f / std::sqrt((float)d);
  hipLaunchKernelGGL(fill_image, dim3(4096), dim3(256), 0, 0, gal, gal_elems, 0x1234ull, scale, f16);
  hipLaunchKernelGGL(fill_image, dim3(512), dim3(256), 0, 0, qry, q_elems, 0x9876ull, scale, f16);
  double *sg = nullptr, *sq = nullptr, *out_sum = nullptr;
  unsigned long long* dbg = nullptr;
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const unsigned grid = (unsigned)(cus / 8) * 8u;
  CK(hipMalloc((void**)&sg, dp * 8));
  CK(hipMalloc((void**)&sq, dp * 8));
  CK(hipMalloc((void**)&out_sum, 8));
  CK(hipMalloc((void**)&dbg, (size_t)grid * 4 * 4 * 8));
  CK(hipMemset(sg, 0, dp * 8));
  CK(hipMemset(sq, 0, dp * 8));
  hipLaunchKernelGGL(image_colsum, dim3(nsl), dim3(256), 0, 0, gal, (uint32_t)ntiles, (uint32_t)nsl, f16, sg);
  hipLaunchKernelGGL(image_colsum, dim3(nsl), dim3(256), 0, 0, qry, (uint32_t)nqt, (uint32_t)nsl, f16, sq);
  CK(hipDeviceSynchronize());
  std::vector<double> hg(dp), hq(dp);
  CK(hipMemcpy(hg.data(), sg, dp * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hq.data(), sq, dp * 8, hipMemcpyDeviceToHost));
  double expect = 0.0, scale_abs = 0.0;
  for (int k = 0; k < dp; ++k) { expect += hg[k] * hq[k]; scale_abs += std::fabs(hg[k] * hq[k]); }
  printf("# tile4_probe rows=%lld d=%d q=%d tiles=%lld slices=%d %s grid=%u  LDS %d B, 256 threads (one wave per SIMD)\n",
         (long long)rows, d, nq, (long long)ntiles, nsl, f16 ? "f16" : "bf16", grid, LDS4_BYTES);
  const double flops = 2.0 * (double)qpad * (double)npad * dp;   // every padded row and query is scored, like kbench counts rows
  const double flops_rows = 2.0 * nq * (double)rows * d;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct V { const char* name; int var; std::vector<float> ms; double sum = 0; double mhz = 0, cps = 0; };
  std::vector<V> vs = {{"tile4", 0}, {"tile4_nodma", 1}, {"tile4_nodma_nofrag", 2}};
  auto launch = [&](int var) {
    const char* g = (const char*)gal;
    const char* q = (const char*)qry;
#define LAUNCH4(F, VV)                                                                                                  \
  do {                                                                                                                  \
    CK(hipFuncSetAttribute((const void*)tile4_kernel<F, VV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4_BYTES)); \
    hipLaunchKernelGGL((tile4_kernel<F, VV>), dim3(grid), dim3(256), LDS4_BYTES, 0, g, q, (uint32_t)ntiles, (uint32_t)nqt,  \
                       (uint32_t)nsl, out_sum, dbg);                                                                    \
  } while (0)
    if (f16) {
      if (var == 0) LAUNCH4(true, 0); else if (var == 1) LAUNCH4(true, 1); else LAUNCH4(true, 2);
    } else {
      if (var == 0) LAUNCH4(false, 0); else if (var == 1) LAUNCH4(false, 1); else LAUNCH4(false, 2);
    }
  };
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) {
      launch(v.var);                                       // warm-up of this variant
      CK(hipDeviceSynchronize());
      for (int k = 0; k < reps; ++k) {
        CK(hipMemset(out_sum, 0, 8));
        CK(hipEventRecord(e0, nullptr));
        launch(v.var);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        v.ms.push_back(ms);
      }
      CK(hipGetLastError());
      CK(hipMemcpy(&v.sum, out_sum, 8, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> c((size_t)grid * 4 * 4);
      CK(hipMemcpy(c.data(), dbg, c.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> mhz, cps;
      for (size_t wv = 0; wv < (size_t)grid * 4; ++wv)
        if (c[wv * 4 + 1] > 0 && c[wv * 4 + 2] > 0) {
          mhz.push_back((double)c[wv * 4] / (double)c[wv * 4 + 1] * 100.0);
          cps.push_back((double)c[wv * 4] / (double)c[wv * 4 + 2]);
        }
      std::sort(mhz.begin(), mhz.end());
      std::sort(cps.begin(), cps.end());
      if (!mhz.empty()) { v.mhz = mhz[mhz.size() / 2]; v.cps = cps[cps.size() / 2]; }
    }
  for (auto& v : vs) {
    std::vector<float> m = v.ms;
    std::sort(m.begin(), m.end());
    const double med = m[m.size() / 2], mn = m.front();
    const double rel = std::fabs(v.sum - expect) / scale_abs;
    printf("%-20s median %.4f ms = %7.1f TF (%.1f TF on the %lld valid rows)   min %.4f ms   in-kernel clock %.0f MHz   loop cycles per "
           "slice %.0f (floor 1024)   checksum %s (sum %.6e, expected %.6e, |diff| / sum|terms| %.1e)\n",
           v.name, med, flops / med / 1e9, flops_rows / med / 1e9, (long long)rows, mn, v.mhz, v.cps,
           v.var == 0 ? (rel < 1e-4 ? "ok" : "MISMATCH") : "n/a", v.sum, expect, rel);
  }
  return 0;
}
