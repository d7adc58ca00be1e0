// Gallery / query ingest for gfx950: strided f32|f64 input -> (a) normalised f32 rows [n][dp],
// (b) tile-blocked, chunk-swizzled bf16 image [npad/256][dp/32][256][32], (c) per-row rounding stats.
//
// Restates the normalisation of matching_L2 (src/utils/nnsearch.py:693-698, no eps), of l2n
// (src/layers/functional.py:129-130, eps 1e-6) and the tail of whitenapply (src/utils/whiten.py:10);
// the reference redoes it on every call, here it is done once per gallery (HBM-bound, one-off).
#include <hip/hip_fp16.h>

#include <algorithm>
#include <type_traits>

#include <cstdio>
#include <cstdlib>

#include <atomic>

#include "common.h"
#include "kernels.h"

#ifndef MI_INGEST_PROBE
#define MI_INGEST_PROBE 0      // scripts/ingestbench.hip: 1 = no rounding statistics, 2 = no f32 row store, 4 = no image store
#endif
// consecutive rows one workgroup of the persistent gallery ingest takes per turn (scripts/ingestbench.hip sweeps it)
#ifndef MI_INGEST_RUN
#define MI_INGEST_RUN 4
#endif

namespace mi {

// 16-bit image element: fp16 (11-bit significand: 8x smaller rounding error, same MFMA rate) or bf16 (f32 range)
__device__ __forceinline__ uint16_t cvt_img(float v, int f16, double& back) {
  if (f16) {
    const _Float16 h = (_Float16)v;
    back = (double)(float)h;
    return __builtin_bit_cast(uint16_t, h);
  }
  const __hip_bfloat16 b = __float2bfloat16(v);
  back = (double)__bfloat162float(b);
  return __builtin_bit_cast(uint16_t, b);
}

// Row-per-workgroup variant for small row counts (query batches): 256 threads stride over the columns of ONE
// row, so a 1024-query batch runs on 1024 workgroups instead of 16.  Same arithmetic as ingest_kernel.
template <typename InT>
__global__ __launch_bounds__(256) void ingest_rowwise_kernel(const InT* __restrict__ src, int64_t n, int32_t d,
                                                             int64_t rs, int64_t cs, int norm_mode,
                                                             float* __restrict__ out_f32,
                                                             uint16_t* __restrict__ out_img, int img_f16,
                                                             RowStat* __restrict__ rowstat, int32_t dp, int64_t npad,
                                                             int64_t row_base) {
  __shared__ double red[3][4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t row = blockIdx.x;
  const bool valid = row < n;
  auto block_sum3 = [&](double& a, double& b, double& c) {
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
    __syncthreads();
    if (lane == 0) { red[0][wv] = a; red[1][wv] = b; red[2][wv] = c; }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  };
  double scale = 1.0;
  if (norm_mode != 0) {
    double ss = 0.0, z0 = 0.0, z1 = 0.0;
    if (valid)
      for (int c = t; c < d; c += 256) {
        const double v = (double)src[row * rs + (int64_t)c * cs];
        ss += v * v;
      }
    block_sum3(ss, z0, z1);
    const double nrm = sqrt(ss);
    scale = (norm_mode == 1) ? 1.0 / nrm : 1.0 / (nrm + 1e-6);
  }
  double s_g = 0.0, s_b = 0.0, s_d = 0.0;
  const int nslices = dp / SLICE_K;
  const int64_t orow = row_base + row;          // output row (gallery append: source row 0 <-> row_base)
  const int64_t tileidx = orow / TILE;
  const uint32_t r = (uint32_t)(orow % TILE);
  for (int c0 = t * 8; c0 < dp; c0 += 2048) {
    float vf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      vf[e] = (valid && c < d) ? (float)((double)src[row * rs + (int64_t)c * cs] * scale) : 0.0f;
    }
    const uint32_t sl = (uint32_t)c0 / SLICE_K, ch = ((uint32_t)c0 % SLICE_K) >> 3;
    uint16_t* blk = out_img + (tileidx * nslices + sl) * (int64_t)SLICE_ELEMS + (int64_t)r * SLICE_K;
    union { uint16_t h[8]; uint4 u; } pk;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      double vb;
      pk.h[e] = cvt_img(vf[e], img_f16, vb);
      s_b += vb * vb;
      s_d += (vb - (double)vf[e]) * (vb - (double)vf[e]);
      s_g += (double)vf[e] * (double)vf[e];
    }
    *reinterpret_cast<uint4*>(blk + (swz_chunk(r, ch) << 3)) = pk.u;
    if (valid) {
      float4* o = reinterpret_cast<float4*>(out_f32 + orow * dp + c0);
      o[0] = make_float4(vf[0], vf[1], vf[2], vf[3]);
      o[1] = make_float4(vf[4], vf[5], vf[6], vf[7]);
    }
  }
  block_sum3(s_g, s_b, s_d);
  if (t == 0) {
    RowStat rsd;
    rsd.norm_f32 = (float)(sqrt(s_g) * (1.0 + 1e-6));
    rsd.norm_img = (float)(sqrt(s_b) * (1.0 + 1e-6));
    rsd.norm_diff = (float)(sqrt(s_d) * (1.0 + 1e-6));
    rowstat[orow] = rsd;
  }
}

// Query-batch ingest of the search path: the row-wise kernel above with (i) ONE pass over the source -- thread t keeps its
// columns t, t + 256, ... in registers between the norm and the write phase (same summation order, so the same bits as
// ingest_rowwise_kernel) and the 8-column chunks of the blocked image are regrouped through LDS -- and (ii) the per-query
// state of the search (init_query_state_kernel of select.hip: error margin from the row's rounding norms, thresholds,
// counters, ladder words) written by the row's workgroup, which saves a dependent launch per batch.  dp <= 4096.
struct QueryInit {
  const float* gstat3;     // gallery maxima of (||g||, ||g_hat||, ||g_hat - g||)
  float gamma;
  int use_img_terms;
  uint32_t first_cnt;
  int32_t nq;
  QueryState st;
  int64_t gallery_rows;    // INIT == false: rows the persistent grid produces (the shard's rows padded to whole tiles)
  uint32_t zero_scores;    // INIT: the first zero_scores 4-byte sample scores of every query are set to 0 (a K-split
                           // bootstrap launch adds its partial scores onto them, ScoreArgs::ksplit); 0 = leave them alone
};
constexpr int QI_MAX_PER_THREAD = 16;
// x of lane (l ^ O): O in {1, 2, 4, 8} as data-parallel-primitive moves inside the 16-lane rows (no LDS crossbar, no
// address register, a few cycles instead of a ds_bpermute round trip per level): quad permutations for 1 and 2,
// half-mirror + quad reversal for 4 (7 - i, then 3 - i inside the quad = i ^ 4), mirror + half-mirror for 8 (15 - i, then
// 7 - i inside the half = i ^ 8).  The same partner as __shfl_xor, so the same sums.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
  // (bound_ctrl set: every lane of these permutations has a source, and the compiler then needs no initialised destination)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int O>
__device__ __forceinline__ double xor_lane_f64(double x) {
  if constexpr (O == 1) return dpp_mov_f64<0xB1>(x);                        // quad_perm [1, 0, 3, 2]
  else if constexpr (O == 2) return dpp_mov_f64<0x4E>(x);                   // quad_perm [2, 3, 0, 1]
  else if constexpr (O == 4) return dpp_mov_f64<0x1B>(dpp_mov_f64<0x141>(x));   // row_half_mirror, quad_perm [3, 2, 1, 0]
  else if constexpr (O == 8) return dpp_mov_f64<0x141>(dpp_mov_f64<0x140>(x));  // row_mirror, row_half_mirror
  else if constexpr ((MI_INGEST_PROBE & 256) != 0) return __shfl_xor(x, O);      // A/B: the LDS crossbar for 16 / 32
  else {
    // 16 / 32: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows (the upper half) of one register with the
    // even rows (the lower half) of another; fed the same value twice they return [x0 x0 x2 x2] / [x1 x1 x3 x3] (rows) resp.
    // [lo lo] / [hi hi] (halves), of which every lane picks its partner's (checked against __shfl_xor on the device)
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    unsigned plo, phi;
    if constexpr (O == 16) {
      const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      const bool odd_row = (threadIdx.x >> 4) & 1;
      plo = odd_row ? a[0] : a[1];
      phi = odd_row ? b[0] : b[1];
    } else {
      static_assert(O == 32, "partner distance");
      const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      const bool upper = threadIdx.x & 32;
      plo = upper ? a[0] : a[1];
      phi = upper ? b[0] : b[1];
    }
    return __hiloint2double((int)phi, (int)plo);
  }
}
// Twelve wave butterflies (partners 32, 16, 8, 4, 2, 1) for the price of two.  After the level with partner P the lanes l and
// l ^ P hold the same sum, so bit log2(P) of the lane number is free to tell two VALUES apart: v_permlane32_swap / v_permlane16_swap
// move half of one register into the other half of a second one in a single instruction -- (a, b) -> one register with a's sums in
// one half and b's in the other, one float64 add for both --, and the 8- and 4-lane levels pack by a select.  The adds are the
// butterfly's adds (x[l] + x[l ^ P], the same operands in every lane that keeps the value), so the sums are the same bits.
__device__ __forceinline__ double swap32_add(double a, double b) {   // lanes < 32: a[l] + a[l + 32]; lanes >= 32: b[l - 32] + b[l]
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap16_add(double p, double q) {   // 16-lane rows: [p0 + p1, q0 + q1, p2 + p3, q2 + q3]
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(p), (unsigned)__double2loint(q), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(p), (unsigned)__double2hiint(q), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int O>
__device__ __forceinline__ double pack_add(double x, double y, bool upper) {   // lanes with bit O clear: x[l] + x[l ^ O]; set: y[l] + y[l ^ O]
  const double u = upper ? y : x, v = upper ? x : y;
  return u + xor_lane_f64<O>(v);
}
// g[k], b[k], d[k] (k = 0..3: the four virtual waves): lane 0 receives ((G0 + G1) + G2) + G3 of the butterflied g's, lane 8 the
// same of the b's, lane 4 of the d's (other lanes: don't care).  Where a value lives: after the 32- and 16-lane levels row
// (lane >> 4) = 0, 1, 2, 3 holds k = 0, 2, 1, 3; bit 3 tells g from b, bit 2 those two from d.
__device__ __forceinline__ double reduce_stats12(const double g[4], const double b[4], const double d[4], int lane) {
  const double rg = swap16_add(swap32_add(g[0], g[1]), swap32_add(g[2], g[3]));
  const double rb = swap16_add(swap32_add(b[0], b[1]), swap32_add(b[2], b[3]));
  const double rd = swap16_add(swap32_add(d[0], d[1]), swap32_add(d[2], d[3]));
  const double z1 = pack_add<8>(rg, rb, (lane & 8) != 0);
  const double z2 = rd + xor_lane_f64<8>(rd);
  double w = pack_add<4>(z1, z2, (lane & 4) != 0);
  w += xor_lane_f64<2>(w);
  w += xor_lane_f64<1>(w);
  // rows 0, 2, 1, 3 <-> lanes l, l ^ 32, l ^ 16, l ^ 48 of a lane in row 0
  const double k1 = xor_lane_f64<32>(w), k2 = xor_lane_f64<16>(w);
  const double k3 = xor_lane_f64<32>(k2);
  return ((w + k1) + k2) + k3;
}

// the butterfly of the block reductions below: partners 32, 16, 8, 4, 2, 1 in this order
__device__ __forceinline__ void wave_butterfly(double& a) {
  a += xor_lane_f64<32>(a); a += xor_lane_f64<16>(a); a += xor_lane_f64<8>(a);
  a += xor_lane_f64<4>(a); a += xor_lane_f64<2>(a); a += xor_lane_f64<1>(a);
}

template <typename InT, bool INIT, int PT>
__global__ __launch_bounds__(256) void ingest_query_kernel(const InT* __restrict__ src, int64_t n, int32_t d, int64_t rs,
                                                           int64_t cs, int norm_mode, float* __restrict__ out_f32,
                                                           uint16_t* __restrict__ out_img, int img_f16,
                                                           RowStat* __restrict__ rowstat, int32_t dp, QueryInit qi,
                                                           int64_t row_base) {
  __shared__ double red[3][4];
  __shared__ __attribute__((aligned(16))) float rowbuf[256 * PT];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  auto block_sum3 = [&](double& a, double& b, double& c) {
    wave_butterfly(a); wave_butterfly(b); wave_butterfly(c);
    __syncthreads();
    if (lane == 0) { red[0][wv] = a; red[1][wv] = b; red[2][wv] = c; }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  };
  // Galleries (INIT == false) run a persistent grid: a workgroup takes rows blockIdx.x, + gridDim.x, ... and requests the
  // NEXT row's elements before it reduces the current one.  A million one-row workgroups were paced by the dispatcher and by
  // one memory latency per row and workgroup (3.1 TB/s of read + write traffic); the arithmetic and its order are untouched.
  // nrows = rows to produce (padding rows of the last tile included: they are written as zeros); query batches (INIT) keep
  // one workgroup per row, whose thread 0 also initialises the query's search state.
  const int64_t nrows = INIT ? (int64_t)gridDim.x : qi.gallery_rows;
  InT nxt[PT];     // PT = columns per thread: 8 for dp <= 2048 (half the registers), 16 up to 4096
  // the loads are unconditional (row and column clamped into the source, the value replaced by 0 afterwards): a load behind
  // a branch makes the number of outstanding loads path-dependent, the compiler then waits with vmcnt(0) right behind the
  // requests and the prefetch is gone
  auto request = [&](int64_t r) {
    const int64_t rr = r < n ? r : n - 1;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const int c = t + 256 * j;
#if MI_INGEST_PROBE & 64
      nxt[j] = (InT)(rr + c);
#else
      nxt[j] = src[rr * rs + (int64_t)(c < d ? c : d - 1) * cs];     // replaced by 0 where it is USED (no wait here)
#endif
    }
  };
  // rows of a workgroup come in runs of RUN consecutive rows (run i of workgroup b starts at (i * gridDim.x + b) * RUN): one
  // row's piece of a slice block of the image is 64 bytes, and consecutive rows of a tile are neighbours in it, so a run
  // written by ONE workgroup fills whole 256-byte stretches in ONE XCD's L2 instead of leaving half lines in eight of them
  constexpr int64_t RUN = INIT ? 1 : MI_INGEST_RUN;
  auto row_of = [&](int64_t it) { return ((it / RUN) * (int64_t)gridDim.x + blockIdx.x) * RUN + it % RUN; };
  request(row_of(0));
  for (int64_t it = 0;; ++it) {
  const int64_t row = row_of(it);
  if (row >= nrows) {
    if (it % RUN == 0) break;          // the first row of a run is past the end: so is everything after it
    continue;
  }
  const bool valid = row < n;
  if (INIT && qi.zero_scores && row < qi.nq) {
    float* z = reinterpret_cast<float*>(qi.st.surv + (uint64_t)row * qi.st.cap);
    for (uint32_t i = t; i < qi.zero_scores; i += 256) z[i] = 0.f;
  }
  double v[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) v[j] = (valid && t + 256 * j < d) ? (double)nxt[j] : 0.0;
  if (!INIT) request(row_of(it + 1));   // past the end: a harmless re-load of the last row
  double scale = 1.0;
  if (norm_mode != 0) {
    double ss = 0.0;
#pragma unroll
    for (int j = 0; j < PT; ++j)
      if (t + 256 * j < d) ss += v[j] * v[j];
    // the one-value form of block_sum3 (same adds in the same order; two thirds of its shuffles carried zeros here)
#if !(MI_INGEST_PROBE & 32)
    wave_butterfly(ss);
    __syncthreads();
    if (lane == 0) red[0][wv] = ss;
    __syncthreads();
    ss = red[0][0] + red[0][1] + red[0][2] + red[0][3];
#endif
#if MI_INGEST_PROBE & 8
    scale = ss * 1e-3;
    if (false)
#endif
    {
    const double nrm = sqrt(ss);
    scale = (norm_mode == 1) ? 1.0 / nrm : 1.0 / (nrm + 1e-6);
    }
  }
#pragma unroll
  for (int j = 0; j < PT; ++j)
    if (t + 256 * j < dp) rowbuf[t + 256 * j] = (valid && t + 256 * j < d) ? (float)(v[j] * scale) : 0.0f;
  __syncthreads();
  double s_g = 0.0, s_b = 0.0, s_d = 0.0;
  const int nslices = dp / SLICE_K;
  const int64_t orow = row_base + row;          // output row (gallery append: source row 0 <-> row_base)
  const int64_t tileidx = orow / TILE;
  const uint32_t r = (uint32_t)(orow % TILE);
  for (int c0 = t * 8; c0 < dp; c0 += 2048) {
    float vf[8];
    const float4 lo = *reinterpret_cast<const float4*>(rowbuf + c0), hi = *reinterpret_cast<const float4*>(rowbuf + c0 + 4);
    vf[0] = lo.x; vf[1] = lo.y; vf[2] = lo.z; vf[3] = lo.w; vf[4] = hi.x; vf[5] = hi.y; vf[6] = hi.z; vf[7] = hi.w;
    const uint32_t sl = (uint32_t)c0 / SLICE_K, ch = ((uint32_t)c0 % SLICE_K) >> 3;
    uint16_t* blk = out_img + (tileidx * nslices + sl) * (int64_t)SLICE_ELEMS + (int64_t)r * SLICE_K;
    union { uint16_t h[8]; uint4 u; } pk;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      double vb;
      pk.h[e] = cvt_img(vf[e], img_f16, vb);
#if !(MI_INGEST_PROBE & 1)
      s_b += vb * vb;
      s_d += (vb - (double)vf[e]) * (vb - (double)vf[e]);
      s_g += (double)vf[e] * (double)vf[e];
#endif
    }
#if !(MI_INGEST_PROBE & 4)
    *reinterpret_cast<uint4*>(blk + (swz_chunk(r, ch) << 3)) = pk.u;
#else
    if (pk.u.x == 0x12345678u) *reinterpret_cast<uint4*>(blk) = pk.u;
#endif
    if (valid && !(MI_INGEST_PROBE & 2)) {
      float4* o = reinterpret_cast<float4*>(out_f32 + orow * dp + c0);
      o[0] = lo;
      o[1] = hi;
    }
  }
#if !(MI_INGEST_PROBE & 16)
  block_sum3(s_g, s_b, s_d);
#endif
  if (t == 0) {
    RowStat rsd;
    rsd.norm_f32 = (float)(sqrt(s_g) * (1.0 + 1e-6));
    rsd.norm_img = (float)(sqrt(s_b) * (1.0 + 1e-6));
    rsd.norm_diff = (float)(sqrt(s_d) * (1.0 + 1e-6));
    rowstat[orow] = rsd;
    if (!INIT) continue;
    // ---- per-query search state (select.hip init_query_state_kernel, same arithmetic)
    const QueryState& st = qi.st;
    const int q = (int)row;
    if (q == 0) *st.repair = 0;           // (st.flags is sticky across batches: read and cleared by the host)
    if (q < qi.nq) {
      const float g_f32 = qi.gstat3[0], g_bf = qi.gstat3[1], g_diff = qi.gstat3[2];
      float eps;
      if (qi.use_img_terms) eps = qi.gamma * rsd.norm_img * g_bf + rsd.norm_img * g_diff + rsd.norm_diff * g_f32;
      else eps = qi.gamma * rsd.norm_f32 * g_f32;
      float margin = 2.0f * eps * 1.001f + 1e-30f;
      float thr0 = -INFINITY;
      if (!(margin == margin)) margin = 0.f;
      if (qi.use_img_terms && !isfinite(rsd.norm_img) && isfinite(rsd.norm_f32)) {
        atomicOr(st.flags, FLAG_RANGE);
        margin = 0.f;
        thr0 = INFINITY;
      }
      st.thr[q] = thr0;
      st.thr2[q] = thr0;
      st.qflag[q] = 0;
      st.margin[q] = margin;
      st.cnt[q * CNT_STRIDE] = qi.first_cnt;
    } else {
      st.thr[q] = INFINITY;
      st.thr2[q] = INFINITY;
      st.qflag[q] = 0;
      st.margin[q] = 0.f;
      st.cnt[q * CNT_STRIDE] = 0;
    }
    if (st.lad_tc) {
      st.lad_tc[q] = INFINITY;
      st.lad_pack[q] = 0x7F807F80u;
      st.lad_cnt[q] = 0;
    }
  }
  }   // rows of this workgroup
}

// ------------------------------------------------------------------------------------------------
// Gallery ingest, rows contiguous in memory (round 5): ONE WAVE PER ROW.
//
// The kernel above gives a row to a 256-thread workgroup: 8 elements per thread, and around them ~850 instructions per thread
// and row -- five barriers, three block reductions, address arithmetic, the float64 square root and division repeated by
// every thread -- i.e. ~3400 wave-instructions per row, which is what its 5 ms for 1 M rows were made of: it was ISSUE-bound
// (982 rows per SIMD x 3400 x ~3.5 cycles = 4.9 ms), not memory-bound.  Here a wave owns a row, a lane 32 of its elements: the
// per-row overhead is paid once per 64 lanes instead of once per 256 threads, there is no barrier (the four waves of a
// workgroup are independent; they take the four consecutive rows of a run, so the image still fills 256-byte stretches), and
// loads and stores are 16 bytes per lane.
//
// THE SUMS ARE THE SAME SUMS, bit for bit (tests/golden/ingest_checksums.json pins the galleries of the old kernel).  The old
// kernel's thread t (0..255) is a "virtual thread" here:
//   norm     virtual thread t sums x[t + 256 j]^2, j = 0, 1, ... in this order (fma chain), then the 64 virtual lanes t & 63 of
//            virtual wave t >> 6 run a butterfly with partners 32, 16, 8, 4, 2, 1, then ((w0 + w1) + w2) + w3.
//            Lane l loads float4s at columns 4 l + 256 i: it holds virtual threads 4 l + e (e = 0..3) with all their j = i.  Virtual
//            lane = 4 (l & 15) + e, virtual wave = l >> 4: partners 32, 16, 8, 4 are lanes l ^ 8, 4, 2, 1 -- data-parallel moves
//            inside the 16-lane rows --, partners 2 and 1 are the lane's own e ^ 2, e ^ 1.
//   stats    virtual thread t sums over the 8 columns 8 t .. 8 t + 7 (then 2048 + 8 t ... for rows wider than 2048), same
//            butterfly.  After the regrouping through the wave's own LDS row lane l takes chunks t = l + 64 k (k = 0..3):
//            virtual lane = l, virtual wave = k -- the plain wave butterfly on 3 x 4 accumulators.
// f32 rows are stored straight from the load layout (16 bytes per lane, 1 KiB per instruction).
// FULL: d == 256 * PT, rows 16-byte aligned, coop -- every column of every float4 is inside the row: no per-element selects, no
//       per-wave image stores compiled in; 152 registers: three waves per SIMD (the generic instantiations take what they need).
// PREFETCH: the wave's next row is requested into the registers of the current one as soon as those are consumed.
// Launch-to-launch, the same launch on the same box takes 3.5-4.1 ms depending on where the driver put the three buffers (one
// process: +-0.5 %; scripts/ingest_context_probe.py, profiles/r05o_*): compare kernels inside ONE process (scripts/ingest_ab.sh).
template <typename InT, int PT, bool PREFETCH, bool FULL>
__global__ __launch_bounds__(256, FULL ? 3 : 1) void ingest_rows_kernel(const InT* __restrict__ src, int64_t n, int32_t d, int64_t rs, int norm_mode,
                                                          float* __restrict__ out_f32, uint16_t* __restrict__ out_img,
                                                          int img_f16, RowStat* __restrict__ rowstat, int32_t dp,
                                                          int64_t nrows, int64_t row_base, int vec_ok, int coop_arg, int xcd_walk) {
  const bool coop = FULL || coop_arg;                // FULL: launched for coop only -- the per-wave image stores are not even compiled in
  __shared__ __attribute__((aligned(16))) float wbuf[4][1024];
  // coop (row_base and nrows multiples of 4): the 16-bit image of a run leaves through the workgroup -- [slice][row of the run][64 B]
  // is, slice by slice, the 256 contiguous bytes the run owns in that slice's block of the tile, so after one barrier every wave
  // copies 16 slices out with 256 contiguous bytes per 16 lanes.  Written by each wave for its own row, a row's piece of a
  // slice block is 64 bytes -- HALF a line, whose other half another wave writes whenever it gets there: 4.1 GB of image cost as
  // much time as the 8.2 GB of f32 rows (profiles/r05d_ingest_probes.txt).  Two buffers, by run parity: one barrier per run.
  __shared__ __attribute__((aligned(16))) uint4 wimg[2][PT * 8][4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* buf = wbuf[wv];
  struct Raw { InT v[4]; };
  Raw cur[PT];
  // loads are unconditional (row and column clamped into the source; the zero is selected where the value is USED): a load
  // behind a branch makes the number of outstanding loads path-dependent and the compiler then waits right behind the request
  auto request = [&](Raw* dst, int64_t r) {
    const int64_t rr = r < n ? r : n - 1;
    const InT* rowp = src + rr * rs + 4 * lane;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = 4 * lane + 256 * i;
      if (FULL || vec_ok) {                      // d % 4 == 0 and 16-byte aligned rows: a group of 4 is all inside or all outside
        const InT* p = (FULL || c < d) ? rowp + 256 * i : rowp - 4 * lane;
        if constexpr (sizeof(InT) == 4) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          dst[i].v[0] = t.x; dst[i].v[1] = t.y; dst[i].v[2] = t.z; dst[i].v[3] = t.w;
        } else {
          const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
          dst[i].v[0] = a.x; dst[i].v[1] = a.y; dst[i].v[2] = b.x; dst[i].v[3] = b.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[i].v[e] = rowp[(c + e < d ? c + e : d - 1) - 4 * lane];
      }
    }
  };
  const int nslices = dp / SLICE_K;
  const int64_t nruns = (nrows + 3) / 4;
  // xcd_walk (grid a multiple of 8): the workgroups of one XCD label (b, b + 8, ...: round-robin dispatch, used for speed only)
  // walk ONE contiguous eighth of the runs (whole tiles) side by side instead of every eighth run of the whole gallery: an
  // XCD's L2 and translation caches see an eighth of the pages
  int64_t run, run_hi, step;
  if (xcd_walk) {
    const int64_t per_xcd = ((nruns + 7) / 8 + 63) / 64 * 64;
    const int64_t lo = (int64_t)(blockIdx.x & 7) * per_xcd;
    run_hi = lo + per_xcd < nruns ? lo + per_xcd : nruns;
    run = lo + (blockIdx.x >> 3);
    step = gridDim.x >> 3;
  } else {
    run = blockIdx.x, run_hi = nruns, step = gridDim.x;
  }
  if (run >= run_hi) return;
  request(cur, run * 4 + wv);
  for (int par = 0; run < run_hi; run += step, par ^= 1) {
    const int64_t row = run * 4 + wv;
    const int64_t next_row = run + step < run_hi ? (run + step) * 4 + wv : row;
    bool requested = false;
    const int64_t orow = row_base + row;
    const int64_t tileidx = orow / TILE;
    const uint32_t r = (uint32_t)(orow % TILE);
    uint16_t* tile_base = out_img + tileidx * nslices * (int64_t)SLICE_ELEMS + (int64_t)r * SLICE_K;
    if (row < nrows && row >= n) {
      // padding row of the last tile: a zero image row and zero norms, no f32 row (what the sums below give for a row of zeros)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int h = 0; h < PT / 8; ++h) {
          const int c0 = 8 * (lane + 64 * k) + 2048 * h;
          if (c0 < dp) {
            const uint32_t sl = (uint32_t)c0 / SLICE_K, ch = ((uint32_t)c0 % SLICE_K) >> 3;
            if (coop) wimg[par][sl][wv ^ (sl & 3)][swz_chunk(r, ch)] = make_uint4(0, 0, 0, 0);
            else *reinterpret_cast<uint4*>(tile_base + (int64_t)sl * SLICE_ELEMS + (swz_chunk(r, ch) << 3)) = make_uint4(0, 0, 0, 0);
          }
        }
      if (lane < 3) reinterpret_cast<float*>(rowstat + orow)[lane] = 0.0f;
    } else if (row < nrows) {                                           // (wave-uniform)
      // ---- norm
      double scale = 1.0;
      if (norm_mode != 0) {
        double ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double v = (FULL || 4 * lane + 256 * i + e < d) ? (double)cur[i].v[e] : 0.0;
            ss[e] = __builtin_fma(v, v, ss[e]);
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ss[e] += xor_lane_f64<8>(ss[e]);
          ss[e] += xor_lane_f64<4>(ss[e]);
          ss[e] += xor_lane_f64<2>(ss[e]);
          ss[e] += xor_lane_f64<1>(ss[e]);
        }
        const double vw = (ss[0] + ss[2]) + (ss[1] + ss[3]);            // partners 2 and 1 of the virtual lane
        auto lane_value = [&](int src_lane) {
          return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(vw), src_lane),
                                  __builtin_amdgcn_readlane(__double2loint(vw), src_lane));
        };
        const double tot = ((lane_value(0) + lane_value(16)) + lane_value(32)) + lane_value(48);
        const double nrm = sqrt(tot);
        scale = (norm_mode == 1) ? 1.0 / nrm : 1.0 / (nrm + 1e-6);      // norm 0 -> inf -> NaN row, like the reference
      }
      // ---- the row in QUARTERS of 1024 columns (four float4 per lane): scaled values to memory from the load layout (16 bytes
      // per lane, 1 KiB per instruction) and into the wave's 4 KiB of LDS for the regrouping; then lane l takes the 8-column
      // chunks l and l + 64 of the quarter -- chunks l + 64 k (k = 2 (q & 1), + 1) of the row's 2048-column block q >> 1: virtual
      // lane l, virtual wave k, blocks ascending: the sums of the old kernel's virtual threads, add for add.
      // (A quarter at a time: 16 KiB of LDS per workgroup instead of 32, and three workgroups per CU instead of two -- half as
      // many bytes again in flight.  Every quarter's values sit in registers of their own: a store's data registers are busy
      // until the store has COMPLETED -- vmcnt, in order, behind the prefetch loads -- so re-using them for the next quarter
      // made the wave wait for the next row's loads in the middle of this row.)
      float* orow_p = out_f32 + orow * dp + 4 * lane;
      double s_g[4] = {0.0, 0.0, 0.0, 0.0}, s_b[4] = {0.0, 0.0, 0.0, 0.0}, s_d[4] = {0.0, 0.0, 0.0, 0.0};
      float4 y[PT];
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        const int c = 4 * lane + 256 * i;
        y[i].x = (FULL || c + 0 < d) ? (float)((double)cur[i].v[0] * scale) : 0.0f;
        y[i].y = (FULL || c + 1 < d) ? (float)((double)cur[i].v[1] * scale) : 0.0f;
        y[i].z = (FULL || c + 2 < d) ? (float)((double)cur[i].v[2] * scale) : 0.0f;
        y[i].w = (FULL || c + 3 < d) ? (float)((double)cur[i].v[3] * scale) : 0.0f;
      }
      // the next row of this wave is requested INTO `cur` as soon as its values are consumed: in flight during the image pass,
      // the stores and the copy-out, at no register cost (a second set of 32 registers held 3 waves per SIMD down to 2)
      if (PREFETCH) {
        request(cur, next_row < nrows ? next_row : row);                  // past the end: a harmless re-load
        requested = true;
      }
      auto image_pass = [&](auto f16_tag) {
        constexpr bool F16 = decltype(f16_tag)::value;
#pragma unroll
        for (int q = 0; q < PT / 4; ++q) {
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) *reinterpret_cast<float4*>(buf + 4 * (lane ^ ((lane >> 4) & 1)) + 256 * ii) = y[4 * q + ii];
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
            if ((FULL || 4 * lane + 256 * (4 * q + ii) < dp) && !(MI_INGEST_PROBE & 2))
              *reinterpret_cast<float4*>(orow_p + 256 * (4 * q + ii)) = y[4 * q + ii];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int k = 2 * (q & 1) + kk;
            const int c0 = 8 * (lane + 64 * k) + 2048 * (q >> 1);
            if (FULL || c0 < dp) {
              // (round 6) 16-byte chunk c of the quarter sits at c ^ ((c >> 4) & 1): the two reads below, 32 bytes apart from lane
              // to lane, were 2-way bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.38, profiles/r05t_pmc_summary.json)
              const float* cp = buf + 512 * kk;
              const int sw = (lane >> 3) & 1;
              const float4 lo = *reinterpret_cast<const float4*>(cp + 4 * ((2 * lane) ^ sw)),
                           hi = *reinterpret_cast<const float4*>(cp + 4 * ((2 * lane + 1) ^ sw));
              const float vf[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
              union { uint16_t hh[8]; uint4 u; } pk;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                double vb;
                pk.hh[e] = cvt_img(vf[e], F16 ? 1 : 0, vb);
#if !(MI_INGEST_PROBE & 1)
                s_b[k] = __builtin_fma(vb, vb, s_b[k]);
                const double df = vb - (double)vf[e];
                s_d[k] = __builtin_fma(df, df, s_d[k]);
                s_g[k] = __builtin_fma((double)vf[e], (double)vf[e], s_g[k]);
#endif
              }
              const uint32_t sl = (uint32_t)c0 / SLICE_K, ch = ((uint32_t)c0 % SLICE_K) >> 3;
#if !(MI_INGEST_PROBE & 4)
              // (round 6) row slot wv ^ (sl & 3): the four slices a 16-lane group writes no longer share their four 16-byte slots
              if (coop) wimg[par][sl][wv ^ (sl & 3)][swz_chunk(r, ch)] = pk.u;
              else *reinterpret_cast<uint4*>(tile_base + (int64_t)sl * SLICE_ELEMS + (swz_chunk(r, ch) << 3)) = pk.u;
#else
              if (pk.u.x == 0x12345678u) *reinterpret_cast<uint4*>(tile_base) = pk.u;
#endif
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();                        // the quarter's LDS is rewritten by the next one
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      };
      if (img_f16) image_pass(std::true_type{});
      else image_pass(std::false_type{});
      // the butterflies of the four virtual waves, then ((w0 + w1) + w2) + w3: lanes 0 / 8 / 4 end up with the g / b / d totals;
      // one square root for the three norms, every one of those lanes stores its float
      const double mine = reduce_stats12(s_g, s_b, s_d, lane);
      const float nf = (float)(sqrt(mine) * (1.0 + 1e-6));      // rounded UP a little: upper bounds after the f32 conversion
      if (lane == 0 || lane == 8 || lane == 4) reinterpret_cast<float*>(rowstat + orow)[lane == 0 ? 0 : (lane == 8 ? 1 : 2)] = nf;
    }
    if (coop && !(MI_INGEST_PROBE & 4)) {
      // (`run` is the same for the four waves of the workgroup: the barrier is uniform; a last run of fewer than four rows --
      // an appended block of any length -- copies its own rows only)
      __syncthreads();
      const int64_t orow0 = row_base + run * 4;                 // first row of the run: a multiple of 4, so the run shares one
      const uint32_t r0 = (uint32_t)(orow0 % TILE);             // tile and one swizzle (swz_chunk depends on row >> 2)
      uint16_t* run_base = out_img + (orow0 / TILE) * nslices * (int64_t)SLICE_ELEMS + (int64_t)r0 * SLICE_K;
      const bool row_ok = run * 4 + ((lane & 15) >> 2) < nrows;
      uint4 piece[PT / 2];
#pragma unroll
      for (int j = 0; j < PT / 2; ++j) {
        const int sl = (PT * 2) * wv + 4 * j + (lane >> 4);
        piece[j] = (&wimg[par][sl][0][0])[(lane & 15) ^ ((sl & 3) << 2)];
      }
#pragma unroll
      for (int j = 0; j < PT / 2; ++j) {
        const int sl = (PT * 2) * wv + 4 * j + (lane >> 4);      // wave w: slices [PT * 2 * w, PT * 2 * (w + 1))
        if (sl < nslices && row_ok) *reinterpret_cast<uint4*>(run_base + (int64_t)sl * SLICE_ELEMS + ((lane & 15) << 3)) = piece[j];
      }
    }
    if (!requested && run + step < run_hi) request(cur, next_row < nrows ? next_row : row);
  }
}

// ------------------------------------------------------------------------------------------------
// Gallery ingest of the reference's own layout (round 5): a [D, N] array whose ROWS OF THE GALLERY are the contiguous axis --
// callers hand over `vecs.T` (src/test_rOP1m.py:136-139,155-157; src/offline.py:107-118; src/online.py:95-102,132): element
// (row, col) at src[row + col * cs].  One pass, producer / consumer inside the workgroup (12 waves):
//   * four LOADER waves fetch the next PANEL -- 64 bytes worth of consecutive rows (16 float32 / 8 float64) x all 2048 columns,
//     16 bytes per lane, of whatever alignment the column stride gives -- into their REGISTERS (128 per lane: the whole 128 KiB
//     panel is in flight while the current one is processed) and, once the consumers are done with the current panel,
//     transpose it into LDS as row-major rows;
//   * eight CONSUMER waves run the body of ingest_rows_kernel on the rows of the panel in LDS, one wave per row, two runs of
//     four rows at a time: the 32 values per lane come from LDS instead of global memory -- the same virtual threads, the same
//     sums, the same bits -- the scaled row overwrites the raw row in place (that IS the regrouping buffer), the 16-bit image
//     leaves through the workgroup 256 bytes at a time.
// Loads and stores never share a wave: a wave's vector-memory operations retire in order, and with both in one instruction stream
// either the loads of the next panel waited behind this panel's stores or the compiler drained everything at the loop head (the
// one-role versions of this kernel ran their load phase and their store phase one after the other: 5.9-7.8 ms against 1.6 +
// 2.7 ms for the phases alone, profiles/r05e_ingest_cols_probes.txt).
// A panel's 64-byte column segments are HALF lines: the workgroups of one XCD label (b, b + 8, ...: observed round-robin
// dispatch, used for speed only) walk ONE contiguous range of panels side by side, so a line is fetched from memory once and
// its other half is an L2 hit a moment later.  (32-byte segments with two panel buffers in LDS were tried: twice the line
// requests per byte, 5.2 ms.)
// LDS row-major with the column index XOR-ed by 8 * (row quad): the transposing 4-byte writes of a loader wave -- 4 row quads x
// 8 columns per 32 lanes -- then hit 32 different banks, and the 16-byte reads of the consumers stay aligned permutations.
template <typename InT, bool COOP>
__global__ __launch_bounds__(768) void ingest_cols_kernel(const InT* __restrict__ src, int64_t n, int64_t cs, int norm_mode,
                                                          float* __restrict__ out_f32, uint16_t* __restrict__ out_img,
                                                          int img_f16, RowStat* __restrict__ rowstat, int64_t nrows,
                                                          int64_t row_base) {
  constexpr bool coop = COOP;                                    // (row_base % 4 == 0: the image leaves through the workgroup)
  constexpr int D = 2048, PT = 8, NSL = D / SLICE_K;
  constexpr int R = 64 / (int)sizeof(InT);                       // rows of a panel: 16 (float32) / 8 (float64)
  constexpr int RQ = R / 4;                                      // row quads = runs of a panel: 4 / 2
  constexpr size_t PANEL_BYTES = (size_t)R * D * sizeof(InT);    // 128 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [R][D] panel | image staging of the two runs in flight [run][slice][row of the run][64 B] (32 KiB)
  InT* panel = reinterpret_cast<InT*>(smem);
  uint4 (*wimg)[NSL][4][4] = reinterpret_cast<uint4 (*)[NSL][4][4]>(smem + PANEL_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wv12 = tid >> 6;
  const bool loader = wv12 >= 8;
  const int64_t npanels = (nrows + R - 1) / R;
  const int64_t nx = gridDim.x >> 3;                             // workgroups per XCD label (the grid is a multiple of 8)
  const int64_t xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int64_t per_x = (npanels + 7) / 8;
  const int64_t p_begin = xcd * per_x + kx, p_end = min((xcd + 1) * per_x, npanels);
  const int64_t p_step = nx;
  // The panel that holds the last row -- and the panels of padding rows behind it -- load the LAST R rows of the source instead
  // (n >= R): row k of such a panel sits `shift` slots further down; nothing is read past the end, there is no element-wise path.
  auto panel_shift = [&](int64_t pp) -> int { return pp * R + R <= n ? 0 : (int)(pp * R - (n - R)); };
  auto slot_swz = [&](int slot) -> int { return 8 * ((slot >> 2) & 3); };

  // ---- loaders: lane -> (row quad, column of a group of 64 / RQ columns); four waves cover 4 * 64 / RQ columns per round
  constexpr int CPW = 64 / RQ;                                   // columns per wave instruction: 16 / 32
  constexpr int NLD = D / (4 * CPW);                             // quads of 4 rows per thread: 32 / 16 (128 registers)
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  // (The loaders run a loop of their own -- same barriers, count for count: s_barrier counts arrivals, not program counters --
  // so that their 128 panel registers are live in their instruction stream only.)
  const int lw = wv12 - 8, lrq = lane % RQ, lcsub = lane / RQ;
  auto request = [&](InT (*t)[4], int64_t pp) {
    const int rq = lrq, csub = lcsub;
    const int64_t first = pp * R + R <= n ? pp * R : n - R;
    const InT* cp = src + (int64_t)(lw * CPW + csub) * cs + first + 4 * rq;
#pragma unroll
    for (int j = 0; j < NLD; ++j, cp += (int64_t)4 * CPW * cs) {
#if MI_INGEST_PROBE & 16384
      for (int e = 0; e < 4; ++e) t[j][e] = (InT)(first + j + e);
#else
      if constexpr (sizeof(InT) == 4) {
        const f4u x = *reinterpret_cast<const f4u*>(cp);
        t[j][0] = x.x; t[j][1] = x.y; t[j][2] = x.z; t[j][3] = x.w;
      } else {
        const d2u a = *reinterpret_cast<const d2u*>(cp), c2 = *reinterpret_cast<const d2u*>(cp + 2);
        t[j][0] = a.x; t[j][1] = a.y; t[j][2] = c2.x; t[j][3] = c2.y;
      }
#endif
    }
  };
  auto to_lds = [&](InT (*t)[4]) {
    const int rq = lrq, csub = lcsub;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int col = (j * 4 + lw) * CPW + csub;
      const int pc = col ^ (8 * (rq & 3));
#pragma unroll
      for (int e = 0; e < 4; ++e) panel[(4 * rq + e) * D + pc] = t[j][e];
    }
  };

  // ---- consumers: two runs at a time; wave wv of the pair's run pr takes row 4 (j0 + pr) + wv of the panel
  const int wv = wv12 & 3, pr = (wv12 >> 2) & 1;
  auto process_row = [&](int64_t p, int j0) {
    const int64_t prow0 = p * R;
    const int64_t row = prow0 + 4 * (j0 + pr) + wv;
    if (row >= nrows) return;
    const int64_t orow = row_base + row;
    const uint32_t r = (uint32_t)(orow % TILE);
    uint16_t* tile_base = out_img + (orow / TILE) * NSL * (int64_t)SLICE_ELEMS + (int64_t)r * SLICE_K;
    if (row >= n) {
      // padding row of the last tile: a zero image row and zero norms, no f32 row
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c0 = 8 * (lane + 64 * k);
        const uint32_t sl = (uint32_t)c0 / SLICE_K, ch = ((uint32_t)c0 % SLICE_K) >> 3;
        if (coop) wimg[pr][sl][wv ^ (sl & 3)][swz_chunk(r, ch)] = make_uint4(0, 0, 0, 0);
        else *reinterpret_cast<uint4*>(tile_base + (int64_t)sl * SLICE_ELEMS + (swz_chunk(r, ch) << 3)) = make_uint4(0, 0, 0, 0);
      }
      if (lane < 3) reinterpret_cast<float*>(rowstat + orow)[lane] = 0.0f;
      return;
    }
    const int slot = 4 * (j0 + pr) + wv + panel_shift(p);        // < R for every row < n
    const int swz = slot_swz(slot);
    InT* prow = panel + (size_t)slot * D;
    float* buf = reinterpret_cast<float*>(prow);                // the scaled row replaces the raw one (first D * 4 bytes of its slot)
    InT cur[PT][4];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const InT* q = prow + ((4 * lane + 256 * i) ^ swz);
      if constexpr (sizeof(InT) == 4) {
        const float4 x = *reinterpret_cast<const float4*>(q);
        cur[i][0] = x.x; cur[i][1] = x.y; cur[i][2] = x.z; cur[i][3] = x.w;
      } else {
        const double2 a = *reinterpret_cast<const double2*>(q), c2 = *reinterpret_cast<const double2*>(q + 2);
        cur[i][0] = a.x; cur[i][1] = a.y; cur[i][2] = c2.x; cur[i][3] = c2.y;
      }
    }
    double scale = 1.0;
    if (norm_mode != 0) {
      double ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double v = (double)cur[i][e];
          ss[e] = __builtin_fma(v, v, ss[e]);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ss[e] += xor_lane_f64<8>(ss[e]);
        ss[e] += xor_lane_f64<4>(ss[e]);
        ss[e] += xor_lane_f64<2>(ss[e]);
        ss[e] += xor_lane_f64<1>(ss[e]);
      }
      const double vw = (ss[0] + ss[2]) + (ss[1] + ss[3]);
      auto lane_value = [&](int src_lane) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(vw), src_lane),
                                __builtin_amdgcn_readlane(__double2loint(vw), src_lane));
      };
      const double tot = ((lane_value(0) + lane_value(16)) + lane_value(32)) + lane_value(48);
      const double nrm = sqrt(tot);
      scale = (norm_mode == 1) ? 1.0 / nrm : 1.0 / (nrm + 1e-6);
    }
    if constexpr (sizeof(InT) == 8) {                             // the float32 row is shorter than the float64 one it overwrites:
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // every lane has read its raw values before any lane writes
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float* orow_p = out_f32 + orow * D + 4 * lane;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      float4 y;
      y.x = (float)((double)cur[i][0] * scale);
      y.y = (float)((double)cur[i][1] * scale);
      y.z = (float)((double)cur[i][2] * scale);
      y.w = (float)((double)cur[i][3] * scale);
      const int wc4 = ((4 * lane + 256 * i) ^ swz) >> 2;              // 16-byte chunk of the scaled row; stored at c ^ ((c >> 4) & 1):
      *reinterpret_cast<float4*>(buf + 4 * (wc4 ^ ((wc4 >> 4) & 1))) = y;   // see ingest_rows_kernel (every lane has read its raw values)
      if (!(MI_INGEST_PROBE & 2)) *reinterpret_cast<float4*>(orow_p + 256 * i) = y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double s_g[4] = {0.0, 0.0, 0.0, 0.0}, s_b[4] = {0.0, 0.0, 0.0, 0.0}, s_d[4] = {0.0, 0.0, 0.0, 0.0};
    auto image_pass = [&](auto f16_tag) {
      constexpr bool F16 = decltype(f16_tag)::value;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c0 = 8 * (lane + 64 * k);
        const int rc4 = (c0 ^ swz) >> 2, sw = (rc4 >> 4) & 1;
        const float4 lo = *reinterpret_cast<const float4*>(buf + 4 * (rc4 ^ sw)),
                     hi = *reinterpret_cast<const float4*>(buf + 4 * ((rc4 + 1) ^ sw));
        const float vf[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        union { uint16_t hh[8]; uint4 u; } pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          double vb;
          pk.hh[e] = cvt_img(vf[e], F16 ? 1 : 0, vb);
          s_b[k] = __builtin_fma(vb, vb, s_b[k]);
          const double df = vb - (double)vf[e];
          s_d[k] = __builtin_fma(df, df, s_d[k]);
          s_g[k] = __builtin_fma((double)vf[e], (double)vf[e], s_g[k]);
        }
        const uint32_t sl = (uint32_t)c0 / SLICE_K, ch = ((uint32_t)c0 % SLICE_K) >> 3;
        if (coop) wimg[pr][sl][wv ^ (sl & 3)][swz_chunk(r, ch)] = pk.u;
        else *reinterpret_cast<uint4*>(tile_base + (int64_t)sl * SLICE_ELEMS + (swz_chunk(r, ch) << 3)) = pk.u;
      }
    };
    if (img_f16) image_pass(std::true_type{});
    else image_pass(std::false_type{});
    const double mine = reduce_stats12(s_g, s_b, s_d, lane);
    const float nf = (float)(sqrt(mine) * (1.0 + 1e-6));
    if (lane == 0 || lane == 8 || lane == 4) reinterpret_cast<float*>(rowstat + orow)[lane == 0 ? 0 : (lane == 8 ? 1 : 2)] = nf;
  };
  auto copy_out = [&](int64_t p, int j0) {                       // the image of run j0 + pr: 256 contiguous bytes per slice and 16 lanes
    const int64_t run_row = p * R + 4 * (j0 + pr);
    if (!coop || run_row >= nrows || (MI_INGEST_PROBE & 4)) return;
    const int64_t orow0 = row_base + run_row;                    // a multiple of 4 (coop): one tile, one swizzle for the run
    const uint32_t r0 = (uint32_t)(orow0 % TILE);
    uint16_t* run_base = out_img + (orow0 / TILE) * NSL * (int64_t)SLICE_ELEMS + (int64_t)r0 * SLICE_K;
    if (run_row + ((lane & 15) >> 2) >= nrows) return;           // a last run of fewer than four rows copies its own rows only
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int sl = 16 * wv + 4 * jj + (lane >> 4);
      *reinterpret_cast<uint4*>(run_base + (int64_t)sl * SLICE_ELEMS + ((lane & 15) << 3)) =
          (&wimg[pr][sl][0][0])[(lane & 15) ^ ((sl & 3) << 2)];
    }
  };

  if (loader) {
    InT t[NLD][4];
    if (p_begin < p_end) {
      request(t, p_begin);
      to_lds(t);
    }
    for (int64_t p = p_begin; p < p_end; p += p_step) {
      __syncthreads();                                           // A: panel p is in LDS
      const bool more = p + p_step < p_end;
      if (more) request(t, p + p_step);                          // in flight, in registers, while panel p is processed
#pragma unroll
      for (int j0 = 0; j0 < RQ; j0 += 2) {
        __syncthreads();                                         // B
        __syncthreads();                                         // C (the last one: the consumers are done with the panel)
      }
      if (more) to_lds(t);
    }
    return;
  }
  for (int64_t p = p_begin; p < p_end; p += p_step) {
    __syncthreads();                                             // A
#pragma unroll
    for (int j0 = 0; j0 < RQ; j0 += 2) {
      if (!(MI_INGEST_PROBE & 8192)) process_row(p, j0);
      __syncthreads();                                           // B: the image of these two runs is staged
      copy_out(p, j0);
      __syncthreads();                                           // C: the staging may be rewritten / the panel replaced
    }
  }
}

// Any strides -> row-major rows [m][d] of the same element type, values untouched (generic shapes of the [D, N] layout and
// fully strided sources go through this copy into a scratch block and then through ingest_rows_kernel: same bits, any shape).
template <typename InT>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const InT* __restrict__ src, int64_t m, int32_t d, int64_t rs, int64_t cs,
                                                             InT* __restrict__ dst) {
  __shared__ InT tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
  const int64_t row0 = (int64_t)blockIdx.x * 32;
  const int col0 = blockIdx.y * 32;
  // read with the lanes along the axis that is contiguous in the source
  const bool rows_contig = rs <= cs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int64_t row = rows_contig ? row0 + tx : row0 + a;
    const int col = rows_contig ? col0 + a : col0 + tx;
    InT v = 0;
    if (row < m && col < d) v = src[row * rs + (int64_t)col * cs];
    if (rows_contig) tile[tx][a] = v; else tile[a][tx] = v;      // tile[row in tile][col in tile]
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int64_t row = row0 + a;
    const int col = col0 + tx;
    if (row < m && col < d) dst[row * d + col] = tile[a][tx];
  }
}

void launch_transpose_rows(const void* src, int dtype, int64_t m, int32_t d, int64_t rs, int64_t cs, void* dst, hipStream_t stream) {
  const dim3 grid((unsigned)((m + 31) / 32), (unsigned)((d + 31) / 32));
  if (dtype == 0)
    hipLaunchKernelGGL(transpose_rows_kernel<float>, grid, dim3(256), 0, stream, (const float*)src, m, d, rs, cs, (float*)dst);
  else
    hipLaunchKernelGGL(transpose_rows_kernel<double>, grid, dim3(256), 0, stream, (const double*)src, m, d, rs, cs, (double*)dst);
}

// does launch_ingest take this source as it is (one pass, canonical sums)?  Otherwise the caller copies the rows into a
// row-major scratch block first (launch_transpose_rows) and ingests that.
bool ingest_takes_layout(int32_t d, int64_t rs, int64_t cs) {
  const int32_t dp = (int32_t)round_up(d, BK);
  if (cs == 1) return true;                                      // rows contiguous: ingest_rows_kernel (dp <= 4096) / row-wise kernel
  if (rs == 1 && d == 2048) return true;                         // the reference's layout at the reference's width: ingest_cols_kernel
  return dp > 256 * QI_MAX_PER_THREAD;                           // very wide rows: the row-wise kernel takes any strides
}

bool launch_ingest_queries(const void* src, int dtype, int32_t nq, int32_t d, int64_t rs, int64_t cs, int norm_mode,
                           float* out_f32, void* out_img, int img_f16, RowStat* rowstat, int32_t dp, int32_t qpad,
                           const float* gstat3, float gamma, int use_img_terms, uint32_t first_cnt, const QueryState& st,
                           hipStream_t stream, uint32_t zero_scores) {
  if (dp > 256 * QI_MAX_PER_THREAD) return false;         // wider rows: the two-launch path
  QueryInit qi;
  qi.zero_scores = zero_scores;
  qi.gstat3 = gstat3;
  qi.gamma = gamma;
  qi.use_img_terms = use_img_terms;
  qi.first_cnt = first_cnt;
  qi.nq = nq;
  qi.st = st;
  qi.gallery_rows = 0;
#define MI_QI_LAUNCH(T, PT)                                                                                             \
  hipLaunchKernelGGL((ingest_query_kernel<T, true, PT>), dim3((unsigned)qpad), dim3(256), 0, stream, (const T*)src,    \
                     (int64_t)nq, d, rs, cs, norm_mode, out_f32, (uint16_t*)out_img, img_f16, rowstat, dp, qi, (int64_t)0)
  if (dtype == 0) { if (dp <= 2048) MI_QI_LAUNCH(float, 8); else MI_QI_LAUNCH(float, 16); }
  else { if (dp <= 2048) MI_QI_LAUNCH(double, 8); else MI_QI_LAUNCH(double, 16); }
#undef MI_QI_LAUNCH
  return true;
}

// max over valid rows of the three norms (non-finite rows -- zero-norm rows normalised to NaN -- skipped)
__global__ __launch_bounds__(256) void rowstat_max_kernel(const RowStat* __restrict__ rowstat, int64_t n,
                                                          float* __restrict__ out3) {
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const RowStat r = rowstat[i];
    if (isfinite(r.norm_f32) && isfinite(r.norm_img) && isfinite(r.norm_diff)) {
      m0 = fmaxf(m0, r.norm_f32); m1 = fmaxf(m1, r.norm_img); m2 = fmaxf(m2, r.norm_diff);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); m2 = fmaxf(m2, __shfl_xor(m2, o));
  }
  // one atomic per workgroup and value (one per wave put 12 k atomics on three addresses: 90 us for the 1 M-row gallery)
  __shared__ float part[3][4];
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = m0; part[1][threadIdx.x >> 6] = m1; part[2][threadIdx.x >> 6] = m2; }
  __syncthreads();
  if (threadIdx.x < 3) {           // non-negative floats order like their bit patterns
    const float m = fmaxf(fmaxf(part[threadIdx.x][0], part[threadIdx.x][1]), fmaxf(part[threadIdx.x][2], part[threadIdx.x][3]));
    atomicMax(reinterpret_cast<unsigned int*>(out3 + threadIdx.x), __float_as_uint(m));
  }
}

void launch_ingest(const void* src, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, int norm_mode,
                   float* out_f32, void* out_img, int img_f16, RowStat* rowstat, int32_t dp, int64_t npad,
                   hipStream_t stream, int64_t row_base) {
  if (cs == 1 && dp <= 256 * QI_MAX_PER_THREAD && npad < (int64_t)1 << 31 && !(MI_INGEST_PROBE & 512)) {
    // rows contiguous in memory (row-major source: device-generated galleries, appended descriptor batches, re-imaging of the
    // stored rows): one WAVE per row, one pass (round 5; the sums of the kernel below, bit for bit)
    const size_t esz = dtype == 0 ? 4 : 8;
    const int vec_ok = (d % 4 == 0) && ((rs * esz) % 16 == 0) && (((uintptr_t)src) % 16 == 0);
    const int coop = (row_base % 4 == 0) && !(MI_INGEST_PROBE & 2048);
#define MI_GR_LAUNCH(T, PT, PF, FULL)                                                                                    \
  do {                                                                                                                  \
    /* occupancy of this instantiation: the same on every (identical) device of the node; cached in an atomic */        \
    static std::atomic<int> occ_cache{0};                                                                               \
    int occ = occ_cache.load(std::memory_order_relaxed);                                                                \
    if (!occ) {                                                                                                         \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ingest_rows_kernel<T, PT, PF, FULL>, 256, 0) !=            \
              hipSuccess || occ < 1)                                                                                    \
        occ = 2;                                                                                                        \
      occ_cache.store(occ, std::memory_order_relaxed);                                                                  \
    }                                                                                                                   \
    if (MI_INGEST_PROBE && getenv("MI_INGEST_WG_PER_CU")) occ = atoi(getenv("MI_INGEST_WG_PER_CU"));                    \
    if (MI_INGEST_PROBE) fprintf(stderr, "ingest (wave per row): %d workgroups per CU\n", occ);                         \
    const unsigned grid = (unsigned)std::min<int64_t>((npad + 3) / 4, (int64_t)current_device_cus() * occ);             \
    const int xcd_walk = grid % 8 == 0 && (npad + 3) / 4 >= 8 * (int64_t)grid && !(MI_INGEST_PROBE & 32768);            \
    hipLaunchKernelGGL((ingest_rows_kernel<T, PT, PF, FULL>), dim3(grid), dim3(256), 0, stream, (const T*)src, n, d,    \
                       rs, norm_mode, out_f32, (uint16_t*)out_img, img_f16, rowstat, dp, npad, row_base, vec_ok, coop,  \
                       xcd_walk);                                                                                       \
  } while (0)
    if (dtype == 0) {
      if (d == 2048 && vec_ok && coop) MI_GR_LAUNCH(float, 8, true, true);  // the descriptors of the reference: 2048-d
      else if (dp <= 2048) MI_GR_LAUNCH(float, 8, false, false);
      else MI_GR_LAUNCH(float, 16, false, false);
    } else {
      if (d == 2048 && vec_ok && coop) MI_GR_LAUNCH(double, 8, false, true);
      else if (dp <= 2048) MI_GR_LAUNCH(double, 8, false, false);
      else MI_GR_LAUNCH(double, 16, false, false);
    }
#undef MI_GR_LAUNCH
    return;
  }
  if (rs == 1 && cs != 1 && d == 2048 && dp == 2048 && n >= 16 && !(MI_INGEST_PROBE & 4096)) {
    // the reference's [D, N] layout: one pass, rows rebuilt in LDS by loader waves, consumed by the row body (ingest_cols_kernel)
    const int coop = row_base % 4 == 0;
    const unsigned grid = (unsigned)std::max<int64_t>(8, current_device_cus() / 8 * 8);      // one workgroup per CU, whole XCD labels
    const size_t lds = (size_t)128 * 1024 + 32 * 1024;
#define MI_GC_LAUNCH(T, C)                                                                                               \
  do {                                                                                                                  \
    ensure_dynamic_lds((const void*)ingest_cols_kernel<T, C>, (int)lds);                                                \
    hipLaunchKernelGGL((ingest_cols_kernel<T, C>), dim3(grid), dim3(768), lds, stream, (const T*)src, n, cs, norm_mode, \
                       out_f32, (uint16_t*)out_img, img_f16, rowstat, npad, row_base);                                  \
  } while (0)
    if (dtype == 0) { if (coop) MI_GC_LAUNCH(float, true); else MI_GC_LAUNCH(float, false); }
    else { if (coop) MI_GC_LAUNCH(double, true); else MI_GC_LAUNCH(double, false); }
#undef MI_GC_LAUNCH
    return;
  }
  if (cs == 1 && dp <= 256 * QI_MAX_PER_THREAD && npad < (int64_t)1 << 31) {
    // rows contiguous in memory (row-major source: device-generated galleries, appended descriptor batches, re-imaging of
    // the stored rows): one workgroup per row, ONE pass over the source with the row held in registers -- a gallery is
    // read once (8 KiB per row) and written once (8 + 4 KiB) instead of being read twice through the 64 x 64 LDS tiles of
    // ingest_kernel, which exists for the reference's strided `vecs.T` views ([D, N] arrays: rs == 1)
    QueryInit none{};
    none.gallery_rows = npad;
    // persistent grid: eight 256-thread workgroups per CU (the kernel's occupancy), each looping over its rows
    // persistent grid = exactly the workgroups the device holds at once (a workgroup that starts late runs its share of the
    // rows at a lower occupancy: the launch is bound by the per-row latency chain, not by bytes); runs of 4 rows
#define MI_GI_LAUNCH(T, PT)                                                                                             \
  do {                                                                                                                  \
    static std::atomic<int> occ_cache{0};                                                                               \
    int occ = occ_cache.load(std::memory_order_relaxed);                                                                \
    if (!occ) {                                                                                                         \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ingest_query_kernel<T, false, PT>, 256, 0) !=              \
              hipSuccess || occ < 1)                                                                                    \
        occ = 4;                                                                                                        \
      occ_cache.store(occ, std::memory_order_relaxed);                                                                  \
    }                                                                                                                   \
    if (MI_INGEST_PROBE && getenv("MI_INGEST_WG_PER_CU")) occ = atoi(getenv("MI_INGEST_WG_PER_CU"));                    \
    if (MI_INGEST_PROBE) fprintf(stderr, "ingest: %d workgroups per CU\n", occ);                                        \
    const unsigned grid = (unsigned)std::min<int64_t>((npad + MI_INGEST_RUN - 1) / MI_INGEST_RUN,                          \
                                                      (int64_t)current_device_cus() * occ);                               \
    hipLaunchKernelGGL((ingest_query_kernel<T, false, PT>), dim3(grid), dim3(256), 0, stream, (const T*)src, n, d, rs,  \
                       cs, norm_mode, out_f32, (uint16_t*)out_img, img_f16, rowstat, dp, none, row_base);               \
  } while (0)
    if (dtype == 0) { if (dp <= 2048) MI_GI_LAUNCH(float, 8); else MI_GI_LAUNCH(float, 16); }
    else { if (dp <= 2048) MI_GI_LAUNCH(double, 8); else MI_GI_LAUNCH(double, 16); }
#undef MI_GI_LAUNCH
    return;
  }
  // anything else -- strided query batches, rows wider than 4096 columns: one workgroup per row, any strides, two passes over the
  // row (the same sums; galleries in a layout this would be slow for are copied into row-major scratch blocks by the caller,
  // ingest_takes_layout / launch_transpose_rows).  (The round-1 kernel that read 64 x 64 tiles of a strided source twice and
  // summed in an order of its own is gone: round 5.)
  if (dtype == 0)
    hipLaunchKernelGGL(ingest_rowwise_kernel<float>, dim3((unsigned)npad), dim3(256), 0, stream, (const float*)src, n,
                       d, rs, cs, norm_mode, out_f32, (uint16_t*)out_img, img_f16, rowstat, dp, npad, row_base);
  else
    hipLaunchKernelGGL(ingest_rowwise_kernel<double>, dim3((unsigned)npad), dim3(256), 0, stream, (const double*)src,
                       n, d, rs, cs, norm_mode, out_f32, (uint16_t*)out_img, img_f16, rowstat, dp, npad, row_base);
}

// Order-independent 64-bit checksum of a device buffer (8-byte words): sum over words of mix(word ^ index * golden ratio).
// Computed where the data lives: the prepared-gallery file stores one per section and the loader recomputes it AFTER the
// copy to the device, so that a flipped bit in the file, in the page cache or on the PCIe path is caught
// ("header (N, D, dtype, norm flag, checksum)").
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void checksum_kernel(const uint64_t* __restrict__ data, size_t nwords,
                                                       unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[256];
  unsigned long long s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
    s += mix64(data[i] ^ (i * 0x9E3779B97F4A7C15ull));
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, red[0]);
}
void launch_checksum(const void* data, size_t bytes, unsigned long long* out, hipStream_t stream) {
  hipMemsetAsync(out, 0, 8, stream);
  const size_t nwords = bytes / 8;
  if (!nwords) return;
  const unsigned blocks = (unsigned)std::min<size_t>(2048, (nwords + 255) / 256);
  hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, stream, (const uint64_t*)data, nwords, out);
}

void launch_rowstat_max(const RowStat* rowstat, int64_t n, float* out3, hipStream_t stream, bool reset) {
  if (reset) hipMemsetAsync(out3, 0, 3 * sizeof(float), stream);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rowstat_max_kernel, dim3(blocks), dim3(256), 0, stream, rowstat, n, out3);
}


// ------------------------------------------------------------------------------------------------
// Bootstrap sample image: n_s rows drawn one per stratum of N / n_s consecutive rows at a hashed offset, copied
// from the tile-blocked gallery image into a small tile-blocked image of its own.  The speculative threshold of the
// single-launch schedule is an order statistic of the scores of THESE rows, so the draw has to be representative
// of the whole shard whatever order the rows were ingested in (the reference's 1M gallery is [rOxford | distractors],
// src/test_rOP1m.py:136-139: a sample made of the first rows would contain every true positive of a query).
__host__ __device__ inline int64_t sample_source_row(int64_t i, int64_t n, int64_t n_s) {
  const int64_t lo = i * n / n_s, hi = (i + 1) * n / n_s;             // stratum [lo, hi), never empty for n >= n_s
  uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return lo + (int64_t)(h % (uint64_t)(hi - lo));
}

__global__ __launch_bounds__(256) void build_sample_kernel(const uint16_t* __restrict__ gal_img,
                                                           uint16_t* __restrict__ samp_img, int64_t n, int64_t n_s,
                                                           int32_t nslices) {
  const int64_t i = blockIdx.x;                                        // sample row
  const int64_t r = sample_source_row(i, n, n_s);
  const uint32_t ri = (uint32_t)(i % TILE), rr = (uint32_t)(r % TILE);
  const uint16_t* src = gal_img + (r / TILE) * nslices * (int64_t)SLICE_ELEMS + (int64_t)rr * SLICE_K;
  uint16_t* dst = samp_img + (i / TILE) * nslices * (int64_t)SLICE_ELEMS + (int64_t)ri * SLICE_K;
  for (uint32_t j = threadIdx.x; j < (uint32_t)nslices * 4u; j += blockDim.x) {
    const uint32_t sl = j >> 2, c = j & 3u;
    const uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)sl * SLICE_ELEMS + (swz_chunk(rr, c) << 3));
    *reinterpret_cast<uint4*>(dst + (int64_t)sl * SLICE_ELEMS + (swz_chunk(ri, c) << 3)) = v;
  }
}

void launch_build_sample(const void* gal_img, void* samp_img, int64_t n, int64_t n_s, int32_t dp, hipStream_t stream) {
  hipLaunchKernelGGL(build_sample_kernel, dim3((uint32_t)n_s), dim3(256), 0, stream,
                     reinterpret_cast<const uint16_t*>(gal_img), reinterpret_cast<uint16_t*>(samp_img), n, n_s,
                     dp / SLICE_K);
}

// the same sample as stored f32 rows (the f32 scorer takes its thresholds from exact scores of the sample rows, round 6)
__global__ __launch_bounds__(256) void build_sample_f32_kernel(const float* __restrict__ gal_f32, float* __restrict__ samp_f32,
                                                               int64_t n, int64_t n_s, int32_t dp) {
  const int64_t i = blockIdx.x;
  const float4* src = reinterpret_cast<const float4*>(gal_f32 + sample_source_row(i, n, n_s) * dp);
  float4* dst = reinterpret_cast<float4*>(samp_f32 + i * dp);
  for (int j = threadIdx.x; j < dp / 4; j += blockDim.x) dst[j] = src[j];
}

void launch_build_sample_f32(const float* gal_f32, float* samp_f32, int64_t n, int64_t n_s, int32_t dp, hipStream_t stream) {
  hipLaunchKernelGGL(build_sample_f32_kernel, dim3((uint32_t)n_s), dim3(256), 0, stream, gal_f32, samp_f32, n, n_s, dp);
}

int64_t sample_source_row_host(int64_t i, int64_t n, int64_t n_s) { return sample_source_row(i, n, n_s); }

}  // namespace mi
