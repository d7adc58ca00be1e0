// Exact top-k of dense f32 score rows: replaces `np.argsort(-scores)` / faiss IndexFlatIP.search when k is a
// large fraction of N (kNN graph of the diffusion re-ranking: k = 2000 of N ~ 5-11k, src/utils/diffusion.py:66;
// top-2000 of the diffused scores, src/utils/Reranking.py:250-253), where threshold filtering cannot help.
// One workgroup per row: 4-pass MSB radix select of the k-th largest key straight from global memory (the row
// stays in L2), a second select on the INDEX among the ties of that key (ties go to the lower index,
// deterministically), then the k chosen entries are bitonic-sorted in LDS by (score desc, idx asc).
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace mi {

// k-th largest (by key) of vals[0..n) restricted to entries where pred(i) holds.  KEYFN maps i -> uint32 key.
template <typename KeyFn>
__device__ uint32_t block_radix_kth_largest(KeyFn keyfn, uint32_t n, uint32_t K, uint32_t* hist, uint32_t* sh) {
  uint32_t prefix = 0, mask = 0, remaining = K;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      uint32_t k;
      if (keyfn(i, k) && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const uint32_t h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
      const uint32_t tot = h0 + h1 + h2 + h3;
      uint32_t suf = tot;
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_down(suf, o);
        if (l + o < 64) suf += v;
      }
      uint32_t c = suf - tot;
      const uint32_t hs[4] = {h3, h2, h1, h0};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c < remaining && c + hs[e] >= remaining) {
          sh[0] = (uint32_t)(4 * l + 3 - e);
          sh[1] = remaining - c;
        }
        c += hs[e];
      }
    }
    __syncthreads();
    prefix |= sh[0] << shift;
    mask |= 255u << shift;
    remaining = sh[1];
    __syncthreads();
  }
  return prefix;
}

template <typename ScoreT, typename IdT>
__device__ void bitonic_sort_desc_t(ScoreT* s, IdT* id, uint32_t n2) {
  for (uint32_t size = 2; size <= n2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (uint32_t t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
        const uint32_t lo = 2 * t - (t & (stride - 1));
        const uint32_t hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const ScoreT a = s[lo], b = s[hi];
        const IdT ia = id[lo], ib = id[hi];
        const bool a_nan = (a != a), b_nan = (b != b);
        bool a_first;
        if (a_nan || b_nan) a_first = (!a_nan) || (b_nan && ia < ib);
        else a_first = (a > b) || (a == b && ia < ib);
        if (a_first != desc) {
          s[lo] = b; s[hi] = a;
          id[lo] = ib; id[hi] = ia;
        }
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void dense_topk_kernel(const float* __restrict__ scores, int64_t ld, uint32_t n,
                                                         int32_t k, int64_t row_offset, int64_t* __restrict__ out_idx,
                                                         float* __restrict__ out_score) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t q = blockIdx.x;
  const float* row = scores + (uint64_t)q * ld;
  uint32_t k2 = 2;
  while (k2 < (uint32_t)k) k2 <<= 1;
  float* s = reinterpret_cast<float*>(smem);
  uint32_t* id = reinterpret_cast<uint32_t*>(smem + (size_t)k2 * 4);
  uint32_t* hist = id + k2;
  uint32_t* sh = hist + 256;
  // 1) key of the k-th largest score
  const uint32_t keyK = block_radix_kth_largest(
      [&](uint32_t i, uint32_t& key) { key = f2key(row[i]); return true; }, n, (uint32_t)k, hist, sh);
  // 2) how many strictly above, and the index threshold among the ties (the (k - above) smallest indices)
  if (threadIdx.x == 0) { sh[2] = 0; sh[3] = 0; }
  __syncthreads();
  uint32_t above_local = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) above_local += (f2key(row[i]) > keyK);
  for (int o = 32; o > 0; o >>= 1) above_local += __shfl_xor(above_local, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sh[2], above_local);
  __syncthreads();
  const uint32_t above = sh[2];
  const uint32_t need_ties = (uint32_t)k - above;          // >= 1
  __syncthreads();
  // (need_ties)-th smallest index among ties == (need_ties)-th largest of ~index
  const uint32_t invT = block_radix_kth_largest(
      [&](uint32_t i, uint32_t& key) { key = ~i; return f2key(row[i]) == keyK; }, n, need_ties, hist, sh);
  const uint32_t idxT = ~invT;
  // 3) collect exactly k entries
  for (uint32_t i = threadIdx.x; i < k2; i += blockDim.x) { s[i] = -INFINITY; id[i] = 0xFFFFFFFFu; }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = row[i];
    const uint32_t key = f2key(v);
    if (key > keyK || (key == keyK && i <= idxT)) {
      const uint32_t pos = atomicAdd(&sh[3], 1u);
      if (pos < k2) { s[pos] = v; id[pos] = i; }
    }
  }
  bitonic_sort_desc_t<float, uint32_t>(s, id, k2);
  for (uint32_t i = threadIdx.x; i < (uint32_t)k; i += blockDim.x) {
    out_idx[(uint64_t)q * k + i] = row_offset + (int64_t)id[i];
    if (out_score) out_score[(uint64_t)q * k + i] = s[i];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Last resort of the exhaustive search (csrc/api_schedule.hip search_sync): data with MASSIVE ties -- more rows within the error
// margin of the K-th score than any candidate buffer holds (thousands of near-duplicates of one image).  No filter can
// help there, so every score is computed at full precision (f32 stored rows, exact f64 products, f64 accumulation: the
// arithmetic of rescore_kernel) into a dense [queries][rows] f64 matrix and the exact top-k is selected from it.
// LDS-tiled 64 queries x 64 rows per workgroup, 4 x 4 per thread, v_fma_f64; throughput is irrelevant next to
// correctness here (the filtered path never gets this far on descriptor data).
__global__ __launch_bounds__(256) void dense_score64_kernel(const float* __restrict__ gal, const float* __restrict__ qry,
                                                            int32_t dp, int64_t n, int32_t nq, double* __restrict__ out,
                                                            int64_t ld) {
  __shared__ double Qs[16][65];
  __shared__ double Gs[16][65];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int q0 = blockIdx.y * 64;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  const int lr = t >> 2, lk = (t & 3) * 4;
  for (int k0 = 0; k0 < dp; k0 += 16) {           // dp is a multiple of 64
    __syncthreads();
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), qv = gv;
    if (row0 + lr < n) gv = *reinterpret_cast<const float4*>(gal + (uint64_t)(row0 + lr) * dp + k0 + lk);
    if (q0 + lr < nq) qv = *reinterpret_cast<const float4*>(qry + (uint64_t)(q0 + lr) * dp + k0 + lk);
    Gs[lk][lr] = (double)gv.x; Gs[lk + 1][lr] = (double)gv.y; Gs[lk + 2][lr] = (double)gv.z; Gs[lk + 3][lr] = (double)gv.w;
    Qs[lk][lr] = (double)qv.x; Qs[lk + 1][lr] = (double)qv.y; Qs[lk + 2][lr] = (double)qv.z; Qs[lk + 3][lr] = (double)qv.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = Qs[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Gs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = q0 + ty * 4 + i;
    if (q >= nq) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t r = row0 + tx * 4 + j;
      if (r < n) out[(uint64_t)q * ld + r] = acc[i][j];
    }
  }
}

// order-preserving 64-bit key of an f64 score: NaN lowest, -0 == +0
__device__ __forceinline__ uint64_t d2key(double d) {
  if (d != d) return 0ull;
  d += 0.0;
  const uint64_t u = (uint64_t)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// dense_topk_kernel on f64 rows: the 64-bit key of the k-th largest score is found in two 32-bit radix selects (high
// word, then low word among the rows that share it), the rest is the same (index select among exact ties, collect,
// bitonic sort by (score desc, idx asc)).
__global__ __launch_bounds__(512) void dense_topk64_kernel(const double* __restrict__ scores, int64_t ld, uint32_t n,
                                                           int32_t k, int64_t row_offset, int64_t* __restrict__ out_idx,
                                                           float* __restrict__ out_score, double* __restrict__ out_score64) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t q = blockIdx.x;
  const double* row = scores + (uint64_t)q * ld;
  uint32_t k2 = 2;
  while (k2 < (uint32_t)k) k2 <<= 1;
  double* s = reinterpret_cast<double*>(smem);
  uint32_t* id = reinterpret_cast<uint32_t*>(smem + (size_t)k2 * 8);
  uint32_t* hist = id + k2;
  uint32_t* sh = hist + 256;                     // [0,1] radix select; [2] above (high word); [3] collect; [4] above (key)
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t hiK = block_radix_kth_largest(
      [&](uint32_t i, uint32_t& key) { key = (uint32_t)(d2key(row[i]) >> 32); return true; }, n, (uint32_t)k, hist, sh);
  uint32_t cnt = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) cnt += ((uint32_t)(d2key(row[i]) >> 32) > hiK);
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sh[2], cnt);
  __syncthreads();
  const uint32_t need_lo = (uint32_t)k - sh[2];              // >= 1
  __syncthreads();
  const uint32_t loK = block_radix_kth_largest(
      [&](uint32_t i, uint32_t& key) {
        const uint64_t kk = d2key(row[i]);
        key = (uint32_t)kk;
        return (uint32_t)(kk >> 32) == hiK;
      },
      n, need_lo, hist, sh);
  const uint64_t keyK = ((uint64_t)hiK << 32) | loK;
  cnt = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) cnt += (d2key(row[i]) > keyK);
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sh[4], cnt);
  __syncthreads();
  const uint32_t need_ties = (uint32_t)k - sh[4];            // >= 1
  __syncthreads();
  const uint32_t invT = block_radix_kth_largest(
      [&](uint32_t i, uint32_t& key) { key = ~i; return d2key(row[i]) == keyK; }, n, need_ties, hist, sh);
  const uint32_t idxT = ~invT;
  for (uint32_t i = threadIdx.x; i < k2; i += blockDim.x) { s[i] = -INFINITY; id[i] = 0xFFFFFFFFu; }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = row[i];
    const uint64_t key = d2key(v);
    if (key > keyK || (key == keyK && i <= idxT)) {
      const uint32_t pos = atomicAdd(&sh[3], 1u);
      if (pos < k2) { s[pos] = v; id[pos] = i; }
    }
  }
  bitonic_sort_desc_t<double, uint32_t>(s, id, k2);
  for (uint32_t i = threadIdx.x; i < (uint32_t)k; i += blockDim.x) {
    out_idx[(uint64_t)q * k + i] = row_offset + (int64_t)id[i];
    if (out_score) out_score[(uint64_t)q * k + i] = (float)s[i];
    if (out_score64) out_score64[(uint64_t)q * k + i] = s[i];
  }
}

void launch_dense_score64(const float* gal_f32, const float* qry_f32, int32_t dp, int64_t n, int32_t nq, double* out,
                          int64_t ld, hipStream_t stream) {
  hipLaunchKernelGGL(dense_score64_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((nq + 63) / 64)), dim3(256), 0, stream,
                     gal_f32, qry_f32, dp, n, nq, out, ld);
}

void launch_dense_topk64(const double* scores, int64_t ld, int64_t n, int32_t nq, int32_t k, int64_t row_offset,
                         int64_t* out_idx, float* out_score, double* out_score64, hipStream_t stream) {
  uint32_t k2 = 2;
  while (k2 < (uint32_t)k) k2 <<= 1;
  const size_t lds = (size_t)k2 * 12 + 256 * 4 + 32;
  hipLaunchKernelGGL(dense_topk64_kernel, dim3(nq), dim3(512), lds, stream, scores, ld, (uint32_t)n, k, row_offset, out_idx,
                     out_score, out_score64);
}

// ------------------------------------------------------------------------------------------------------------------
// Full-length ranking: `np.argsort(-scores, axis=0)` over ALL rows (src/main_retrieve.py:176,
// src/utils/Reranking.py:207, --mode mAP of src/test_rOP1m.py:144-149).  One 1024-thread workgroup per query runs a
// stable LSD radix sort (4 passes of 8 bits) on key = ~f2key(score) with the row index as payload, so the result is
// (score descending, index ascending; NaN last).  Every wave owns a contiguous segment of the array: per-wave digit
// histograms -> offsets ordered (digit, wave) -> each wave scatters its own segment in order, ranking equal digits
// inside a 64-element step with ballots.  Waves never touch each other's offsets, so no sync inside the scatter.
constexpr int RANK_THREADS = 1024, RANK_WAVES = 16;

__global__ __launch_bounds__(RANK_THREADS) void rank_all_kernel(const float* __restrict__ scores, int64_t ld, uint32_t n,
                                                                 uint32_t* __restrict__ keys_a, uint32_t* __restrict__ idx_a,
                                                                 uint32_t* __restrict__ keys_b, uint32_t* __restrict__ idx_b,
                                                                 int64_t row_offset, int64_t* __restrict__ out_idx,
                                                                 float* __restrict__ out_score) {
  __shared__ uint32_t hist[RANK_WAVES][256];
  __shared__ uint32_t total[256];
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* row = scores + (uint64_t)q * ld;
  uint32_t* ka = keys_a + (uint64_t)q * n;
  uint32_t* ia = idx_a + (uint64_t)q * n;
  uint32_t* kb = keys_b + (uint64_t)q * n;
  uint32_t* ib = idx_b + (uint64_t)q * n;
  const uint32_t seg = (n + RANK_WAVES - 1) / RANK_WAVES;
  const uint32_t s0 = min(n, w * seg), s1 = min(n, s0 + seg);
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass * 8;
    const uint32_t* ksrc = (pass & 1) ? kb : ka;
    const uint32_t* isrc = (pass & 1) ? ib : ia;
    uint32_t* kdst = (pass & 1) ? ka : kb;
    uint32_t* idst = (pass & 1) ? ia : ib;
    for (uint32_t i = lane; i < 256; i += 64) hist[w][i] = 0;
    __syncthreads();
    for (uint32_t i = s0 + lane; i < s1; i += 64) {
      const uint32_t key = pass == 0 ? ~f2key(row[i]) : ksrc[i];
      atomicAdd(&hist[w][(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    // offsets in (digit, wave) order: thread d < 256 owns digit d
    if (threadIdx.x < 256) {
      uint32_t t = 0;
      for (int ww = 0; ww < RANK_WAVES; ++ww) t += hist[ww][threadIdx.x];
      total[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < 64) {                 // exclusive scan of 256 totals by one wave (4 per lane)
      const uint32_t t0 = total[4 * lane], t1 = total[4 * lane + 1], t2 = total[4 * lane + 2], t3 = total[4 * lane + 3];
      uint32_t sum = t0 + t1 + t2 + t3, inc = sum;
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
      }
      const uint32_t ex = inc - sum;
      total[4 * lane] = ex;
      total[4 * lane + 1] = ex + t0;
      total[4 * lane + 2] = ex + t0 + t1;
      total[4 * lane + 3] = ex + t0 + t1 + t2;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      uint32_t run = total[threadIdx.x];
      for (int ww = 0; ww < RANK_WAVES; ++ww) {
        const uint32_t c = hist[ww][threadIdx.x];
        hist[ww][threadIdx.x] = run;
        run += c;
      }
    }
    __syncthreads();
    // stable scatter of this wave's segment
    for (uint32_t base = s0; base < s1; base += 64) {
      const uint32_t i = base + lane;
      const bool valid = i < s1;
      uint32_t key = 0, id = 0;
      if (valid) {
        key = pass == 0 ? ~f2key(row[i]) : ksrc[i];
        id = pass == 0 ? i : isrc[i];
      }
      const uint32_t d = (key >> shift) & 255u;
      unsigned long long same = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot((d >> b) & 1u);
        same &= ((d >> b) & 1u) ? m : ~m;
      }
      if (valid) {
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        const uint32_t pos = hist[w][d] + rank;
        kdst[pos] = key;
        idst[pos] = id;
      }
      // the last lane of every digit group advances that digit's offset (after all lanes of the wave have read it)
      if (valid && (same >> lane) == 1ull) hist[w][d] += (uint32_t)__popcll(same);
    }
    __syncthreads();
  }
  // 4 passes: the result is back in the A buffers
  for (uint32_t i = threadIdx.x; i < n; i += RANK_THREADS) {
    const uint32_t id = ia[i];
    out_idx[(uint64_t)q * n + i] = row_offset + (int64_t)id;
    if (out_score) out_score[(uint64_t)q * n + i] = row[id];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Positions of a FEW listed rows in the full ranking of every query, without producing the ranking: position of row r =
// #{ j : key(j) < key(r)  or  (key(j) == key(r) and j < r) } with key = ~f2key(score), i.e. exactly the place
// rank_all_kernel would give it (score descending, index ascending, NaN last).  This is all the revisited mAP protocol
// needs of `ranks_aqe` [N, Q] (src/utils/Reranking.py:207, 280-283; src/utils/evaluate2.py:73-86 looks up the
// positions of the positive and junk images only).  grid = (row chunks, queries); the m listed keys sit in LDS.
constexpr int POS_THREADS = 256, POS_MAX_LISTED = 2048;
__global__ __launch_bounds__(POS_THREADS) void rank_positions_kernel(const float* __restrict__ scores, int64_t ld,
                                                                      uint32_t n, const int64_t* __restrict__ ids,
                                                                      int32_t m, int64_t row_offset,
                                                                      unsigned long long* __restrict__ out_pos) {
  __shared__ uint32_t lkey[POS_MAX_LISTED];
  __shared__ uint32_t lrow[POS_MAX_LISTED];
  __shared__ uint32_t lcnt[POS_MAX_LISTED];
  const uint32_t q = blockIdx.y;
  const float* row = scores + (uint64_t)q * ld;
  for (int i = threadIdx.x; i < m; i += POS_THREADS) {
    const int64_t id = ids[(uint64_t)q * m + i] - row_offset;
    const bool ok = id >= 0 && id < (int64_t)n;
    lrow[i] = ok ? (uint32_t)id : 0xFFFFFFFFu;
    lkey[i] = ok ? ~f2key(row[id]) : 0u;
    lcnt[i] = 0;
  }
  __syncthreads();
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t j0 = blockIdx.x * per, j1 = min(n, j0 + per);
  // every thread owns a strided subset of this chunk's rows and compares each against all listed rows; the per-listed-row
  // counts are reduced over the wave by ballots and added to LDS once per 64 rows
  const int lane = threadIdx.x & 63;
  for (uint32_t jb = j0 + (threadIdx.x & ~63u); jb < j1; jb += POS_THREADS) {
    const uint32_t jj = jb + lane;
    const bool valid = jj < j1;
    const uint32_t kj = valid ? ~f2key(row[jj]) : 0xFFFFFFFFu;
    for (int i = 0; i < m; ++i) {
      const uint32_t kr = lkey[i], rr = lrow[i];
      const bool before = valid && rr != 0xFFFFFFFFu && (kj < kr || (kj == kr && jj < rr));
      const unsigned long long bm = __ballot(before);
      if (lane == 0 && bm) atomicAdd(&lcnt[i], (uint32_t)__popcll(bm));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += POS_THREADS)
    if (lrow[i] != 0xFFFFFFFFu && lcnt[i]) atomicAdd(&out_pos[(uint64_t)q * m + i], (unsigned long long)lcnt[i]);
}

void launch_rank_positions(const float* scores, int64_t ld, int64_t n, int32_t nq, const int64_t* ids, int32_t m,
                           int64_t row_offset, unsigned long long* out_pos, hipStream_t stream) {
  const unsigned chunks = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, n / 4096));
  hipLaunchKernelGGL(rank_positions_kernel, dim3(chunks, nq), dim3(POS_THREADS), 0, stream, scores, ld, (uint32_t)n, ids,
                     m, row_offset, out_pos);
}
int rank_positions_max_listed() { return POS_MAX_LISTED; }

void launch_rank_all(const float* scores, int64_t ld, int64_t n, int32_t nq, uint32_t* keys_a, uint32_t* idx_a,
                     uint32_t* keys_b, uint32_t* idx_b, int64_t row_offset, int64_t* out_idx, float* out_score,
                     hipStream_t stream) {
  hipLaunchKernelGGL(rank_all_kernel, dim3(nq), dim3(RANK_THREADS), 0, stream, scores, ld, (uint32_t)n, keys_a, idx_a,
                     keys_b, idx_b, row_offset, out_idx, out_score);
}

void launch_dense_topk(const float* scores, int64_t ld, int64_t n, int32_t nq, int32_t k, int64_t row_offset,
                       int64_t* out_idx, float* out_score, hipStream_t stream) {
  uint32_t k2 = 2;
  while (k2 < (uint32_t)k) k2 <<= 1;
  const size_t lds = (size_t)k2 * 8 + 256 * 4 + 32;
  hipLaunchKernelGGL(dense_topk_kernel, dim3(nq), dim3(512), lds, stream, scores, ld, (uint32_t)n, k, row_offset,
                     out_idx, out_score);
}

}  // namespace mi
