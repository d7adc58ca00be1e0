// Exact top-k of dense f32 score rows: replaces `np.argsort(-scores)` / faiss IndexFlatIP.search when k is a
// large fraction of N (kNN graph of the diffusion re-ranking: k = 2000 of N ~ 5-11k, src/utils/diffusion.py:66;
// top-2000 of the diffused scores, src/utils/Reranking.py:250-253), where threshold filtering cannot help.
// One workgroup per row: 4-pass MSB radix select of the k-th largest key straight from global memory (the row
// stays in L2), a second select on the INDEX among the ties of that key (ties go to the lower index,
// deterministically), then the k chosen entries are bitonic-sorted in LDS by (score desc, idx asc).
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

void launch_dense_topk(const float* scores, int64_t ld, int64_t n, int32_t nq, int32_t k, int64_t row_offset,
                       int64_t* out_idx, float* out_score, hipStream_t stream) {
  uint32_t k2 = 2;
  while (k2 < (uint32_t)k) k2 <<= 1;
  const size_t lds = (size_t)k2 * 8 + 256 * 4 + 32;
  hipLaunchKernelGGL(dense_topk_kernel, dim3(nq), dim3(512), lds, stream, scores, ld, (uint32_t)n, k, row_offset,
                     out_idx, out_score);
}

}  // namespace mi
