// k-reciprocal re-ranking on the GPU: src/utils/Reranking.py:447-624 (kr_reranking; Zhong et al.,
// "Re-ranking Person Re-identification with k-reciprocal Encoding", CVPR 2017), constants k1 = 20, k2 = 6, lambda = 0.3.
//
// The reference builds dense all x all arrays on the host (V float32, V_qe float16, :505, :586) and walks them with Python
// loops over every image; all = queries + gallery (5063 for rOxford5k).  Here the all x all inner products come from the
// exact f32 scorer (exact_score.hip, dense mode), the k1 + 1 nearest of every row from dense_topk_kernel, and this file
// holds the set logic and the encodings:
//   kr_sets_kernel     R[i]: k-reciprocal set of i, expanded by the k1/2-reciprocal sets of its members that share more
//                      than 2/3 of their elements with it (:565-575); sorted, unique (np.unique)
//   kr_weights_kernel  V[i, R[i]] = softmax(-dist(i, .) / max_j dist(i, j)), dist = 2 - 2 a.b (:514-525)
//   kr_expand_kernel   V_qe[i] = mean of the rows of the k2 nearest, rounded to float16 (:585-589), stored twice
//                      (row-major and transposed: the Jaccard pass reads columns)
//   kr_final_kernel    jaccard[i, j] = 1 - m / (2 - m), m = sum_c min(V_qe[i, c], V_qe[j, c]) over the non-zero columns c
//                      of query i in ascending order, float32 accumulation (:602-609); final = (1 - lambda) jaccard +
//                      lambda dist / colmax (:611-613) for the gallery columns
// The float operations follow the reference's order (no fused multiply-add where numpy / torch round twice), so the
// result differs from it only where torch's float32 matmul and its exp round differently from the k-ordered fmaf chain
// and expf used here: near-ties of final_dist.
#include <hip/hip_fp16.h>

#include "common.h"
#include "kernels.h"

namespace mi {

constexpr int KR_RMAX = 256;      // |R[i]| <= (k1 + 1) + (k1 + 1) * (k1 / 2 + 1) = 252 at k1 = 20
constexpr int KR_NZMAX = 2048;    // non-zero columns of one V_qe row <= k2 * |R| = 1512 at the reference's constants

// wave-cooperative k-reciprocal neighbours of `node` (src/utils/Reranking.py:527-531): out = { fwd[t] : node in
// rank[fwd[t], :k + 1] }, fwd = rank[node, :k + 1], in the order of t.  One wave; k + 1 <= 64.
__device__ __forceinline__ int kr_neigh(const int64_t* __restrict__ rank, int ld, int node, int k, int lane, int32_t* out) {
  int fwd = -1;
  bool member = false;
  if (lane <= k) {
    fwd = (int)rank[(int64_t)node * ld + lane];
    for (int u = 0; u <= k; ++u) member |= (rank[(int64_t)fwd * ld + u] == node);
  }
  const unsigned long long m = __ballot(member);
  if (member) out[__popcll(m & ((1ull << lane) - 1ull))] = fwd;
  return __popcll(m);
}

__global__ __launch_bounds__(64) void kr_sets_kernel(const int64_t* __restrict__ rank, int ld, int k1, int khalf,
                                                     int32_t* __restrict__ R, int32_t* __restrict__ Rcnt,
                                                     uint32_t* __restrict__ flags) {
  __shared__ int32_t base[64], cs[64], list[KR_RMAX + 64];
  __shared__ uint8_t first[KR_RMAX + 64];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int nbase = kr_neigh(rank, ld, i, k1, lane, base);
  __syncthreads();
  if (lane < nbase) list[lane] = base[lane];
  int nlist = nbase;
  for (int b = 0; b < nbase; ++b) {
    const int ncs = kr_neigh(rank, ld, base[b], khalf, lane, cs);
    __syncthreads();
    bool common = false;
    if (lane < ncs)
      for (int u = 0; u < nbase; ++u) common |= (base[u] == cs[lane]);
    const int inter = __popcll(__ballot(common));
    if (inter * 3 > 2 * ncs) {                     // len(intersect1d) > 2/3 * len(candidate set)
      if (nlist + ncs <= KR_RMAX + 64) {
        if (lane < ncs) list[nlist + lane] = cs[lane];
        nlist += ncs;
      } else if (lane == 0) atomicOr(flags, 1u);
    }
    __syncthreads();
  }
  // np.unique: first occurrences, ascending
  for (int p = lane; p < nlist; p += 64) {
    bool f = true;
    for (int q = 0; q < p; ++q) f &= (list[q] != list[p]);
    first[p] = f;
  }
  __syncthreads();
  int nuniq = 0;
  for (int p = 0; p < nlist; ++p) nuniq += first[p];
  if (nuniq > KR_RMAX) {
    if (lane == 0) atomicOr(flags, 1u);
    nuniq = KR_RMAX;
  }
  for (int p = lane; p < nlist; p += 64)
    if (first[p]) {
      int pos = 0;
      for (int q = 0; q < nlist; ++q) pos += (first[q] && list[q] < list[p]);
      if (pos < KR_RMAX) R[(int64_t)i * KR_RMAX + pos] = list[p];
    }
  if (lane == 0) Rcnt[i] = nuniq;
}

// dmax[i] = max_j (2 - 2 S[i, j]);  V[i, t] = exp(-d_t) / sum exp(-d), d_t = (2 - 2 S[i, R[i][t]]) / dmax[i]
__global__ __launch_bounds__(256) void kr_weights_kernel(const float* __restrict__ S, int all, const int32_t* __restrict__ R,
                                                         const int32_t* __restrict__ Rcnt, float* __restrict__ V,
                                                         float* __restrict__ dmax_out) {
  __shared__ float red[256];
  __shared__ float w[KR_RMAX];
  const int i = blockIdx.x, t = threadIdx.x;
  const float* row = S + (int64_t)i * all;
  float m = -INFINITY;
  for (int j = t; j < all; j += 256) m = fmaxf(m, __fsub_rn(2.0f, __fmul_rn(2.0f, row[j])));
  red[t] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] = fmaxf(red[t], red[t + s]);
    __syncthreads();
  }
  const float dmax = red[0];
  __syncthreads();
  const int n = Rcnt[i];
  float e = 0.f;
  if (t < n) {
    const float d = __fdiv_rn(__fsub_rn(2.0f, __fmul_rn(2.0f, row[R[(int64_t)i * KR_RMAX + t]])), dmax);
    e = expf(-d);
    w[t] = e;
  }
  red[t] = e;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] = __fadd_rn(red[t], red[t + s]);
    __syncthreads();
  }
  if (t < n) V[(int64_t)i * KR_RMAX + t] = __fdiv_rn(w[t], red[0]);
  if (t == 0) dmax_out[i] = dmax;
}

// V_qe[i, :] = float16(mean of V[rank[i, 0..k2), :]): the k2 sparse rows are added in the reference's order, column by
// column in float32 (np.mean over axis 0 of a [k2, all] float32 array), divided by k2, rounded to float16
__global__ __launch_bounds__(256) void kr_expand_kernel(const int64_t* __restrict__ rank, int ld, int k2, int all,
                                                        const int32_t* __restrict__ R, const int32_t* __restrict__ Rcnt,
                                                        const float* __restrict__ V, __half* __restrict__ Vqe,
                                                        __half* __restrict__ VqeT) {
  extern __shared__ float acc[];                   // [all]
  const int i = blockIdx.x, t = threadIdx.x;
  for (int c = t; c < all; c += 256) acc[c] = 0.f;
  __syncthreads();
  for (int nb = 0; nb < k2; ++nb) {
    const int j = (int)rank[(int64_t)i * ld + nb];
    const int n = Rcnt[j];
    if (t < n) {
      const int c = R[(int64_t)j * KR_RMAX + t];
      acc[c] = __fadd_rn(acc[c], V[(int64_t)j * KR_RMAX + t]);       // the columns of one row are distinct
    }
    __syncthreads();
  }
  const float k2f = (float)k2;
  for (int c = t; c < all; c += 256) {
    const __half h = __float2half_rn(k2 != 1 ? __fdiv_rn(acc[c], k2f) : acc[c]);
    Vqe[(int64_t)i * all + c] = h;
    VqeT[(int64_t)c * all + i] = h;
  }
}

// one workgroup per (256 columns j, query i): the non-zero columns of V_qe[i] in ascending order (LDS), then per j the
// float32 sum of min(V_qe[i, c], V_qe[j, c]), the Jaccard distance, the normalised original distance and the mix
__global__ __launch_bounds__(256) void kr_final_kernel(const __half* __restrict__ Vqe, const __half* __restrict__ VqeT,
                                                       const float* __restrict__ S, const float* __restrict__ dmax, int all,
                                                       int nq, float w_jac, float w_org, float* __restrict__ neg_final,
                                                       uint32_t* __restrict__ flags) {
  __shared__ uint32_t nz_col[KR_NZMAX];
  __shared__ __half nz_val[KR_NZMAX];
  __shared__ int wave_cnt[4];
  __shared__ int total;
  const int i = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const __half* rowi = Vqe + (int64_t)i * all;
  if (t == 0) total = 0;
  __syncthreads();
  for (int c0 = 0; c0 < all; c0 += 256) {
    const int c = c0 + t;
    const __half v = c < all ? rowi[c] : __float2half(0.f);
    const bool nzf = c < all && __half2float(v) != 0.f;
    const unsigned long long m = __ballot(nzf);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = total;
    for (int u = 0; u < wv; ++u) off += wave_cnt[u];
    const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
    if (nzf) {
      if (pos < KR_NZMAX) {
        nz_col[pos] = (uint32_t)c;
        nz_val[pos] = v;
      } else atomicOr(flags, 2u);
    }
    __syncthreads();
    if (t == 0) total += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  const int nnz = min(total, KR_NZMAX);
  const int j = blockIdx.x * 256 + t;
  if (j >= all || j < nq) return;                  // only the gallery columns are kept (final_dist[:Q, Q:], :617)
  float tm = 0.f;
  for (int e = 0; e < nnz; ++e) {
    const __half vj = VqeT[(int64_t)nz_col[e] * all + j];
    if (__half2float(vj) != 0.f) tm = __fadd_rn(tm, __half2float(__hlt(nz_val[e], vj) ? nz_val[e] : vj));
  }
  const float jac = __fsub_rn(1.0f, __fdiv_rn(tm, __fsub_rn(2.0f, tm)));
  const float org = __fdiv_rn(__fsub_rn(2.0f, __fmul_rn(2.0f, S[(int64_t)j * all + i])), dmax[i]);
  const float fin = __fadd_rn(__fmul_rn(jac, w_jac), __fmul_rn(org, w_org));
  neg_final[(int64_t)i * (all - nq) + (j - nq)] = -fin;
}

// strided f32 | f64 rows -> float32 rows [n][dp], columns d..dp zero (torch.tensor(qvecs.T, dtype=torch.float32), :550-551)
template <typename T>
__global__ __launch_bounds__(256) void kr_pack_kernel(const T* __restrict__ src, int64_t n, int32_t d, int64_t rs, int64_t cs,
                                                      float* __restrict__ out, int32_t dp) {
  const int64_t r = blockIdx.x;
  for (int c = threadIdx.x; c < dp; c += 256) out[r * dp + c] = c < d ? (float)src[r * rs + c * cs] : 0.f;
}
void launch_kr_pack(const void* src, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, float* out, int32_t dp,
                    hipStream_t stream) {
  if (n <= 0) return;
  if (dtype == 0) hipLaunchKernelGGL(kr_pack_kernel<float>, dim3((unsigned)n), dim3(256), 0, stream, (const float*)src, n, d, rs, cs, out, dp);
  else hipLaunchKernelGGL(kr_pack_kernel<double>, dim3((unsigned)n), dim3(256), 0, stream, (const double*)src, n, d, rs, cs, out, dp);
}

void launch_kr_sets(const int64_t* rank, int ld, int all, int k1, int32_t* R, int32_t* Rcnt, uint32_t* flags,
                    hipStream_t stream) {
  const int khalf = (int)nearbyint(k1 / 2.0);      // int(np.around(k1 / 2)): round half to even
  hipLaunchKernelGGL(kr_sets_kernel, dim3(all), dim3(64), 0, stream, rank, ld, k1, khalf, R, Rcnt, flags);
}
void launch_kr_weights(const float* S, int all, const int32_t* R, const int32_t* Rcnt, float* V, float* dmax,
                       hipStream_t stream) {
  hipLaunchKernelGGL(kr_weights_kernel, dim3(all), dim3(256), 0, stream, S, all, R, Rcnt, V, dmax);
}
void launch_kr_expand(const int64_t* rank, int ld, int k2, int all, const int32_t* R, const int32_t* Rcnt, const float* V,
                      void* Vqe, void* VqeT, hipStream_t stream) {
  ensure_dynamic_lds((const void*)kr_expand_kernel);
  hipLaunchKernelGGL(kr_expand_kernel, dim3(all), dim3(256), (size_t)all * 4, stream, rank, ld, k2, all, R, Rcnt, V,
                     (__half*)Vqe, (__half*)VqeT);
}
void launch_kr_final(const void* Vqe, const void* VqeT, const float* S, const float* dmax, int all, int nq, float w_jac,
                     float w_org, float* neg_final, uint32_t* flags, hipStream_t stream) {
  hipLaunchKernelGGL(kr_final_kernel, dim3((all + 255) / 256, nq), dim3(256), 0, stream, (const __half*)Vqe,
                     (const __half*)VqeT, S, dmax, all, nq, w_jac, w_org, neg_final, flags);
}
int kr_rmax() { return KR_RMAX; }

}  // namespace mi
