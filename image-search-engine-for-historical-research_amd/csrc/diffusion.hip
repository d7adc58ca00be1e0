// Truncated graph diffusion on the GPU (small-database re-ranking, N < 120000).
//
// Restates src/utils/diffusion.py (fyang93/diffusion fork):
//   get_affinity  :101-116  sims<0 -> 0, sims**gamma, keep (i, ids[i,j]) iff i is in the kd-list of ids[i,j]
//                           (mutual), position 0 always dropped
//   get_laplacian :87-98    deg = A@1 + 1e-12 (f64), S = D^-1/2 A D^-1/2 (f32), L = I - alpha*S (f32)
//   get_offline_result :15-19 + :67-83   for every node i: L_i = L[ids_i][:, ids_i] (n_trunc x n_trunc), solve
//                           L_i x = e0 by CG (x0 = 0, tol 1e-6 relative to |b| = 1, at most 20 iterations, f64
//                           vectors, f32 matrix entries), scatter into the sparse `offline` matrix
// and the online combination of src/utils/Reranking.py:242-253 (scores = sims**3 @ offline[idx]).
// The reference runs the N solves on joblib threads through scipy; here one persistent workgroup solves one
// node at a time with the CG vectors in LDS and the (<= kd)-sparse Laplacian rows read through L2.
#include "common.h"
#include "kernels.h"

namespace mi {

// ---- affinity + Laplacian entries in ELL form aligned with the kNN lists: entry (i, j) <-> column ids[i][j]
__global__ __launch_bounds__(256) void affinity_kernel(const int64_t* __restrict__ ids, const float* __restrict__ sims,
                                                       int64_t ld, int32_t kd, int32_t gamma,
                                                       float* __restrict__ aff /*[n][kd]*/) {
  const int64_t i = blockIdx.x;
  for (int j = threadIdx.x; j < kd; j += blockDim.x) {
    const int64_t nb = ids[i * ld + j];
    bool mutual = false;
    if (j != 0) {                                     // ismutual[0] = False (src/utils/diffusion.py:108)
      const int64_t* row = ids + nb * ld;
      for (int t = 0; t < kd; ++t) mutual |= (row[t] == i);
    }
    float s = sims[i * ld + j];
    s = s < 0.f ? 0.f : s;
    float v = 1.f;
    for (int e = 0; e < gamma; ++e) v *= s;
    aff[i * kd + j] = mutual ? v : 0.f;
  }
}

__global__ __launch_bounds__(256) void degree_kernel(const float* __restrict__ aff, int64_t n, int32_t kd,
                                                     float* __restrict__ dinv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double deg = 0.0;
  for (int j = 0; j < kd; ++j) deg += (double)aff[i * kd + j];
  deg += 1e-12;
  dinv[i] = (float)(1.0 / sqrt(deg));                 // degrees ** (-0.5), stored in a float32 dia_matrix
}

// lap[i][j] = -(alpha * ((dinv[i] * A[i][j]) * dinv[col]))  in f32 like scipy's f32 products; a self entry is folded
// into diag[i] = 1 - alpha*S[i][i]
__global__ __launch_bounds__(256) void laplacian_kernel(const int64_t* __restrict__ ids, int64_t ld,
                                                        float* __restrict__ aff_to_lap, const float* __restrict__ dinv,
                                                        int32_t kd, float alpha, float* __restrict__ diag) {
  const int64_t i = blockIdx.x;
  __shared__ float dsum;
  if (threadIdx.x == 0) dsum = 0.f;
  __syncthreads();
  for (int j = threadIdx.x; j < kd; j += blockDim.x) {
    const int64_t c = ids[i * ld + j];
    const float a = aff_to_lap[i * kd + j];
    const float s = (dinv[i] * a) * dinv[c];
    const float l = alpha * s;
    if (c == i) {
      atomicAdd(&dsum, l);
      aff_to_lap[i * kd + j] = 0.f;
    } else {
      aff_to_lap[i * kd + j] = -l;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) diag[i] = 1.0f - dsum;
}

// ---- per-node truncated CG.  LDS: ids_i[T] int32 | x, r, p, q double[T] | red[8]
__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

__global__ __launch_bounds__(256) void diffusion_cg_kernel(const int64_t* __restrict__ ids, int64_t ld, int64_t n,
                                                           int32_t T, int32_t kd, const float* __restrict__ lap,
                                                           const float* __restrict__ diag, int32_t maxiter, double tol,
                                                           int32_t* __restrict__ map_all /*[grid][n], all -1*/,
                                                           int32_t* __restrict__ out_ids /*[n][T]*/,
                                                           float* __restrict__ out_vals /*[n][T]*/, int64_t node0,
                                                           int64_t node1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* lids = reinterpret_cast<int32_t*>(smem);
  double* x = reinterpret_cast<double*>(smem + (((size_t)T * 4 + 15) / 16) * 16);
  double* r = x + T;
  double* p = r + T;
  double* qv = p + T;
  double* red = qv + T;
  int32_t* map = map_all + (int64_t)blockIdx.x * n;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int64_t node = node0 + blockIdx.x; node < node1; node += gridDim.x) {   // the solves are independent: any node range
    for (int c = threadIdx.x; c < T; c += blockDim.x) {
      const int32_t gid = (int32_t)ids[node * ld + c];
      lids[c] = gid;
      map[gid] = c;
      x[c] = 0.0;
      r[c] = (c == 0) ? 1.0 : 0.0;                    // b = e0 (trunc_init[0] = 1, src/utils/diffusion.py:70-71)
    }
    __threadfence_block();
    __syncthreads();
    double rho_prev = 0.0;
    for (int it = 0; it < maxiter; ++it) {
      double rr = 0.0;
      for (int c = threadIdx.x; c < T; c += blockDim.x) rr += r[c] * r[c];
      rr = block_sum(rr, red);
      if (sqrt(rr) < tol) break;                      // ||r|| < rtol * ||b||, ||b|| = 1
      const double rho = rr;                          // z = r (no preconditioner)
      if (it > 0) {
        const double beta = rho / rho_prev;
        for (int c = threadIdx.x; c < T; c += blockDim.x) p[c] = p[c] * beta + r[c];
      } else {
        for (int c = threadIdx.x; c < T; c += blockDim.x) p[c] = r[c];
      }
      __syncthreads();
      // q = L_i p : one wave per local row, lanes over the <= kd stored entries of the global row
      for (int row = wv; row < T; row += nw) {
        const int32_t u = lids[row];
        const int64_t* urow = ids + (int64_t)u * ld;
        const float* lrow = lap + (int64_t)u * kd;
        double acc = 0.0;
        for (int j = lane; j < kd; j += 64) {
          const float w = lrow[j];
          if (w != 0.f) {
            const int32_t c = map[urow[j]];
            if (c >= 0) acc += (double)w * p[c];
          }
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) qv[row] = (double)diag[u] * p[row] + acc;
      }
      __syncthreads();
      double pq = 0.0;
      for (int c = threadIdx.x; c < T; c += blockDim.x) pq += p[c] * qv[c];
      pq = block_sum(pq, red);
      const double alpha = rho / pq;
      for (int c = threadIdx.x; c < T; c += blockDim.x) {
        x[c] += alpha * p[c];
        r[c] -= alpha * qv[c];
      }
      rho_prev = rho;
      __syncthreads();
    }
    for (int c = threadIdx.x; c < T; c += blockDim.x) {
      out_ids[node * T + c] = lids[c];
      out_vals[node * T + c] = (float)x[c];
      map[lids[c]] = -1;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- online: scores[q][:] = sum_j sims[q][j]**gamma * offline[idx[q][j]]   (dense row per query, f32)
__global__ __launch_bounds__(256) void diffusion_combine_kernel(const int64_t* __restrict__ nn_idx,
                                                                const float* __restrict__ nn_sims, int32_t kq,
                                                                int32_t gamma, const int32_t* __restrict__ off_ids,
                                                                const float* __restrict__ off_vals, int32_t T,
                                                                int64_t n, float* __restrict__ dense /*[nq][n]*/) {
  const int64_t q = blockIdx.x;
  float* row = dense + q * n;
  for (int64_t c = threadIdx.x; c < n; c += blockDim.x) row[c] = 0.f;
  __threadfence_block();
  __syncthreads();
  for (int j = 0; j < kq; ++j) {                      // ids within one offline row are unique: no conflicts
    const int64_t src = nn_idx[q * kq + j];
    float s = nn_sims[q * kq + j], w = 1.f;
    for (int e = 0; e < gamma; ++e) w *= s;
    for (int c = threadIdx.x; c < T; c += blockDim.x) row[off_ids[src * T + c]] += w * off_vals[src * T + c];
    __threadfence_block();
    __syncthreads();
  }
}

void launch_affinity(const int64_t* ids, const float* sims, int64_t ld, int64_t n, int32_t kd, int32_t gamma,
                     float alpha, float* lap, float* dinv, float* diag, hipStream_t stream) {
  hipLaunchKernelGGL(affinity_kernel, dim3((unsigned)n), dim3(256), 0, stream, ids, sims, ld, kd, gamma, lap);
  hipLaunchKernelGGL(degree_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, lap, n, kd, dinv);
  hipLaunchKernelGGL(laplacian_kernel, dim3((unsigned)n), dim3(256), 0, stream, ids, ld, lap, dinv, kd, alpha, diag);
}

void launch_diffusion_cg(const int64_t* ids, int64_t ld, int64_t n, int32_t T, int32_t kd, const float* lap,
                         const float* diag, int32_t maxiter, double tol, int32_t* map_all, unsigned grid,
                         int32_t* out_ids, float* out_vals, hipStream_t stream, int64_t node0, int64_t node1) {
  if (node1 < 0) node1 = n;
  const size_t lds = (((size_t)T * 4 + 15) / 16) * 16 + (size_t)T * 8 * 4 + 64;
  ensure_dynamic_lds((const void*)diffusion_cg_kernel);
  hipLaunchKernelGGL(diffusion_cg_kernel, dim3(grid), dim3(256), lds, stream, ids, ld, n, T, kd, lap, diag, maxiter,
                     tol, map_all, out_ids, out_vals, node0, node1);
}

void launch_diffusion_combine(const int64_t* nn_idx, const float* nn_sims, int32_t kq, int32_t gamma,
                              const int32_t* off_ids, const float* off_vals, int32_t T, int64_t n, int32_t nq,
                              float* dense, hipStream_t stream) {
  hipLaunchKernelGGL(diffusion_combine_kernel, dim3((unsigned)nq), dim3(256), 0, stream, nn_idx, nn_sims, kq, gamma,
                     off_ids, off_vals, T, n, dense);
}

}  // namespace mi
