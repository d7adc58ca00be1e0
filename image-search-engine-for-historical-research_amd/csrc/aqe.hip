// alpha query expansion (feature_enhancement, src/utils/Reranking.py:195-208 / :288-301):
//   q'[d] = sum_{j<k} ((k-j)/k)^w * G[ranks[j,q]][d] ;  q' /= (||q'||_2 + eps)
// The reference's f64 weight array promotes the sum to f64; here the gathered f32 rows are accumulated
// in f64 as well.  `partial` adds only the rows this shard owns (the sum is all-reduced across shards by
// the host); `finish` normalises.  HBM-bound gather of k rows per query (k = 3 or 10).
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace mi {

// One weight and one accumulation step for every form of the sum (one shard, partial sums, gathered rows): the expanded
// query of a sharded gallery must be the single-GPU one bit for bit, so all kernels share the operation, not just the formula.
__device__ __forceinline__ double aqe_weight(int j, int32_t k_qe, double w, const double* __restrict__ weights) {
  return weights ? weights[j] : pow((double)(k_qe - j) / (double)k_qe, w);
}
__device__ __forceinline__ double aqe_step(double acc, float x, double wt) { return fma((double)x, wt, acc); }

__global__ __launch_bounds__(256) void aqe_partial_kernel(const float* __restrict__ gal, int32_t dp, int32_t d,
                                                          int64_t n, int64_t row_offset,
                                                          const int64_t* __restrict__ ranks, int64_t sj, int64_t sq,
                                                          int32_t k_qe, double w, const double* __restrict__ weights,
                                                          double* __restrict__ out_sum) {
  const int64_t q = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < k_qe; ++j) {
      const int64_t gid = ranks[j * sj + q * sq] - row_offset;
      if (gid < 0 || gid >= n) continue;
      acc = aqe_step(acc, gal[gid * dp + c], aqe_weight(j, k_qe, w, weights));
    }
    out_sum[q * d + c] = acc;
  }
}

// Sharded alpha-QE, round 4: every shard contributes the ROWS it owns of the k_qe x Q requested ones (zeros elsewhere),
// the [k_qe][Q][d] f32 blocks are summed across the shards -- every element has exactly one non-zero contributor, so that sum
// is exact in any order -- and every rank then adds the rows in j order with the single-GPU kernel's own step.
__global__ __launch_bounds__(256) void aqe_rows_kernel(const float* __restrict__ gal, int32_t dp, int32_t d, int64_t n,
                                                       int64_t row_offset, const int64_t* __restrict__ ranks, int64_t sj,
                                                       int64_t sq, int64_t nq, float* __restrict__ out_rows) {
  const int64_t q = blockIdx.x, j = blockIdx.y;
  const int64_t gid = ranks[j * sj + q * sq] - row_offset;
  const bool mine = gid >= 0 && gid < n;
  float* dst = out_rows + (j * nq + q) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = mine ? gal[gid * dp + c] : 0.f;
}

__global__ __launch_bounds__(256) void aqe_combine_kernel(const float* __restrict__ rows, int64_t nq, int32_t d, int32_t k_qe,
                                                          double w, const double* __restrict__ weights,
                                                          double* __restrict__ out_sum) {
  const int64_t q = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < k_qe; ++j) acc = aqe_step(acc, rows[((int64_t)j * nq + q) * d + c], aqe_weight(j, k_qe, w, weights));
    out_sum[q * d + c] = acc;
  }
}

__global__ __launch_bounds__(256) void aqe_finish_kernel(const double* __restrict__ sum, int32_t d, double eps,
                                                         float* __restrict__ out_q, double* __restrict__ out_q64) {
  __shared__ double red[4];
  const int64_t q = blockIdx.x;
  double ss = 0.0;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const double v = sum[q * d + c];
    ss += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
  const double inv = 1.0 / (nrm + eps);
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const double v = sum[q * d + c] * inv;
    out_q[q * d + c] = (float)v;
    if (out_q64) out_q64[q * d + c] = v;
  }
}

void launch_aqe_partial(const float* gal_f32, int32_t dp, int32_t d, int64_t n, int64_t row_offset,
                        const int64_t* ranks, int64_t sj, int64_t sq, int64_t nq, int32_t k_qe, double w,
                        const double* weights, double* out_sum, hipStream_t stream) {
  hipLaunchKernelGGL(aqe_partial_kernel, dim3((unsigned)nq), dim3(256), 0, stream, gal_f32, dp, d, n, row_offset,
                     ranks, sj, sq, k_qe, w, weights, out_sum);
}

void launch_aqe_rows(const float* gal_f32, int32_t dp, int32_t d, int64_t n, int64_t row_offset, const int64_t* ranks,
                     int64_t sj, int64_t sq, int64_t nq, int32_t k_qe, float* out_rows, hipStream_t stream) {
  hipLaunchKernelGGL(aqe_rows_kernel, dim3((unsigned)nq, (unsigned)k_qe), dim3(256), 0, stream, gal_f32, dp, d, n, row_offset,
                     ranks, sj, sq, nq, out_rows);
}

void launch_aqe_combine(const float* rows, int64_t nq, int32_t d, int32_t k_qe, double w, const double* weights,
                        double* out_sum, hipStream_t stream) {
  hipLaunchKernelGGL(aqe_combine_kernel, dim3((unsigned)nq), dim3(256), 0, stream, rows, nq, d, k_qe, w, weights, out_sum);
}

void launch_aqe_finish(const double* sum, int64_t nq, int32_t d, double eps, float* out_q, double* out_q64,
                       hipStream_t stream) {
  hipLaunchKernelGGL(aqe_finish_kernel, dim3((unsigned)nq), dim3(256), 0, stream, sum, d, eps, out_q, out_q64);
}

// column means of a strided [n][d] matrix in float64 (centring step of AQE / DBA, src/utils/Reranking.py:315-318)
template <typename InT>
__global__ __launch_bounds__(256) void column_sum_kernel(const InT* __restrict__ X, int64_t n, int32_t d, int64_t rs,
                                                         int64_t cs, double* __restrict__ out) {
  // block (bx, by): columns bx*64.., row slab by; 4 row-lanes x 64 column-lanes
  __shared__ double red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int64_t slab = (n + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * slab, r1 = min(n, r0 + slab);
  double acc = 0.0;
  if (col < d)
    for (int64_t r = r0 + rl; r < r1; r += 4) acc += (double)X[r * rs + (int64_t)col * cs];
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && col < d) atomicAdd(&out[col], red[0][col & 63] + red[1][col & 63] + red[2][col & 63] + red[3][col & 63]);
}

void launch_column_sum(const void* X, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, double* out,
                       hipStream_t stream) {
  hipMemsetAsync(out, 0, (size_t)d * 8, stream);
  dim3 grid((d + 63) / 64, (unsigned)std::min<int64_t>(256, (n + 255) / 256));
  if (dtype == 0)
    hipLaunchKernelGGL(column_sum_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, n, d, rs, cs, out);
  else
    hipLaunchKernelGGL(column_sum_kernel<double>, grid, dim3(256), 0, stream, (const double*)X, n, d, rs, cs, out);
}

}  // namespace mi
