// whitenapply (src/utils/whiten.py:4-12, identical copy src/layers/whiten.py):
//   Y = P[:dims, :] @ (X - m) ;  Y /= (||Y||_2 over each column + 1e-6)
// X is the reference's [D, N] column-per-image matrix; here every image is a row (strided input), so
//   y_n = P_dims (x_n - m),  y_n /= (||y_n|| + eps).
// The reference computes in float64 (P and m come out of numpy eig/cholesky), so this is an f64 GEMM:
// LDS-tiled 64 x 64 outputs per workgroup, 4 x 4 per thread, v_fma_f64.  Off the per-query hot path (it is
// applied once per gallery, src/main_train.py:711-712).
#include "common.h"
#include "kernels.h"

namespace mi {

template <typename InT>
__global__ __launch_bounds__(256) void whiten_gemm_kernel(const InT* __restrict__ X, int64_t n, int32_t d, int64_t rs,
                                                          int64_t cs, const double* __restrict__ m,
                                                          const double* __restrict__ P /*[dims][d]*/, int32_t dims,
                                                          double* __restrict__ Y /*[n][dims]*/) {
  __shared__ double As[16][65];
  __shared__ double Bs[16][65];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  const int lr = t >> 2, lk = (t & 3) * 4;
  for (int k0 = 0; k0 < d; k0 += 16) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + lk + e;
      const int64_t r = row0 + lr;
      As[lk + e][lr] = (r < n && k < d) ? (double)X[r * rs + (int64_t)k * cs] - m[k] : 0.0;
      const int j = j0 + lr;
      Bs[lk + e][lr] = (j < dims && k < d) ? P[(int64_t)j * d + k] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + ty * 4 + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = j0 + tx * 4 + j;
      if (c < dims) Y[r * dims + c] = acc[i][j];
    }
  }
}

__global__ __launch_bounds__(256) void rownorm_f64_kernel(double* __restrict__ Y, int32_t dims, double eps) {
  __shared__ double red[4];
  const int64_t r = blockIdx.x;
  double ss = 0.0;
  for (int c = threadIdx.x; c < dims; c += blockDim.x) {
    const double v = Y[r * dims + c];
    ss += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x; c < dims; c += blockDim.x) Y[r * dims + c] = Y[r * dims + c] / (nrm + eps);
}

void launch_whiten(const void* X, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, const double* m,
                   const double* P, int32_t dims, double eps, double* Y, hipStream_t stream) {
  dim3 grid((unsigned)((n + 63) / 64), (unsigned)((dims + 63) / 64));
  if (dtype == 0)
    hipLaunchKernelGGL(whiten_gemm_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, n, d, rs, cs, m, P, dims, Y);
  else
    hipLaunchKernelGGL(whiten_gemm_kernel<double>, grid, dim3(256), 0, stream, (const double*)X, n, d, rs, cs, m, P, dims, Y);
  if (eps >= 0.0) hipLaunchKernelGGL(rownorm_f64_kernel, dim3((unsigned)n), dim3(256), 0, stream, Y, dims, eps);
}

}  // namespace mi
