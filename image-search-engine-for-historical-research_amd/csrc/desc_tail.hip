// Descriptor tail of the SOLAR / GeM extractor on the GPU, so that descriptors go from the CNN's
// last feature map straight into the device gallery without the per-image, per-scale `.cpu()` round trips of
// src/networks/imageretrievalnet.py:462,473:
//   GeM pooling   src/layers/functional.py:20-22    avg_pool(clamp(x, eps)^p)^(1/p) over H x W
//   L2N           src/layers/functional.py:129-130  x / (||x|| + 1e-6)
//   whiten        src/networks/imageretrievalnet.py:186-187   Linear(C, C_out, bias) then L2N
//   multi-scale   src/networks/imageretrievalnet.py:464-479   mean over scales of v^msp, ^(1/msp), / ||v||
// All HBM-bound (one read of the feature map, one read of the whitening matrix per call).
#include "common.h"
#include "kernels.h"

namespace mi {

// one wave per (image, channel): out[b][c] = (mean_hw clamp(x, eps)^p)^(1/p)
__global__ __launch_bounds__(256) void gem_pool_kernel(const float* __restrict__ feat, int64_t rows, int32_t hw, float p,
                                                       float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* x = feat + r * hw;
  float acc = 0.f;
  for (int i = lane; i < hw; i += 64) acc += powf(fmaxf(x[i], eps), p);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[r] = powf(acc / (float)hw, 1.0f / p);
}

// x[b][:] /= (||x[b]|| + eps), f32 like the reference's torch ops; one workgroup per row
__global__ __launch_bounds__(256) void l2n_rows_kernel(float* __restrict__ x, int32_t d, float eps) {
  __shared__ float red[4];
  float* row = x + (int64_t)blockIdx.x * d;
  float ss = 0.f;
  for (int c = threadIdx.x; c < d; c += 256) ss += row[c] * row[c];
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x; c < d; c += 256) row[c] = row[c] / (nrm + eps);
}

// y[b][j] = bias[j] + sum_c W[j][c] * x[b][c]   for up to 8 rows b at a time (x staged in LDS); one wave per j
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, int32_t nb, int32_t c,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          int32_t c_out, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);                 // [nb][c]
  for (int i = threadIdx.x; i < nb * c; i += 256) xs[i] = x[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= c_out) return;
  const float* w = W + (int64_t)j * c;
  float acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = 0.f;
  for (int k = lane; k < c; k += 64) {
    const float wv = w[k];
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < nb) acc[b] = fmaf(wv, xs[b * c + k], acc[b]);
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    float v = acc[b];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0 && b < nb) y[(int64_t)b * c_out + j] = v + (bias ? bias[j] : 0.f);
  }
}

__global__ __launch_bounds__(256) void ms_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ desc,
                                                            int64_t count, float msp, int first) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float v = (msp == 1.0f) ? desc[i] : powf(desc[i], msp);
  acc[i] = first ? v : acc[i] + v;
}

// v = (acc / nscales)^(1/msp); v /= ||v||   (no eps, src/networks/imageretrievalnet.py:475-477)
__global__ __launch_bounds__(256) void ms_finish_kernel(float* __restrict__ acc, int32_t d, int32_t nscales, float msp) {
  __shared__ float red[4];
  float* row = acc + (int64_t)blockIdx.x * d;
  float ss = 0.f;
  for (int c = threadIdx.x; c < d; c += 256) {
    float v = row[c] / (float)nscales;
    if (msp != 1.0f) v = powf(v, 1.0f / msp);
    row[c] = v;
    ss += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x; c < d; c += 256) row[c] = row[c] / nrm;
}

void launch_desc_tail(const float* feat, int32_t b, int32_t c, int32_t hw, float p, float eps, const float* W,
                      const float* bias, int32_t c_out, float* pooled /*[b][c] scratch*/, float* out, hipStream_t stream) {
  const int64_t rows = (int64_t)b * c;
  float* pool_dst = W ? pooled : out;
  hipLaunchKernelGGL(gem_pool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, feat, rows, hw, p, eps,
                     pool_dst);
  hipLaunchKernelGGL(l2n_rows_kernel, dim3(b), dim3(256), 0, stream, pool_dst, c, 1e-6f);
  if (W) {
    for (int b0 = 0; b0 < b; b0 += 8) {
      const int nb = b - b0 < 8 ? b - b0 : 8;
      hipLaunchKernelGGL(linear_rows_kernel, dim3((c_out + 3) / 4), dim3(256), (size_t)nb * c * 4, stream,
                         pooled + (int64_t)b0 * c, nb, c, W, bias, c_out, out + (int64_t)b0 * c_out);
    }
    hipLaunchKernelGGL(l2n_rows_kernel, dim3(b), dim3(256), 0, stream, out, c_out, 1e-6f);
  }
}

void launch_ms_accumulate(float* acc, const float* desc, int64_t count, float msp, int first, hipStream_t stream) {
  hipLaunchKernelGGL(ms_accumulate_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, acc, desc, count,
                     msp, first);
}

void launch_ms_finish(float* acc, int32_t b, int32_t d, int32_t nscales, float msp, hipStream_t stream) {
  hipLaunchKernelGGL(ms_finish_kernel, dim3(b), dim3(256), 0, stream, acc, d, nscales, msp);
}

}  // namespace mi
