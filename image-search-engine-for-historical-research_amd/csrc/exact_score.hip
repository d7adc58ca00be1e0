// f32 FMA scoring with the same survivor filter as the MFMA kernel.  Used (a) when `force_exact` is set,
// (b) as the fallback when the bf16 pass overflowed its buffers.  Scores are k-ordered f32 fmaf chains
// over the stored f32 rows, so the error margin is only the f32 accumulation bound.  Not a fast path:
// plain LDS-tiled VALU kernel (64 gallery rows x 64 queries per workgroup, 4x4 outputs per thread).
#include "common.h"
#include "kernels.h"

namespace mi {

// MODE 0: threshold filter, 1: bootstrap chunk (packed store-all), 2: dense -- every score to dense_out[q * ld + row]
template <int MODE>
__global__ __launch_bounds__(256) void exact_select_kernel(ExactArgs p) {
  constexpr bool FIRST = (MODE == 1);
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int t = threadIdx.x;
  const int tx = t & 15, ty = t >> 4;
  const int64_t row0 = p.row0 + (int64_t)blockIdx.x * 64;
  const int q0 = blockIdx.y * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int lr = t >> 2, lk = (t & 3) * 4;
  const int64_t grow = row0 + lr;
  const bool grow_ok = grow < p.row1;
  for (int k0 = 0; k0 < p.dp; k0 += 16) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grow_ok) a = *reinterpret_cast<const float4*>(p.gal_f32 + grow * p.dp + k0 + lk);
    const float4 b = *reinterpret_cast<const float4*>(p.qry_f32 + (int64_t)(q0 + lr) * p.dp + k0 + lk);
    __syncthreads();
    As[lk + 0][lr] = a.x; As[lk + 1][lr] = a.y; As[lk + 2][lr] = a.z; As[lk + 3][lr] = a.w;
    Bs[lk + 0][lr] = b.x; Bs[lk + 1][lr] = b.y; Bs[lk + 2][lr] = b.z; Bs[lk + 3][lr] = b.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float ar[4] = {av.x, av.y, av.z, av.w};
      const float br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], br[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t q = (uint32_t)(q0 + tx * 4 + j);
    if (q >= (uint32_t)p.nq) continue;
    const float thr = (MODE != 0) ? -INFINITY : p.st.thr[q];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = row0 + ty * 4 + i;
      if (row >= p.row1) continue;
      const float v = acc[i][j];
      if (MODE == 2) {
        p.dense_out[(uint64_t)q * p.dense_ld + (uint64_t)row] = v;
      } else if (FIRST) {
        p.st.surv[(uint64_t)q * p.st.cap + (uint64_t)row] = pack_entry(v, (uint32_t)row);
      } else if (v >= thr) {
        const uint32_t pos = atomicAdd(&p.st.cnt[q * CNT_STRIDE], 1u);
        if (pos < p.st.cap) p.st.surv[(uint64_t)q * p.st.cap + pos] = pack_entry(v, (uint32_t)row);
        else atomicOr(p.st.flags, FLAG_SURV_OVERFLOW);
      }
    }
  }
}

void launch_exact_select(const ExactArgs& a, bool first, hipStream_t stream) {
  const int64_t rows = a.row1 - a.row0;
  if (rows <= 0) return;
  const int qpad64 = (int)round_up(a.nq, 64);
  dim3 grid((unsigned)((rows + 63) / 64), (unsigned)(qpad64 / 64));
  if (a.dense_out) hipLaunchKernelGGL(exact_select_kernel<2>, grid, dim3(256), 0, stream, a);
  else if (first) hipLaunchKernelGGL(exact_select_kernel<1>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(exact_select_kernel<0>, grid, dim3(256), 0, stream, a);
}

}  // namespace mi
