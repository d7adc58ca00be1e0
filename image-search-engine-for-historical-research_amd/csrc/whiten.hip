// whitenapply (src/utils/whiten.py:4-12, identical copy src/layers/whiten.py):
//   Y = P[:dims, :] @ (X - m) ;  Y /= (||Y||_2 over each column + 1e-6)
// X is the reference's [D, N] column-per-image matrix; here every image is a row (strided input), so
//   y_n = P_dims (x_n - m),  y_n /= (||y_n|| + eps).
// The reference computes in float64 (P and m come out of numpy eig / cholesky; src/main_train.py:711-712 keeps float64 all the
// way into the scores), so this is an f64 GEMM of 2 * dims * d flop per image -- 8.8 TFLOP per million 2048-d descriptors --
// and it runs on the f64 matrix pipe:
//  * v_mfma_f64_16x16x4_f64, 128 x 128 outputs per 256-thread workgroup = 4 waves x (64 x 64) = 4 x 4 MFMA blocks per wave
//    (128 accumulator registers); two workgroups per CU, so one of them stages while the other multiplies;
//  * K streams in chunks of 16: the X chunk is converted to f64 and centred ((double)x - m[k]) ON LOAD, the P chunk is
//    copied, both into LDS rows of 18 doubles (18 r mod 32 is a different even 8-byte slot for each of the 16 rows of a
//    ds_read_b64 fragment read, + k: conflict-free); the global loads of chunk c + 1 are in flight under the 64 MFMAs of chunk c;
//  * the K order inside a 16x16x4 step is the hardware's; across steps it is ascending, like a row-major dot product;
//  * XCD-aware tile order when dims is a multiple of 1024: XCD x owns the column blocks x, x + 8, .. (its 2 MB slab of P
//    stays in its L2) and all XCDs walk the row tiles in the same order (an X tile leaves HBM once and is served to the
//    other seven from the Infinity Cache);
//  * the normalisation reads every row ONCE into registers (dims <= 4096), reduces and writes it back scaled -- or the
//    rows go straight into the gallery ingest (mi_gallery_append_whitened_device), whose MI_NORM_L2_EPS mode is this very
//    tail, so that the [N, dims] float64 matrix never exists beyond a chunk.
// The round-1 kernel (16 x 16 LDS tiles, v_fma_f64, 4 x 4 outputs per thread) is gone: bench.py `whiten` has both figures.
#include "common.h"
#include "kernels.h"

namespace mi {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int W_TILE = 128;          // rows and columns of Y per workgroup
constexpr int W_KC = 16;             // K-chunk staged in LDS
constexpr int W_LD = W_KC + 2;       // LDS row stride in doubles: 18 r mod 32 is a different even slot for each of 16 rows
constexpr int W_PER = W_TILE * W_KC / 256;   // elements of one operand chunk per thread (8)

// ROWS: the K index is the contiguous one of X (cs == 1): a wave reads 4 rows x 16 consecutive k per instruction; otherwise
// (the reference's [D, N] layout, rs == 1, and any other strides) 64 consecutive rows of one k.
template <typename InT, bool ROWS>
__global__ __launch_bounds__(256, 2) void whiten_mfma_kernel(const InT* __restrict__ X, int64_t n, int32_t d, int64_t rs,
                                                             int64_t cs, const double* __restrict__ m,
                                                             const double* __restrict__ P /*[dims][d]*/, int32_t dims,
                                                             double* __restrict__ Y /*[n][ldy]*/, int64_t ldy, uint32_t ncb,
                                                             uint32_t nrb) {
  // two buffers per operand (73.7 KB of dynamic LDS per workgroup, two workgroups per CU): chunk c + 1 is written into the other
  // buffer while chunk c is multiplied, ONE barrier per chunk (the first version had one buffer and two barriers per 64 MFMAs:
  // matrix pipe busy 0.69, profiles/r06z_whiten_pmc.txt)
  extern __shared__ __attribute__((aligned(16))) double w_lds[];
  double* const Xs0 = w_lds;
  double* const Ps0 = w_lds + 2 * W_TILE * W_LD;
  const uint32_t b = blockIdx.x;
  uint32_t cb, rt;
  if ((ncb & 7u) == 0) {
    const uint32_t x = b & 7u, j = b >> 3;
    cb = x + 8u * (j / nrb);
    rt = j % nrb;
  } else {
    cb = b % ncb;
    rt = b / ncb;
  }
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1, l15 = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)rt * W_TILE;
  const int32_t col0 = (int32_t)cb * W_TILE;
  // element i of this thread in an operand chunk: X (r, kk) = (xr0 + xdr * (i & 1 | rows: i), xk0 + xdk * (i >> 1)),
  // P (pj0 + 16 i, pk).  ROWS: 16 consecutive k of 16 rows per instruction and thread-constant k (one mean per thread and
  // chunk); otherwise 64 consecutive rows of one k per wave and instruction, k = wave + 4 j: four means per thread and chunk
  constexpr int NM = ROWS ? 1 : W_PER / 2;
  const int xr0 = ROWS ? t / W_KC : t % 64, xk0 = ROWS ? t % W_KC : t / 64;
  auto x_r = [&](int i) { return ROWS ? xr0 + (256 / W_KC) * i : xr0 + 64 * (i & 1); };
  auto x_k = [&](int i) { return ROWS ? xk0 : xk0 + 4 * (i >> 1); };
  const int pj0 = t / W_KC, pk = t % W_KC;
  const bool interior = row0 + W_TILE <= n && col0 + W_TILE <= dims;
  const InT* xp0 = X + (row0 + xr0) * rs + (int64_t)xk0 * cs;
  const double* pp0 = P + (int64_t)(col0 + pj0) * d + pk;

  InT xr[W_PER];
  double pr[W_PER], mr[NM];
  bool fast = false;                                        // the chunk in the registers lies inside X and P: no masking
  auto load_chunk = [&](int32_t k0) {
    fast = interior && k0 + W_KC <= d;
    // the means of this thread's k: fetched with the chunk (a load at the point of use stalls the store phase on its round
    // trip, once per element -- the first version of this kernel did), clamped index, no branch
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const int32_t k = k0 + xk0 + 4 * j;
      mr[j] = m[k < d ? k : d - 1];
    }
    if (fast) {
      const InT* xp = xp0 + (int64_t)k0 * cs;
      const double* pp = pp0 + k0;
#pragma unroll
      for (int i = 0; i < W_PER; ++i) {
        xr[i] = ROWS ? xp[(int64_t)(256 / W_KC) * i * rs] : xp[(int64_t)(64 * (i & 1)) * rs + (int64_t)(4 * (i >> 1)) * cs];
        pr[i] = pp[(int64_t)(256 / W_KC) * i * d];
      }
    } else {
#pragma unroll
      for (int i = 0; i < W_PER; ++i) {
        const int64_t row = row0 + x_r(i);
        const int32_t k = k0 + x_k(i);
        xr[i] = (row < n && k < d) ? X[row * rs + (int64_t)k * cs] : (InT)0;
        const int32_t pj = col0 + pj0 + (256 / W_KC) * i;
        pr[i] = (pj < dims && k0 + pk < d) ? P[(int64_t)pj * d + k0 + pk] : 0.0;
      }
    }
  };
  auto store_chunk = [&](int32_t k0, int buf) {
    double* const Xs = Xs0 + buf * W_TILE * W_LD;
    double* const Ps = Ps0 + buf * W_TILE * W_LD;
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int r = x_r(i), kk = x_k(i);
      // centred in float64 exactly like `X - m` promotes in the reference; padded rows / k: zero
      double v = (double)xr[i] - mr[ROWS ? 0 : i >> 1];
      if (!fast) v = (row0 + r < n && k0 + kk < d) ? v : 0.0;
      Xs[r * W_LD + kk] = v;
      Ps[(pj0 + (256 / W_KC) * i) * W_LD + pk] = pr[i];
    }
  };

  f64x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f64x4){0.0, 0.0, 0.0, 0.0};

  const int xa_off = (wr * 64 + l15) * W_LD + lq;          // A[i = lane & 15][k = lane >> 4]
  const int pb_off = (wc * 64 + l15) * W_LD + lq;          // B[k = lane >> 4][j = lane & 15] = P[j][k]
  auto mfma_steps = [&](int buf, int ks0, int ks1) {
    const double* xa = Xs0 + buf * W_TILE * W_LD + xa_off;
    const double* pb = Ps0 + buf * W_TILE * W_LD + pb_off;
#pragma unroll
    for (int ks = ks0; ks < ks1; ++ks) {
      double a[4], bb[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = xa[mi * 16 * W_LD + ks * 4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bb[ni] = pb[ni * 16 * W_LD + ks * 4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
    }
  };
  load_chunk(0);
  store_chunk(0, 0);
  __syncthreads();
  int buf = 0;
  for (int32_t k0 = 0; k0 < d; k0 += W_KC, buf ^= 1) {
    const bool more = k0 + W_KC < d;
    if (more) load_chunk(k0 + W_KC);                       // in flight under the first three quarters of this chunk's MFMAs
    mfma_steps(buf, 0, 3);
    if (more) store_chunk(k0 + W_KC, buf ^ 1);             // the other buffer: nobody reads it before the barrier below
    mfma_steps(buf, 3, W_KC / 4);
    __syncthreads();                                       // chunk c + 1 is complete, chunk c's fragments are done with
  }
  // C layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * register
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + wr * 64 + mi * 16 + lq + 4 * r;
      if (row >= n) continue;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int32_t c = col0 + wc * 64 + ni * 16 + l15;
        if (c < dims) Y[row * ldy + c] = acc[mi][ni][r];
      }
    }
}

// y <- y / (||y|| + eps), every row read once (PT doubles per thread in registers) and written once
template <int PT>
__global__ __launch_bounds__(256) void rownorm_f64_kernel(double* __restrict__ Y, int64_t ldy, int32_t dims, double eps) {
  __shared__ double red[4];
  double* y = Y + (int64_t)blockIdx.x * ldy;
  double v[PT];
  double ss = 0.0;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int c = threadIdx.x + 256 * i;
    v[i] = c < dims ? y[c] : 0.0;
    ss += v[i] * v[i];
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const double nrm = sqrt((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < dims) y[c] = v[i] / (nrm + eps);
  }
}

// wider rows than 4096: two passes over the row
__global__ __launch_bounds__(256) void rownorm_f64_wide_kernel(double* __restrict__ Y, int64_t ldy, int32_t dims, double eps) {
  __shared__ double red[4];
  double* y = Y + (int64_t)blockIdx.x * ldy;
  double ss = 0.0;
  for (int c = threadIdx.x; c < dims; c += 256) ss += y[c] * y[c];
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const double nrm = sqrt((red[0] + red[1]) + (red[2] + red[3]));
  for (int c = threadIdx.x; c < dims; c += 256) y[c] = y[c] / (nrm + eps);
}

void launch_whiten(const void* X, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, const double* m,
                   const double* P, int32_t dims, double eps, double* Y, hipStream_t stream) {
  const uint32_t ncb = (uint32_t)((dims + W_TILE - 1) / W_TILE), nrb = (uint32_t)((n + W_TILE - 1) / W_TILE);
  const dim3 grid(ncb * nrb), block(256);
  const int64_t ldy = dims;
  const int lds = 4 * W_TILE * W_LD * (int)sizeof(double);          // X and P, two buffers each
#define MI_W_LAUNCH(T, ROWS)                                                                                            \
  do {                                                                                                                  \
    ensure_dynamic_lds((const void*)whiten_mfma_kernel<T, ROWS>, lds);                                                  \
    hipLaunchKernelGGL((whiten_mfma_kernel<T, ROWS>), grid, block, lds, stream, (const T*)X, n, d, rs, cs, m, P, dims, Y, \
                       ldy, ncb, nrb);                                                                                  \
  } while (0)
  if (dtype == 0) { if (cs == 1) MI_W_LAUNCH(float, true); else MI_W_LAUNCH(float, false); }
  else { if (cs == 1) MI_W_LAUNCH(double, true); else MI_W_LAUNCH(double, false); }
#undef MI_W_LAUNCH
  if (eps >= 0.0) {
    if (dims <= 2048) hipLaunchKernelGGL(rownorm_f64_kernel<8>, dim3((unsigned)n), block, 0, stream, Y, ldy, dims, eps);
    else if (dims <= 4096) hipLaunchKernelGGL(rownorm_f64_kernel<16>, dim3((unsigned)n), block, 0, stream, Y, ldy, dims, eps);
    else hipLaunchKernelGGL(rownorm_f64_wide_kernel, dim3((unsigned)n), block, 0, stream, Y, ldy, dims, eps);
  }
}

}  // namespace mi
