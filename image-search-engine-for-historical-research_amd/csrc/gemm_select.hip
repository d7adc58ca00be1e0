// bf16 MFMA scoring kernel with fused survivor filter (gfx950 / CDNA4).
//
// Replaces the hot loop of matching_L2 (src/utils/nnsearch.py:699-703: per query an N x D temporary,
// a norm and a full argsort) and the `vecs.T @ qvecs` + argsort of src/main_retrieve.py:175-176 /
// src/utils/Reranking.py:206-207.  S = G_hat (rows) x Q_hat^T is a dense contraction, so it runs on
// v_mfma_f32_16x16x32_bf16; the Q x N score matrix is never written: every accumulator is compared
// with its query's running threshold in registers and only survivors are appended.
//
// Geometry: workgroup = 512 threads = 8 waves as 2 (gallery) x 4 (query); tile 256 gallery rows x 256
// queries x BK 64; per wave 128 x 64 outputs = 8 x 4 MFMA blocks of 16x16 (128 accumulator VGPRs).
// Operands are streamed from the tile-blocked images (common.h) by global_load_lds_dwordx4 into a
// double-buffered LDS ring (2 x (32 KiB A + 32 KiB B) = 128 KiB, one workgroup per CU); the images
// are already chunk-swizzled so that the DMA is linear and ds_read_b128 is conflict free.
#include "common.h"
#include "kernels.h"

namespace mi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)ldst, 16, 0, 0);
}

__device__ __forceinline__ void append_survivor(const QueryState& st, uint32_t q, float v, uint32_t row) {
  const uint32_t pos = atomicAdd(&st.cnt[q], 1u);
  if (pos < st.cap)
    st.surv[(uint64_t)q * st.cap + pos] = pack_entry(v, row);
  else
    atomicOr(st.flags, FLAG_SURV_OVERFLOW);
}

template <bool FIRST>
__global__ __launch_bounds__(512, 2) void gemm_select_kernel(ScoreArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware tile map: blocks b and b+8 share an XCD (round-robin dispatch), so the nqt query tiles of
  // one gallery tile are given to consecutive blocks of one XCD and the gallery tile is fetched from
  // HBM once and served to the other query tiles from that XCD's L2.
  const uint32_t b = blockIdx.x;
  const uint32_t xcd = b & 7u, j = b >> 3;
  const uint32_t qt = j % (uint32_t)p.nqt;
  const uint32_t tl = (j / (uint32_t)p.nqt) * 8u + xcd;
  if (tl >= (uint32_t)p.ntiles) return;
  const uint32_t gt = (uint32_t)p.tile0 + tl;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int l15 = lane & 15, lq = lane >> 4;

  const char* gA = (const char*)p.gal_bf16 + (int64_t)gt * p.ksteps * (BLOCK_ELEMS * 2);
  const char* gB = (const char*)p.qry_bf16 + (int64_t)qt * p.ksteps * (BLOCK_ELEMS * 2);

  // stage K-step ks into ring slot buf: each wave copies 4 KiB of A and 4 KiB of B (8 DMA pieces of 1 KiB)
  auto stage = [&](int buf, int ks) {
    const char* sa = gA + (int64_t)ks * (BLOCK_ELEMS * 2) + w * 4096 + lane * 16;
    const char* sb = gB + (int64_t)ks * (BLOCK_ELEMS * 2) + w * 4096 + lane * 16;
    char* la = smem + buf * 32768 + w * 4096;
    char* lb = smem + 65536 + buf * 32768 + w * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(sa + i * 1024, la + i * 1024);
      glds16(sb + i * 1024, lb + i * 1024);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes inside one 32 KiB operand image); kk = 1 is the same offset ^ 64
  const uint32_t sw = (uint32_t)(l15 >> 1);
  const uint32_t a_off = (uint32_t)(wr * 128 + l15) * 128u + ((((uint32_t)lq) ^ sw) << 4);
  const uint32_t b_off = (uint32_t)(wc * 64 + l15) * 128u + ((((uint32_t)lq) ^ sw) << 4);

  stage(0, 0);
  const int KS = p.ksteps;
  for (int ks = 0; ks < KS; ++ks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ks + 1 < KS) stage((ks + 1) & 1, ks + 1);
    const char* Ab = smem + (ks & 1) * 32768;
    const char* Bb = smem + 65536 + (ks & 1) * 32768;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[8], bfr[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        bfr[nb] = *reinterpret_cast<const bf16x8*>(Bb + ((b_off ^ (kk * 64)) + nb * 2048));
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
        af[mb] = *reinterpret_cast<const bf16x8*>(Ab + ((a_off ^ (kk * 64)) + mb * 2048));
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
    }
  }

  // ---- epilogue: C layout of 16x16x32: column (query) = lane & 15, row (gallery) = (lane >> 4) * 4 + reg
  const uint32_t row_base = gt * TILE + wr * 128 + lq * 4;          // + mb*16 + reg
  const uint32_t q_base = qt * TILE + wc * 64 + l15;                // + nb*16
  const int64_t rows_valid = p.n - (int64_t)gt * TILE;              // rows of this tile that exist
  const bool full_tile = rows_valid >= TILE;

  if (FIRST) {
    // bootstrap chunk: keep everything, slot = local row (chunk starts at row 0 of the shard)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const uint32_t q = q_base + nb * 16;
      if (q < (uint32_t)p.nq) {
        uint64_t* dst = p.st.surv + (uint64_t)q * p.st.cap;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t row = row_base + mb * 16 + r;
            if (row < (uint64_t)p.n) dst[row] = pack_entry(acc[mb][nb][r], row);
          }
      }
    }
    return;
  }

#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const uint32_t q = q_base + nb * 16;
    const float thr = p.st.thr[q];      // +inf for padded queries
    float m = acc[0][nb][0];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mb][nb][r]);
    if (!__any(m >= thr)) continue;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[mb][nb][r];
        const uint32_t row = row_base + mb * 16 + r;
        if (v >= thr && (full_tile || row < (uint64_t)p.n)) append_survivor(p.st, q, v, row);
      }
  }
}

void launch_gemm_select(const ScoreArgs& a, bool first, hipStream_t stream) {
  const unsigned groups = (unsigned)((a.ntiles + 7) / 8);
  const unsigned grid = groups * 8u * (unsigned)a.nqt;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)gemm_select_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)gemm_select_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_done = true;
  }
  if (first)
    hipLaunchKernelGGL(gemm_select_kernel<true>, dim3(grid), dim3(512), 131072, stream, a);
  else
    hipLaunchKernelGGL(gemm_select_kernel<false>, dim3(grid), dim3(512), 131072, stream, a);
}

}  // namespace mi
