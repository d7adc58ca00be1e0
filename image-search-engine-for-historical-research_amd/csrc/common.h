// Shared definitions for the gfx950 retrieval kernels (internal; the public ABI is include/mi355_retrieval.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace mi {

// ---- geometry of the tile-blocked bf16 images streamed by the MFMA kernel -------------------
// A matrix X[rows][dp] (dp = d rounded up to 64) is stored as
//   blocked[tile = row / 256][ks = k / 64][row % 256][64]      (bf16, 32 KiB per (tile, ks) block)
// with the eight 16-byte chunks of each 128-byte row permuted: physical chunk = c ^ ((row >> 1) & 7).
// One (tile, ks) block is exactly the LDS image of one K-step, so the global->LDS DMA is a linear
// copy and `ds_read_b128` of the MFMA fragments is bank-conflict free (see DESIGN.md "LDS image").
constexpr int TILE = 256;     // gallery rows / queries per workgroup tile
constexpr int BK = 64;        // K-step (bf16 elements)
constexpr int BLOCK_ELEMS = TILE * BK;

__host__ __device__ inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

__device__ __forceinline__ uint32_t swz_chunk(uint32_t row, uint32_t c) { return c ^ ((row >> 1) & 7u); }

// element offset of (row, k) inside the blocked image
__device__ __forceinline__ int64_t blocked_offset(int64_t row, int32_t k, int32_t ksteps) {
  const int64_t tile = row / TILE;
  const uint32_t r = (uint32_t)(row % TILE);
  const uint32_t ks = (uint32_t)k / BK, kk = (uint32_t)k % BK;
  const uint32_t c = kk >> 3;
  return ((tile * ksteps + ks) * (int64_t)BLOCK_ELEMS) + (int64_t)r * BK + (swz_chunk(r, c) << 3) + (kk & 7);
}

// float <-> order-preserving uint32 key (larger float -> larger key); NaN maps below -inf
__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  if (f != f) return 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

// survivor entry: (score bits << 32) | local row
__device__ __forceinline__ uint64_t pack_entry(float s, uint32_t row) {
  return ((uint64_t)__float_as_uint(s) << 32) | row;
}
__device__ __forceinline__ float entry_score(uint64_t e) { return __uint_as_float((uint32_t)(e >> 32)); }
__device__ __forceinline__ uint32_t entry_row(uint64_t e) { return (uint32_t)e; }

// per-row rounding statistics produced by ingest (norms of the stored f32 row, of its bf16 image and
// of their difference) -- inputs of the rigorous error margin, DESIGN.md "Exactness certificate"
struct RowStat {
  float norm_f32;    // ||g||      (stored f32 row)
  float norm_bf16;   // ||g_hat||  (bf16 image)
  float norm_diff;   // ||g_hat - g||
};

// sticky device-side flags
enum : uint32_t { FLAG_SURV_OVERFLOW = 1u, FLAG_CAND_OVERFLOW = 2u };

struct QueryState {       // all arrays sized for qpad queries
  float* thr;             // current pass threshold (approx-score domain), +inf for padded queries
  float* margin;          // 2 * eps_q  (rigorous |approx - exact| bound, both sides)
  uint32_t* cnt;          // survivors appended
  uint64_t* surv;         // [qpad][cap]
  uint32_t* flags;        // [1]
  uint32_t cap;
};

}  // namespace mi
