// Shared definitions for the gfx950 retrieval kernels (internal; the public ABI is include/mi355_retrieval.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace mi {

// ---- geometry of the tile-blocked 16-bit (fp16 | bf16) images streamed by the MFMA kernel ----
// A matrix X[rows][dp] (dp = d rounded up to 64) is stored as
//   blocked[tile = row / 256][slice = k / 32][row % 256][32]     (16-bit elements, 16 KiB per (tile, slice) block)
// with the four 16-byte chunks of each 64-byte row permuted: physical chunk = c ^ ((-(row >> 2)) & 3).
// One (tile, slice) block is exactly the LDS image of one K-slice of one operand, so the global->LDS
// DMA is a linear copy and the `ds_read_b128` fragment reads of v_mfma_f32_16x16x32_{f16,bf16} (lane l: row
// l & 15, chunk l >> 4) hit 16 distinct 16-byte slots in each of the instruction's four lane groups,
// i.e. they are bank-conflict free (derivation in DESIGN.md "LDS image").
constexpr int TILE = 256;     // gallery rows / queries per workgroup tile
constexpr int BK = 64;        // column padding granule of dp
constexpr int SLICE_K = 32;   // K-depth of one slice = one v_mfma_f32_16x16x32_{f16,bf16}
constexpr int SLICE_ELEMS = TILE * SLICE_K;          // 8192 elements = 16 KiB
constexpr int SLICE_BYTES = SLICE_ELEMS * 2;

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device): a process that opens handles on
// several devices has to opt every kernel in on each of them.  Returns the current device.
int ensure_dynamic_lds(const void* kernel, int bytes = 160 * 1024);
int current_device_cus();     // compute units of the current device (cached per device)

__host__ __device__ inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

__host__ __device__ __forceinline__ uint32_t swz_chunk(uint32_t row, uint32_t c) {
  return c ^ ((0u - (row >> 2)) & 3u);
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

// f32 -> upper 16 bits (bf16 layout), rounded toward -inf: a threshold may only get looser by the packing
__host__ __device__ __forceinline__ uint32_t bf16_down(float f) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(f);
#else
  __builtin_memcpy(&u, &f, 4);
#endif
  if (f != f) return 0xFF80u;                                  // NaN -> -inf (nothing is filtered out)
  uint32_t hi = u >> 16;
  if ((u & 0x80000000u) && (u & 0xFFFFu)) hi += 1;             // negative: truncation rounds up, step one down
  return hi & 0xFFFFu;
}

// f32 -> upper 16 bits (bf16 layout), rounded toward +inf: a COUNT level may only get stricter by the packing
__host__ __device__ __forceinline__ uint32_t bf16_up(float f) {
  if (f != f) return 0x7F80u;                                  // NaN -> +inf (nothing is counted)
  return (bf16_down(-f) ^ 0x8000u) & 0xFFFFu;
}

// survivor entry: (score bits << 32) | local row
__device__ __forceinline__ uint64_t pack_entry(float s, uint32_t row) {
  return ((uint64_t)__float_as_uint(s) << 32) | row;
}
__device__ __forceinline__ float entry_score(uint64_t e) { return __uint_as_float((uint32_t)(e >> 32)); }
__device__ __forceinline__ uint32_t entry_row(uint64_t e) { return (uint32_t)e; }

// per-row rounding statistics produced by ingest (norms of the stored f32 row, of its 16-bit image and
// of their difference) -- inputs of the rigorous error margin, DESIGN.md "Exactness certificate"
struct RowStat {
  float norm_f32;    // ||g||      (stored f32 row)
  float norm_img;    // ||g_hat||  (16-bit image row)
  float norm_diff;   // ||g_hat - g||
};

// survivor record written by the scoring kernel into wave-private segments (no atomics in the hot kernel)
struct __attribute__((aligned(16))) SurvRec {
  float score;
  uint32_t row;
  uint32_t q;
  uint32_t pad;
};

// sticky device-side flags
enum : uint32_t { FLAG_SURV_OVERFLOW = 1u, FLAG_CAND_OVERFLOW = 2u, FLAG_REC_OVERFLOW = 4u, FLAG_RANGE = 8u, FLAG_SPEC_FAIL = 16u };

constexpr uint32_t CNT_STRIDE = 32;   // one survivor counter per 128-byte line (atomics to one line serialise in L2)

// Share of the gallery tiles each XCD label (blockIdx % 8) gets from the tile kernel.  The eight XCDs of one MI355X hold
// different clocks under the same load (measured: 1.47 vs 1.54 GHz, odd vs even XCDs of one device), so an even split
// leaves the fast ones idle at the end of every launch.  `w` are relative speeds (tiles per unit time, measured by the
// kernel itself: per-workgroup s_memrealtime around its loop), updated after every large launch by block 0 of the scatter
// kernel; `cum` is their cumulative sum in 2^-20 units, read by the next launch.  Any split gives the same answers.
struct XccBalance {
  uint32_t cum[9];
  uint32_t launches;
  float w[8];
};
constexpr uint32_t XCC_ONE = 1u << 20;

struct QueryState {       // all arrays sized for qpad queries
  float* thr;             // current pass threshold (approx-score domain), +inf for padded queries
  float* margin;          // 2 * eps_q  (rigorous |approx - exact| bound, both sides)
  uint32_t* cnt;          // survivors appended: counter of query q at cnt[q * CNT_STRIDE]
  uint64_t* surv;         // [qpad][cap]
  uint32_t* flags;        // [0] sticky error flags (one word for both workspaces of a handle)
  uint32_t* repair;       // 'repair needed' word of the CURRENT batch (speculative threshold failed for some query): per
                          // workspace, since the pre part of the next batch may run beside this batch's maintain launch
  float* thr2;            // fallback (looser) speculative threshold per query
  uint32_t* qflag;        // per query: 1 = its speculative threshold failed verification, repair it
  // in-launch threshold ladder (tile kernel, single-launch schedule; DESIGN.md "Thresholds while streaming" (iii)):
  float* lad_tc;          // count level t_c: a sample order statistic tighter than the speculative one (+inf = off)
  uint32_t* lad_pack;     // bf16(thr, rounded down) | bf16(t_c - margin, rounded down) << 16: the two thresholds a wave may apply
  uint32_t* lad_cnt;      // rows seen with approx >= t_c; once >= K, t_c - margin is a RIGOROUS threshold
  uint32_t cap;
};

}  // namespace mi
