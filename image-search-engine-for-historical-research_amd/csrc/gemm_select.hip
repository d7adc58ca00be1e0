// bf16 MFMA scoring kernel with fused survivor filter (gfx950 / CDNA4).
//
// Replaces the hot loop of matching_L2 (src/utils/nnsearch.py:699-703: per query an N x D temporary,
// a norm and a full argsort) and the `vecs.T @ qvecs` + argsort of src/main_retrieve.py:175-176 /
// src/utils/Reranking.py:206-207.  S = G_hat (rows) x Q_hat^T is a dense contraction, so it runs on
// v_mfma_f32_16x16x32_bf16; the Q x N score matrix is never written: every accumulator is compared
// with its query's running threshold in registers and only survivors are appended.
//
// Structure (DESIGN.md "Scoring kernel"):
//  * persistent grid, one 512-thread workgroup per CU, walking (gallery tile, query tile) pairs in an
//    XCD-aware order (the query tiles of one gallery tile run on one XCD, so a gallery tile is fetched
//    from HBM once and re-served from that XCD's L2);
//  * tile 256 gallery rows x 256 queries; 8 waves as 2 (gallery) x 4 (query), 128 x 64 outputs per wave
//    = 8 x 4 blocks of 16x16 (128 accumulator VGPRs);
//  * operands stream as K-slices of 32 (one MFMA depth), 16 KiB of A (gallery) + 16 KiB of B (queries) per
//    slice, copied by global_load_lds_dwordx4 into LDS rings that never drain (counted `s_waitcnt vmcnt(N)`,
//    raw `s_barrier`).  vmcnt retires in order per wave, so the loader roles are split by wave group:
//    waves 0-3 stream only A through a 5-slot ring (4 slices = 64 KiB in flight: the gallery comes from HBM,
//    ~2 us loaded latency), waves 4-7 stream only B through a 4-slot ring (3 in flight: the query tile is
//    L2-resident).  4 DMA pieces of 1 KiB per wave per slice either way, issued between the MFMAs;
//  * the two wave groups (one wave of each per SIMD) are staggered by one barrier: while one group issues its
//    32 MFMAs of a slice, the other reads its fragments (12 ds_read_b128) -- the matrix pipe of every SIMD
//    alternates between its two waves and stays busy;
//  * the rings keep running across tile boundaries (no prologue/epilogue bubble per tile).
#include <hip/hip_ext.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace mi {

static thread_local hipEvent_t g_launch_ev[2] = {nullptr, nullptr};
void set_launch_events(hipEvent_t start, hipEvent_t stop) {
  g_launch_ev[0] = start;
  g_launch_ev[1] = stop;
}
void take_launch_events(hipEvent_t* start, hipEvent_t* stop) {
  *start = g_launch_ev[0];
  *stop = g_launch_ev[1];
  g_launch_ev[0] = g_launch_ev[1] = nullptr;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

constexpr int A_SLOTS = 5, B_SLOTS = 4;       // 6 + 3 measured slower (3.88 vs 3.78 ms): the query stream needs its lead too
constexpr int A_RING = 0, B_RING = A_SLOTS * SLICE_BYTES;                 // byte offsets in LDS
constexpr int RING_BYTES = (A_SLOTS + B_SLOTS) * SLICE_BYTES;             // 144 KiB
constexpr int HIT_SLOTS = 8;                                // per-wave filter scratch: 8 (lane, block) pairs x 32 scores + meta
constexpr int WAVE_SCRATCH = HIT_SLOTS * 32 * 4 + HIT_SLOTS * 16;         // 1152 B
constexpr int STAGE_BYTES = 8 * WAVE_SCRATCH;               // 9 KiB per workgroup
constexpr int THR_WORDS = 192;                              // per wave: thresholds | ladder counters | ladder levels (64 each)

// LDS stores of the filter scratch as inline asm.  The compiler orders every DS store it can see behind ALL pending
// LDS-DMA transfers (it cannot tell that the scratch and the rings are disjoint) with an s_waitcnt vmcnt(0), which
// drains the DMA rings at every tile boundary; the hardware needs no such wait for disjoint addresses.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(3))) uint32_t u32x3;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const LDS_AS char*)(const char*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_store16(uint32_t addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_store16u(uint32_t addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const char* gsrc, char* ldst) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)ldst, 16, 0, 0);
}


// DBG: diagnostics-only build variants; 0 = product.  bit0 (1) skip DMA, bit1 (2) skip MFMA, bit2 (4) skip the filter,
// bit3 (8) per-segment stamps, bit5 (32) skip the gallery DMA only, bit6 (64) skip the query DMA only, bit7 (128) read the
// fragments once (no LDS reads in the loop), bit8 (256) no barriers in the loop.  Every diagnostic build without stamps
// records the in-kernel clock (s_memtime / s_memrealtime around the loop) in dbg[6], dbg[7].
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// REPAIR: the conditional second pass of the speculative schedule (its own instantiation, so that profiles of the
// main launch are not diluted by repair launches that exit immediately)
template <bool FIRST, int DBG, bool F16, bool REPAIR = false>
__global__ __launch_bounds__(512, 2) void gemm_select_kernel(ScoreArgs p) {
  using frag_t = typename std::conditional<F16, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // rings | per-wave scratch | per-wave thresholds  (ONE LDS object)

  // ---- work assignment.  Blocks b, b+8, ... share an XCD (round-robin dispatch; speed only).  XCD label x
  // owns gallery tiles tl = x (mod 8); its virtual list v -> (tl = (v / nqt) * 8 + x, qt = v % nqt) is dealt
  // round-robin to the nwg blocks of that label, so concurrently running blocks share gallery tiles.
  if (REPAIR && *p.cond == 0) return;                           // repair pass that is not needed
  const uint32_t b = blockIdx.x, nwg = gridDim.x >> 3;
  const uint32_t xcd = b & 7u, j = b >> 3;
  const uint32_t nqt = (uint32_t)p.nqt;
  const uint32_t cnt_x = ((uint32_t)p.ntiles > xcd) ? ((uint32_t)p.ntiles - xcd + 7u) / 8u : 0u;
  const uint32_t nvirt = cnt_x * nqt;
  if (j >= nvirt) {
    if (!FIRST && (threadIdx.x & 63) == 0) p.rec_cnt[b * 8 + (threadIdx.x >> 6)] = 0;
    return;
  }
  const uint32_t my_tiles = (nvirt - j + nwg - 1) / nwg;
  const uint32_t KSL = (uint32_t)p.nslices;
  const uint32_t T_total = my_tiles * KSL;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w >> 2;                 // wave group = gallery half (wr)
  const int wr = grp, wc = w & 3;
  const int l15 = lane & 15, lq = lane >> 4;
  // thresholds of this wave's 64 queries (of the current query tile), private to the wave: no cross-wave sync;
  // reloaded (a plain global load, which drains the DMA ring) only when the workgroup's query tile changes,
  // which never happens when the blocks-per-XCD count is a multiple of the query-tile count
  float* thr_w = reinterpret_cast<float*>(smem + RING_BYTES + STAGE_BYTES) + w * 64;
  uint32_t thr_qt = 0xFFFFFFFFu;


  auto tile_of = [&](uint32_t i, uint32_t& gt, uint32_t& qt) {
    const uint32_t v = j + i * nwg;
    qt = v % nqt;
    gt = (uint32_t)p.tile0 + (v / nqt) * 8u + xcd;
  };

  // ---- DMA prefetch state.  Group 0 owns the A stream (lead 4 slices), group 1 the B stream (lead 3 slices).
  // A wave copies 4 KiB (4 pieces) of its operand's 16 KiB slice block.
  uint32_t pf_i = 0, pf_sl = 0;
  // The source address is kept as a wave-uniform base (scalar registers) plus a constant per-lane byte offset, so that
  // the DMA instruction takes the saddr + 32-bit voffset form and advancing the stream costs scalar adds only: 64-bit
  // vector adds in the load segment would compete with the partner wave's MFMAs for the SIMD's vector issue.
  const char* pf;                                          // uniform
  const uint32_t pf_lane = (uint32_t)lane * 16u;
  auto pf_set = [&](uint32_t i) {
    uint32_t gt, qt;
    tile_of(i < my_tiles ? i : my_tiles - 1, gt, qt);     // past the end: harmless re-load of the last tile
    pf = (grp == 0 ? (const char*)p.gal_img + (int64_t)gt * KSL * SLICE_BYTES
                   : (const char*)p.qry_img + (int64_t)qt * KSL * SLICE_BYTES) + (w & 3) * 4096;
  };
  pf_set(0);
  constexpr bool dbg_nomfma = DBG & 2;
  const bool dbg_nodma = (DBG & 1) || ((DBG & 32) && grp == 0) || ((DBG & 64) && grp == 1);
  const uint32_t ring_base = (grp == 0 ? A_RING : B_RING) + (w & 3) * 4096;
  uint32_t wr_slot = 0;                                   // ring slot the next issued slice goes to
  const uint32_t my_slots = grp == 0 ? A_SLOTS : B_SLOTS;
  uint32_t off = 0;
  auto issue_piece = [&](int piece) {
    // the zero extension of the lane offset has to be visible in this basic block for the saddr form to be selected;
    // the piece is the instruction's immediate offset, which the hardware adds to the global AND the LDS address, so
    // the four pieces of a slice share one scalar base, one offset register and one M0 value
    if (!dbg_nodma) {
      const GLOBAL_AS void* src = (const GLOBAL_AS void*)(pf + off);
      LDS_AS void* dst = (LDS_AS void*)(smem + ring_base + wr_slot * SLICE_BYTES);
      switch (piece) {
        case 0: __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0); break;
        case 1: __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0); break;
        default: __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0); break;
      }
    }
  };
  auto issue_advance = [&]() {
    pf += SLICE_BYTES;
    if (++pf_sl == KSL) {
      pf_sl = 0;
      pf_set(++pf_i);
    }
    if (++wr_slot == my_slots) wr_slot = 0;
  };
  auto issue = [&]() {
    off = pf_lane;
    asm volatile("" : "+v"(off));
    issue_piece(0);
    issue_piece(1);
    issue_piece(2);
    issue_piece(3);
    issue_advance();
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside one operand image ([256 rows][32 k] bf16, 64-byte rows, chunk-swizzled)
  const uint32_t fsw = (0u - (uint32_t)(l15 >> 2)) & 3u;
  const uint32_t a_off = (uint32_t)(wr * 128 + l15) * 64u + ((((uint32_t)lq) ^ fsw) << 4);
  uint32_t b_off = (uint32_t)B_RING + (uint32_t)(wc * 64 + l15) * 64u + ((((uint32_t)lq) ^ fsw) << 4);
  // opaque to constant folding: otherwise B_RING (80 KiB, beyond the 16-bit ds offset field) is split off again, every
  // B fragment read gets an address register of its own, and the registers' reuse puts an lgkmcnt(0) wait in the
  // middle of the read burst
  asm volatile("" : "+v"(b_off));

  uint32_t cur_i = 0, cur_sl = 0, gt, qt;
  tile_of(0, gt, qt);
  // wave-private survivor record segment: positions come from ballot/popcount, so the hot kernel issues no
  // returning atomics (a returning atomic forces vmcnt(0) and drains the DMA ring)
  SurvRec* my_rec = p.rec + (uint64_t)(b * 8 + w) * p.rec_cap;
  uint32_t my_cnt = 0;
  // filter scratch of this wave: scores of up to HIT_SLOTS "hit" lanes (32 each) + their (thr, q, row base)
  float* sc_val = reinterpret_cast<float*>(smem + RING_BYTES + w * WAVE_SCRATCH);  // after both rings
  uint4* sc_meta = reinterpret_cast<uint4*>(smem + RING_BYTES + w * WAVE_SCRATCH + HIT_SLOTS * 32 * 4);
  const uint32_t sc_val_lds = lds_addr(sc_val), sc_meta_lds = lds_addr(sc_meta);

  // ---- per-tile filter of the accumulators (+ reset).  Called by group 1 right after its last MFMA segment of
  // the tile and by group 0 ONE INTERVAL LATER (before its first MFMA segment of the next tile), so that both
  // groups filter in the same barrier interval instead of stalling each other in two different ones.
  unsigned long long d_e1 = 0, d_e2 = 0, d_hits = 0;
  auto tile_epilogue = [&](uint32_t gt, uint32_t qt) {
      // ---- tile finished: filter.  C layout of 16x16x32: column (query) = lane & 15, row = (lane >> 4) * 4 + reg
    const uint32_t row_base = gt * TILE + wr * 128 + lq * 4;          // + mb*16 + reg
    const uint32_t ql_base = qt * TILE + wc * 64 + l15;               // + nb*16
    if (DBG & 4) {
      // diagnostics: no filter, accumulators kept live
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) asm volatile("" ::"v"(acc[mb][nb]));
    } else if (FIRST) {
      // bootstrap chunk: keep everything, slot = local row (the chunk starts at row 0 of the shard)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const uint32_t q = ql_base + nb * 16;
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
    } else {
      if (qt != thr_qt) {
        thr_w[lane] = p.st.thr[qt * TILE + wc * 64 + lane];
        thr_qt = qt;
      }
      // Filter.  The common case (no score of this lane reaches its query's threshold) is branch-free VALU: a
      // 32-value max per (lane, query block), for all four query blocks first.  Then ALL hit (lane, block) pairs
      // dump their 32 scores into the per-wave LDS scratch in one burst and ONE rolled loop scans them: one LDS
      // write->read latency chain per tile instead of one per query block (the partner wave group saturates the
      // LDS with fragment reads meanwhile, so every dependent LDS round trip costs hundreds of cycles), and the
      // unrolled code stays small (a fully unrolled compare+append per accumulator was measured 17 % slower).
      unsigned long long te0 = 0, te1 = 0, te2 = 0;
      if (DBG & 8) te0 = stamp();
      float thr4[4];
      unsigned long long hm[4];
      uint32_t base[5];
      base[0] = 0;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        thr4[nb] = thr_w[nb * 16 + l15];      // +inf for padded queries
        float m = acc[0][nb][0];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mb][nb][r]);
        hm[nb] = __ballot(m >= thr4[nb]);
        base[nb + 1] = base[nb] + (uint32_t)__popcll(hm[nb]);
      }
      const uint32_t total = base[4];
      if (DBG & 8) te1 = stamp();
      for (uint32_t r0 = 0; r0 < total; r0 += HIT_SLOTS) {          // almost always zero or one round
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          if (hm[nb] == 0) continue;                                  // wave-uniform: no write burst for this block
          const uint32_t rank = base[nb] + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm[nb] >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)hm[nb], 0u));
          const bool mine = (hm[nb] >> lane) & 1ull;
          if (mine && rank >= r0 && rank < r0 + HIT_SLOTS) {
            const uint32_t dst = sc_val_lds + (rank - r0) * 128;
            lds_store16<0>(dst, acc[0][nb]);
            lds_store16<16>(dst, acc[1][nb]);
            lds_store16<32>(dst, acc[2][nb]);
            lds_store16<48>(dst, acc[3][nb]);
            lds_store16<64>(dst, acc[4][nb]);
            lds_store16<80>(dst, acc[5][nb]);
            lds_store16<96>(dst, acc[6][nb]);
            lds_store16<112>(dst, acc[7][nb]);
            lds_store16u(sc_meta_lds + (rank - r0) * 16,
                         (u32x4){__float_as_uint(thr4[nb]), ql_base + nb * 16, row_base, 0u});
          }
        }
        const uint32_t nslots = min(total - r0, (uint32_t)HIT_SLOTS);
        // scan of nslots x 32 scores (entry e -> slot e >> 5, value index e & 31 = mb * 4 + r): all LDS reads of the
        // round are issued before the first ballot, so the round pays ONE read latency, not one per 64 entries
        constexpr int SCAN = HIT_SLOTS * 32 / 64;
        u32x3 mt[SCAN];                                              // (threshold, query, row base)
        float vv[SCAN];
#pragma unroll
        for (int it = 0; it < SCAN; ++it) {
          const uint32_t e = it * 64 + lane;
          const bool valid = e < nslots * 32;
          // inline asm for the same reason as the stores (a visible DS load of the scratch is ordered behind the
          // LDS-DMA transfers as well); the results become usable after the lgkmcnt(0) below, which every value passes
          asm volatile("ds_read_b96 %0, %2\n\tds_read_b32 %1, %3"
                       : "=&v"(mt[it]), "=v"(vv[it])      // mt must not share a register with the second address
                       : "v"(sc_meta_lds + (valid ? (e >> 5) : 0u) * 16u), "v"(sc_val_lds + (valid ? e : 0u) * 4u)
                       : "memory");
        }
        static_assert(SCAN == 4, "the wait below lists the scan registers explicitly");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(mt[0]), "+v"(mt[1]), "+v"(mt[2]), "+v"(mt[3]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3])
                     :
                     : "memory");
#pragma unroll
        for (int it = 0; it < SCAN; ++it) {
          if ((uint32_t)(it * 64) >= nslots * 32) break;              // wave-uniform
          const uint32_t e = it * 64 + lane;
          const uint32_t i = e & 31u;
          const uint32_t row = mt[it].z + (i >> 2) * 16 + (i & 3u);
          const bool keep = e < nslots * 32 && vv[it] >= __uint_as_float(mt[it].x) && row < (uint64_t)p.n;
          const unsigned long long km = __ballot(keep);
          if (km) {
            const uint32_t pos = my_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            if (keep && pos < p.rec_cap)
              reinterpret_cast<uint4*>(my_rec)[pos] = make_uint4(__float_as_uint(vv[it]), row, mt[it].y, 0u);
            my_cnt += (uint32_t)__popcll(km);
          }
        }
      }
      if (DBG & 8) { te2 = stamp(); d_e1 += te1 - te0; d_e2 += te2 - te1; d_hits += total; }
      // no LDS operation may stay pending past the filter: the scan's early exit leaves unused reads in flight, and the
      // compiler then protects their destination registers (reused for fragments) with an lgkmcnt(0) wait in the middle
      // of EVERY slice's fragment read burst.  A real s_waitcnt instruction (not inline asm) so that its counter
      // tracking sees it: lgkmcnt(0), vmcnt / expcnt untouched.
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  uint32_t ep_gt = 0, ep_qt = 0;
  bool ep_pending = false;

  // ---- prologue: group 0 puts A(0..3) in flight, group 1 B(0..2); slice 0 landed for everybody
  if (grp == 0) {
#pragma unroll
    for (int d = 0; d < A_SLOTS - 1; ++d) issue();
    vm_wait<(A_SLOTS - 2) * 4>();
  } else {
#pragma unroll
    for (int d = 0; d < B_SLOTS - 1; ++d) issue();
    vm_wait<(B_SLOTS - 2) * 4>();
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave group by one barrier

  uint32_t a_rd = 0, b_rd = 0;                           // ring slots holding slice S
  unsigned long long t_abs0 = 0, t_abs2 = 0;
  unsigned long long clk0 = 0, rt0 = 0;
  if (DBG & 8) { clk0 = stamp(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  unsigned long long d_load = 0, d_b1 = 0, d_mfma = 0, d_b2 = 0, d_epi = 0, tt0 = 0, tt1 = 0, tt2 = 0, tt3 = 0, tt4 = 0;
  frag_t af[8], bfr[4];
  if ((DBG & ~8) != 0 && !(DBG & 8)) { clk0 = stamp(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  for (uint32_t S = 0; S < T_total; ++S) {
    if (DBG & 8) tt0 = stamp();
    // ================= LOAD segment (the partner group is in its MFMA segment) =================
    const char* abase = smem + a_rd * SLICE_BYTES;
    const char* bbase = smem + b_rd * SLICE_BYTES;
    if (++a_rd == A_SLOTS) a_rd = 0;
    if (++b_rd == B_SLOTS) b_rd = 0;
    if (!(DBG & 128) || S == 0) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) bfr[nb] = *reinterpret_cast<const frag_t*>(bbase + b_off + nb * 1024);
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) af[mb] = *reinterpret_cast<const frag_t*>(abase + a_off + mb * 1024);
    }
    if (DBG & 128) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+v"(bfr[nb]));
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) asm volatile("" : "+v"(af[mb]));
    }
    // DMA issue sits behind the 12 fragment reads: its ~60 cycles per piece overlap the LDS read latency instead
    // of stalling this wave's MFMAs (in the MFMA segment the partner wave cannot fill the matrix pipe).
    // group 0 issues A(S+4) into the slot of A(S-1), group 1 issues B(S+3) into the slot of B(S-1); both groups'
    // reads of slice S-1 retired before the barrier behind us.
    __builtin_amdgcn_sched_barrier(0);
    issue();
    // group 1 (B loader): B(S+1) landed before the barrier that opens group 0's LOAD(S+1); B(S+2), B(S+3) may fly
    if (grp == 1) vm_wait<(B_SLOTS - 2) * 4>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired BEFORE the barrier: frees the slots (WAR)
    __builtin_amdgcn_sched_barrier(0);
    if (DBG & 8) tt1 = stamp();
    if (!(DBG & 256)) __builtin_amdgcn_s_barrier();
    if (DBG & 8) tt2 = stamp();
    __builtin_amdgcn_sched_barrier(0);
    // ================= MFMA segment: 32 back-to-back MFMAs =================
    if (grp == 0 && ep_pending) {          // deferred filter of the previous tile (fragments of slice S stay live)
      tile_epilogue(ep_gt, ep_qt);
      ep_pending = false;
    }
    // no s_setprio around the MFMAs: measured 1.4 % faster without (A/B on one box, 1338 vs 1318 TF)
    if (!dbg_nomfma) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          if constexpr (F16)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
          else
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
    } else {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) asm volatile("" ::"v"(af[mb]));
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) asm volatile("" ::"v"(bfr[nb]));
    }
    // group 0 (A loader): A(S+1) landed, A(S+2..S+4) may be in flight
    if (grp == 0) vm_wait<(A_SLOTS - 2) * 4>();
    __builtin_amdgcn_sched_barrier(0);
    if (DBG & 8) tt3 = stamp();
    if (!(DBG & 256)) __builtin_amdgcn_s_barrier();
    if (DBG & 8) {
      tt4 = stamp();
      d_load += tt1 - tt0; d_b1 += tt2 - tt1; d_mfma += tt3 - tt2; d_b2 += tt4 - tt3;
      if (S == 200) { t_abs0 = tt0; t_abs2 = tt2; }
    }
    __builtin_amdgcn_sched_barrier(0);

    if (++cur_sl == KSL) {
      cur_sl = 0;
      if (grp == 1) tile_epilogue(gt, qt);
      else { ep_gt = gt; ep_qt = qt; ep_pending = true; }
      ++cur_i;
      if (DBG & 8) d_epi += stamp() - tt4;
      tile_of(cur_i < my_tiles ? cur_i : my_tiles - 1, gt, qt);
    }
  }
  if (grp == 0 && ep_pending) tile_epilogue(ep_gt, ep_qt);
  if (grp == 0) __builtin_amdgcn_s_barrier();            // balance the stagger barrier
  if ((DBG & ~8) != 0 && !(DBG & 8) && lane == 0) {
    unsigned long long* dbgp = p.dbg + (uint64_t)(b * 8 + w) * 8;
    dbgp[5] = T_total; dbgp[6] = stamp() - clk0; dbgp[7] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
  if ((DBG & 8) && lane == 0) {
    unsigned long long* dbgp = p.dbg + (uint64_t)(b * 8 + w) * 8;
    dbgp[0] = d_load; dbgp[1] = d_b1; dbgp[2] = d_mfma; dbgp[3] = d_b2; dbgp[4] = d_epi; dbgp[5] = T_total; dbgp[6] = (p.debug & 16) ? (stamp() - clk0) : d_e1; dbgp[7] = (p.debug & 16) ? (__builtin_amdgcn_s_memrealtime() - rt0) : d_e2 + (d_hits << 40);
  }
  if (!FIRST && lane == 0) {
    p.rec_cnt[b * 8 + w] = my_cnt < p.rec_cap ? my_cnt : p.rec_cap;
    if (my_cnt > p.rec_cap) atomicOr(p.st.flags, FLAG_REC_OVERFLOW);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trailing (unused) DMA pieces land before the LDS is released
}


// =====================================================================================================================
// Structure 2 of the tile kernel: the same tile, rings, ping-pong and filter, with the code laid out for the
// instruction fetch.  The per-slice loop of structure 1 carries the tile epilogue (the filter, ~10 KB of code) inside
// its body, so every slice hops over it with three far taken branches, one of them right behind the barrier that opens
// the MFMA segment; the probe of scripts/mfma_probe.hip (ping-pong of bare 32-MFMA segments: 16.5 cycles per MFMA)
// against structure 1 with neither DMA nor LDS reads (18.7) locates ~70 cycles per segment there.  Here
//   * each wave group runs its own copy of the loop (GRP is a compile-time constant: no group tests in the loop),
//   * the slices of a tile are an inner loop whose body is straight-line code (LOAD, barrier, 32 MFMAs, barrier),
//   * slice 0 of a tile is peeled (group 0 runs its deferred filter there), the filter sits outside the inner loop.
// ORDER 1 issues the MFMAs query-block-major (the B fragment stays on the operand bus for 8 MFMAs instead of the A
// fragment for 4).
// OPT (round 3): bit 0 = the first K-slice of a tile accumulates onto the inline constant 0 (no 128-register reset after the
// filter); bit 1 = the filter's decide step (32-value maxima per lane and query block, ballots) is computed inside the MFMA
// segment of the tile's LAST K-slice, in the issue gaps of the matrix instructions, instead of after it.
// POL (round 4): cache policy of the DMA pieces, gallery aux | query aux << 8 (aux of global_load_lds: 1 = sc0, 2 = nt, 16 = sc1).
// DBG 16384: the query fragments are read from LDS once per launch (energy model of a query operand that bypasses LDS);
// DBG 32768: every wave also loads its 4 KiB of query fragments per slice straight into (discarded) registers -- with
// DBG 64 | 16384 the traffic of the "global -> VGPR query operand" structure without its pipeline (scripts/kbench.hip).
// (A second ladder level -- round 4, "LAD2" -- was built and measured at +1.3 %: profiles/r04k_ladder2_ab.txt; removed in round 5.)
template <bool FIRST, int DBG, bool F16, bool REPAIR, int ORDER, int OPT = 0, int POL = 0>
__global__ __launch_bounds__(512, 2) void gemm_tile_kernel(ScoreArgs p) {
  constexpr bool ZC = (OPT & 1) != 0;
  constexpr int BSL = B_SLOTS;                               // slots of the query ring
  constexpr bool INTER = (OPT & 2) != 0 && !FIRST && !(DBG & (4 | 4096 | 8192));
  using frag_t = typename std::conditional<F16, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // rings | per-wave scratch | per-wave thresholds  (ONE LDS object)
  if (REPAIR && *p.cond == 0) return;
  const uint32_t b = blockIdx.x, nwg = gridDim.x >> 3;
  const uint32_t xcd = b & 7u, j = b >> 3;
  const uint32_t nqt = (uint32_t)p.nqt;
  // gallery tiles of this XCD label: the contiguous range [start_x, start_x + cnt_x) of the launch's tiles, sized by the
  // label's measured speed (XccBalance); rounded to whole rounds of the label's workgroups when the launch is large
  // enough for that (every workgroup of an XCD then gets the same number of tiles)
  const uint32_t ntl = (uint32_t)p.ntiles;
  // walk 1 (A/B, ScoreArgs::walk): XCD labels 2y and 2y + 1 share the gallery range of both and take one half of the query
  // tiles each -- half the query image per L2, every gallery tile fetched by two XCDs (the second time from the Infinity Cache)
  const bool pairs = p.walk == 1 && nqt >= 2 && (nqt & 1u) == 0;
  const uint32_t nqt_l = pairs ? nqt / 2u : nqt;                      // query tiles this label works on
  const uint32_t qt0 = pairs ? (xcd & 1u) * nqt_l : 0u;
  uint32_t unit = (nwg >= nqt_l && nwg % nqt_l == 0) ? nwg / nqt_l : 1u;   // gallery tiles per round of the label's workgroups
  if (ntl < 8u * 32u * unit) unit = 1u;                               // rounding to rounds must stay below ~1.5 % of a share
  auto cum_of = [&](uint32_t x) -> uint32_t {
    if (x == 0) return 0u;
    if (x >= 8) return ntl;
    const uint64_t c = p.bal ? (uint64_t)p.bal->cum[x] : (uint64_t)x * (XCC_ONE / 8u);
    uint32_t t = (uint32_t)(((uint64_t)ntl * c) >> 20);
    t = (t + unit / 2u) / unit * unit;
    return t < ntl ? t : ntl;
  };
  const uint32_t start_x = cum_of(pairs ? (xcd & ~1u) : xcd);
  const uint32_t cnt_x = cum_of(pairs ? (xcd | 1u) + 1u : xcd + 1u) - start_x;
  const uint32_t nvirt = cnt_x * nqt_l;
  if (j >= nvirt) {
    if (!FIRST && (threadIdx.x & 63) == 0) p.rec_cnt[b * 8 + (threadIdx.x >> 6)] = 0;
    return;
  }
  const uint32_t my_tiles = (nvirt - j + nwg - 1) / nwg;
  const uint32_t KSL = (uint32_t)p.nslices;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = w & 3;
  const int l15 = lane & 15, lq = lane >> 4;

  auto run = [&](auto grp_tag) {
    constexpr int GRP = decltype(grp_tag)::value;              // wave group = gallery half; 0 streams A, 1 streams B
    constexpr int MY_SLOTS = GRP == 0 ? A_SLOTS : BSL;
    constexpr bool dbg_nodma = (DBG & 1) || ((DBG & 32) && GRP == 0) || ((DBG & 64) && GRP == 1);
    // per-wave threshold words of its 64 queries: [0..63] threshold (f32) -- or, ladder on, the packed pair of thresholds --,
    // [64..127] ladder counters (refreshed by a 256-byte DMA per tile), [128..191] ladder count levels t_c
    float* thr_w = reinterpret_cast<float*>(smem + RING_BYTES + STAGE_BYTES) + w * THR_WORDS;
    const bool lad = !FIRST && !REPAIR && p.lad_k > 0 && p.st.lad_cnt != nullptr;
    const uint32_t* lad_cnt_src = p.st.lad_cnt;                 // + query of this lane, set with the thresholds
    uint32_t thr_qt = 0xFFFFFFFFu;
    auto load_thresholds = [&](uint32_t qt) {                   // plain loads: drains the DMA rings (query tile changes only)
      const uint32_t q = qt * TILE + wc * 64 + lane;
      if (lad) {
        reinterpret_cast<uint32_t*>(thr_w)[lane] = p.st.lad_pack[q];
        reinterpret_cast<uint32_t*>(thr_w)[64 + lane] = p.st.lad_cnt[q];
        thr_w[128 + lane] = p.st.lad_tc[q];
      } else {
        thr_w[lane] = p.st.thr[q];
      }
      thr_qt = qt;
    };
    // ladder: this wave's 64 counters, straight into LDS (one 4-byte-per-lane DMA piece in the wave's vmcnt order)
    auto refresh_counts = [&](uint32_t qt) {
      if (lad && qt == thr_qt)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(lad_cnt_src + qt * TILE + wc * 64 + lane),
                                         (LDS_AS void*)(thr_w + 64), 4, 0, 0);
    };
    auto tile_of = [&](uint32_t i, uint32_t& gt, uint32_t& qt) {
      const uint32_t v = j + i * nwg;
      qt = qt0 + v % nqt_l;
      gt = (uint32_t)p.tile0 + start_x + v / nqt_l;
    };
    // ---- DMA stream of this group's operand: wave-uniform scalar base + one constant per-lane offset (saddr form)
    uint32_t pf_i = 0, pf_sl = 0;
    const char* pf;
    const char* pfq = nullptr;                                // DBG 32768: this wave's query fragments (both groups)
    const uint32_t pf_lane = (uint32_t)lane * 16u;
    auto pf_set = [&](uint32_t i) {
      uint32_t gt, qt;
      tile_of(i < my_tiles ? i : my_tiles - 1, gt, qt);       // past the end: harmless re-load of the last tile
      pf = (GRP == 0 ? (const char*)p.gal_img + (int64_t)gt * KSL * SLICE_BYTES
                     : (const char*)p.qry_img + (int64_t)qt * KSL * SLICE_BYTES) + wc * 4096;
      if (DBG & 32768) pfq = (const char*)p.qry_img + (int64_t)qt * KSL * SLICE_BYTES + wc * 4096;
    };
    pf_set(0);
    const uint32_t ring_base = (GRP == 0 ? A_RING : B_RING) + wc * 4096;
    uint32_t wr_slot = 0;
    constexpr int AUX = GRP == 0 ? (POL & 0xFF) : ((POL >> 8) & 0xFF);
    constexpr int VM_PER_SLICE = ((DBG & 32768) ? 4 : 0) + (dbg_nodma ? 0 : 4);   // vector-memory operations a wave issues per slice
    u32x4 qdummy[4] = {};
    auto issue = [&]() {
      uint32_t off = pf_lane;
      asm volatile("" : "+v"(off));
      if (!dbg_nodma) {
        const GLOBAL_AS void* src = (const GLOBAL_AS void*)(pf + off);
        LDS_AS void* dst = (LDS_AS void*)(smem + ring_base + wr_slot * SLICE_BYTES);
        __builtin_amdgcn_global_load_lds(src, dst, 16, 0, AUX);
        __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, AUX);
        __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, AUX);
        __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, AUX);
      }
      if (DBG & 32768) {
        // the destinations are read-write operands of every statement, so they stay allocated for the whole loop; their
        // contents are never used (loads may land in any order relative to the next statement's issue)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                     : "+v"(qdummy[0]), "+v"(qdummy[1]), "+v"(qdummy[2]), "+v"(qdummy[3])
                     : "v"(off), "s"(pfq)
                     : "memory");
        pfq += SLICE_BYTES;
      }
      pf += SLICE_BYTES;
      if (++pf_sl == KSL) {
        pf_sl = 0;
        pf_set(++pf_i);
      }
      if (++wr_slot == MY_SLOTS) wr_slot = 0;
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t fsw = (0u - (uint32_t)(l15 >> 2)) & 3u;
    const uint32_t a_off = (uint32_t)(GRP * 128 + l15) * 64u + ((((uint32_t)lq) ^ fsw) << 4);
    uint32_t b_off = (uint32_t)B_RING + (uint32_t)(wc * 64 + l15) * 64u + ((((uint32_t)lq) ^ fsw) << 4);
    asm volatile("" : "+v"(b_off));     // opaque: keeps ONE address register for the four B reads (see structure 1)

    SurvRec* my_rec = p.rec + (uint64_t)(b * 8 + w) * p.rec_cap;
    uint32_t my_cnt = 0;
    float* sc_val = reinterpret_cast<float*>(smem + RING_BYTES + w * WAVE_SCRATCH);
    uint4* sc_meta = reinterpret_cast<uint4*>(smem + RING_BYTES + w * WAVE_SCRATCH + HIT_SLOTS * 32 * 4);
    const uint32_t sc_val_lds = lds_addr(sc_val), sc_meta_lds = lds_addr(sc_meta);

    // ---- per-tile filter of the accumulators (+ reset); same code as structure 1
    unsigned long long fdbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // DBG 2048: cycles in decide / write burst / read wait / emit
    // decide step of the filter, in two pieces so that INTER builds can place them inside the last MFMA segment of a tile:
    // decide_fetch = this wave's thresholds of its 4 query blocks (LDS words; ladder: packed pair + live counter),
    // decide_block(nb) = maximum of the lane's 32 scores of query block nb against its threshold -> ballot hm[nb]
    float thr4[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned long long hm[4] = {0ull, 0ull, 0ull, 0ull};
    auto decide_fetch = [&]() {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        if (lad) {
          // the tighter threshold t_c - margin once K rows with approx >= t_c have been counted (by any wave of the
          // launch: the counter is a fact about rows already scored, so it is a rigorous bound whenever it is read)
          const uint32_t pk = reinterpret_cast<const uint32_t*>(thr_w)[nb * 16 + l15];
          // the counters are written by this wave's own DMA piece, issued a tile ago and long retired by the counted
          // vmcnt waits of the slices in between; read through asm so that the compiler does not order the read behind
          // ALL pending DMA with a vmcnt(0), which would drain the rings at every tile boundary
          uint32_t cnt;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cnt) : "v"(lds_addr(thr_w + 64 + nb * 16 + l15)) : "memory");
          thr4[nb] = __uint_as_float(cnt >= (uint32_t)p.lad_k ? (pk & 0xFFFF0000u) : (pk << 16));
        } else {
          thr4[nb] = thr_w[nb * 16 + l15];      // +inf for padded queries
        }
      }
    };
    auto decide_block = [&](int nb) {
      // gallery blocks in the order the snake issues their MFMAs (odd query blocks run 7 .. 0): inside the last MFMA segment
      // the first values read are then the oldest results
      const int first = (ORDER == 3 && (nb & 1)) ? 7 : 0;
      float m = acc[first][nb][0];
#pragma unroll
      for (int m2 = 0; m2 < 8; ++m2) {
        const int mb = (ORDER == 3 && (nb & 1)) ? 7 - m2 : m2;
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mb][nb][r]);
      }
      hm[nb] = __ballot(m >= thr4[nb]);
    };
    auto tile_epilogue = [&](uint32_t gt, uint32_t qt) {
      const uint32_t row_base = gt * TILE + GRP * 128 + lq * 4;          // + mb*16 + reg
      const uint32_t ql_base = qt * TILE + wc * 64 + l15;                // + nb*16
      if (DBG & 4) {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) asm volatile("" ::"v"(acc[mb][nb]));
      } else if (FIRST) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const uint32_t q = ql_base + nb * 16;
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
      } else if (DBG & 8192) {
        // Filter variant C (A/B, measured 0.3-1.3 % faster than the scratch filter below, but its list can overflow on
        // duplicate-heavy tiles, which the scratch filter handles in rounds: not the default): per 16 x 16 block the lane maximum of its 4 scores is compared with the query's threshold
        // (3 VALU + one wave-uniform branch per block); a block with hits appends (4 scores, threshold, packed position)
        // of its hit lanes to a compact per-wave LDS list -- 2 LDS writes per hit block instead of 9 per hit (lane, query
        // block) pair, and the write path of the LDS is what the scratch variant waits for.  One scan of the list at the
        // end of the tile emits the records.
        constexpr uint32_t ENT = WAVE_SCRATCH / 24;                 // 48 entries: 16 B of scores + 8 B (threshold, position)
        if (qt != thr_qt) load_thresholds(qt);
        const uint32_t vals_lds = sc_val_lds, meta_lds = sc_val_lds + ENT * 16u;
        const uint32_t tile_row0 = gt * TILE + GRP * 128, tile_q0 = qt * TILE + wc * 64;
        const uint32_t pk_lane = (uint32_t)l15 | ((uint32_t)lq << 9);
        uint32_t ecnt = 0;                                          // entries in the list (wave-uniform)
        unsigned long long fc0 = 0;
        if (DBG & 2048) fc0 = stamp();
        auto emit_list = [&]() {
          unsigned long long fs0 = 0;
          if (DBG & 2048) { fs0 = stamp(); fdbg[4] += ecnt; fdbg[6] += 1; }
          const uint32_t nval = min(ecnt, ENT) * 4u;
          for (uint32_t v0 = 0; v0 < nval; v0 += 64) {
            const uint32_t vi = v0 + lane;
            const bool valid = vi < nval;
            float val;
            unsigned long long meta;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(val), "=&v"(meta)
                         : "v"(vals_lds + (valid ? vi : 0u) * 4u), "v"(meta_lds + (valid ? (vi >> 2) : 0u) * 8u)
                         : "memory");
            const uint32_t pk = (uint32_t)(meta >> 32);
            const uint32_t row = tile_row0 + ((pk >> 6) & 7u) * 16u + ((pk >> 9) & 3u) * 4u + (vi & 3u);
            const bool keep = valid && val >= __uint_as_float((uint32_t)meta) && row < (uint64_t)p.n;
            const unsigned long long km = __ballot(keep);
            if (km) {
              const uint32_t pos = my_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
              if (keep && pos < p.rec_cap)
                reinterpret_cast<uint4*>(my_rec)[pos] = make_uint4(__float_as_uint(val), row, tile_q0 + (pk & 63u), 0u);
              my_cnt += (uint32_t)__popcll(km);
            }
          }
          if (ecnt > ENT && lane == 0) atomicOr(p.st.flags, FLAG_REC_OVERFLOW);   // list overflow: the batch is answered again
          ecnt = 0;
          if (DBG & 2048) fdbg[3] += stamp() - fs0;
        };
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float thr = thr_w[nb * 16 + l15];                  // +inf for padded queries
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) {
            // (no canonicalising v_max(x, x) in front: the scores are finite sums of finite products)
            float m3;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(acc[mb][nb][0]), "v"(acc[mb][nb][1]), "v"(acc[mb][nb][2]));
            const bool mine = (m3 >= thr) | (acc[mb][nb][3] >= thr);
            const unsigned long long hit = __ballot(mine);
            if (hit) {
              const uint32_t pos = ecnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0u));
              if (mine && pos < ENT) {
                lds_store16<0>(vals_lds + pos * 16u, acc[mb][nb]);
                const unsigned long long mt = (unsigned long long)__float_as_uint(thr) |
                                              ((unsigned long long)(pk_lane | (uint32_t)(nb * 16) | (uint32_t)(mb << 6)) << 32);
                asm volatile("ds_write_b64 %0, %1" ::"v"(meta_lds + pos * 8u), "v"(mt) : "memory");
              }
              ecnt += (uint32_t)__popcll(hit);
            }
          }
          if (nb < 3 && ecnt >= ENT / 2) emit_list();              // keep room for the next query block's hits
        }
        if (ecnt) emit_list();
        __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): see structure 1
        if (DBG & 2048) { fdbg[0] += stamp() - fc0; fdbg[5] += 1; }
      } else if (DBG & 4096) {
        // Filter variant B (A/B): no LDS.  Every accumulator is compared with its query's threshold (one v_cmp into a
        // scalar lane mask each); the four masks of one 16 x 16 block are OR-ed and ONE wave-uniform branch per block
        // skips the append, which runs for ~11 of the 128 blocks of a tile.
        if (qt != thr_qt) load_thresholds(qt);
        unsigned long long fb0 = 0;
        if (DBG & 2048) fb0 = stamp();
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float thr = thr_w[nb * 16 + l15];
          const uint32_t q = ql_base + nb * 16;
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) {
            unsigned long long m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = __ballot(acc[mb][nb][r] >= thr);
            if (__builtin_expect(((m[0] | m[1]) | (m[2] | m[3])) != 0ull, 0)) {      // cold: laid out out of line
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const uint32_t row = row_base + mb * 16 + r;
                const bool keep = ((m[r] >> lane) & 1ull) && row < (uint64_t)p.n;
                const unsigned long long km = __ballot(keep);
                if (km) {
                  const uint32_t pos = my_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                  if (!(DBG & 1024) && keep && pos < p.rec_cap)
                    reinterpret_cast<uint4*>(my_rec)[pos] = make_uint4(__float_as_uint(acc[mb][nb][r]), row, q, 0u);
                  if (DBG & 1024) asm volatile("" ::"v"(pos), "v"(row));
                  my_cnt += (uint32_t)__popcll(km);
                }
              }
              if (DBG & 2048) fdbg[4] += 1;
            }
          }
        }
        if (DBG & 2048) { fdbg[0] += stamp() - fb0; fdbg[5] += 1; }
      } else {
        unsigned long long f0 = 0, f1 = 0, f2 = 0, f3 = 0;
        if (DBG & 2048) f0 = stamp();
        if (!INTER) {
          if (qt != thr_qt) load_thresholds(qt);
          decide_fetch();
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) decide_block(nb);
        }
        uint32_t base[5];
        base[0] = 0;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) base[nb + 1] = base[nb] + (uint32_t)__popcll(hm[nb]);
        const uint32_t total = (DBG & 512) ? 0u : base[4];            // DBG 512: decide only, no hit path (diagnostics)
        if (DBG & 512) asm volatile("" ::"s"(base[4]));
        if (DBG & 2048) { f1 = stamp(); fdbg[0] += f1 - f0; fdbg[4] += total; fdbg[5] += 1; }
        for (uint32_t r0 = 0; r0 < total; r0 += HIT_SLOTS) {          // almost always zero or one round
          if (DBG & 2048) f1 = stamp();
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            if (hm[nb] == 0) continue;                                  // wave-uniform
            const uint32_t rank = base[nb] + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm[nb] >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)hm[nb], 0u));
            const bool mine = (hm[nb] >> lane) & 1ull;
            if (mine && rank >= r0 && rank < r0 + HIT_SLOTS) {
              const uint32_t dst = sc_val_lds + (rank - r0) * 128;
              lds_store16<0>(dst, acc[0][nb]);
              lds_store16<16>(dst, acc[1][nb]);
              lds_store16<32>(dst, acc[2][nb]);
              lds_store16<48>(dst, acc[3][nb]);
              lds_store16<64>(dst, acc[4][nb]);
              lds_store16<80>(dst, acc[5][nb]);
              lds_store16<96>(dst, acc[6][nb]);
              lds_store16<112>(dst, acc[7][nb]);
              lds_store16u(sc_meta_lds + (rank - r0) * 16,
                           (u32x4){__float_as_uint(thr4[nb]), ql_base + nb * 16, row_base,
                                   lad ? __float_as_uint(thr_w[128 + nb * 16 + l15]) : 0x7F800000u});
            }
          }
          const uint32_t nslots = min(total - r0, (uint32_t)HIT_SLOTS);
          if (DBG & 2048) { f2 = stamp(); fdbg[1] += f2 - f1; }
          constexpr int SCAN = HIT_SLOTS * 32 / 64;
          u32x4 mt[SCAN];                                              // (threshold, query, row base, ladder level t_c)
          float vv[SCAN];
#pragma unroll
          for (int it = 0; it < SCAN; ++it) {
            const uint32_t e = it * 64 + lane;
            const bool valid = e < nslots * 32;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %3"
                         : "=&v"(mt[it]), "=v"(vv[it])
                         : "v"(sc_meta_lds + (valid ? (e >> 5) : 0u) * 16u), "v"(sc_val_lds + (valid ? e : 0u) * 4u)
                         : "memory");
          }
          static_assert(SCAN == 4, "the wait below lists the scan registers explicitly");
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(mt[0]), "+v"(mt[1]), "+v"(mt[2]), "+v"(mt[3]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3])
                       :
                       : "memory");
          if (DBG & 2048) { f3 = stamp(); fdbg[2] += f3 - f2; }
#pragma unroll
          for (int it = 0; it < SCAN; ++it) {
            if ((uint32_t)(it * 64) >= nslots * 32) break;              // wave-uniform
            const uint32_t e = it * 64 + lane;
            const uint32_t i = e & 31u;
            const uint32_t row = mt[it].z + (i >> 2) * 16 + (i & 3u);
            const bool keep = e < nslots * 32 && vv[it] >= __uint_as_float(mt[it].x) && row < (uint64_t)p.n;
            const unsigned long long km = __ballot(keep);
            if (km) {
              const uint32_t pos = my_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
              if (!(DBG & 1024) && keep) {                           // DBG 1024: no record stores (diagnostics)
                if (__builtin_expect(pos < p.rec_cap, 1)) reinterpret_cast<uint4*>(my_rec)[pos] = make_uint4(__float_as_uint(vv[it]), row, mt[it].y, 0u);
                else spill_record(p.st, vv[it], row, mt[it].y);
              }
              if (lad && keep && vv[it] >= __uint_as_float(mt[it].w)) atomicAdd(&p.st.lad_cnt[mt[it].y], 1u);
              if (DBG & 1024) asm volatile("" ::"v"(pos), "v"(row));
              my_cnt += (uint32_t)__popcll(km);
            }
          }
          if (DBG & 2048) { fdbg[3] += stamp() - f3; fdbg[6] += 1; }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): see structure 1
      }
      if (!ZC) {                                // ZC: the next tile's first MFMAs take the constant 0 as their C operand
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };

    // ---- prologue: all but one slot of this group's ring in flight, slice 0 landed
#pragma unroll
    for (int d = 0; d < MY_SLOTS - 1; ++d) issue();
    vm_wait<(MY_SLOTS - 2) * VM_PER_SLICE>();
    __builtin_amdgcn_s_barrier();
    if (GRP == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave group by one barrier

    uint32_t a_rd = 0, b_rd = 0;                           // ring slots holding the current slice (query ring: BSL slots)
    unsigned long long clk0 = 0, rt0 = 0;
    frag_t bkeep[4] = {};                                  // DBG 16384: the query fragments of the launch's first slice
    // in-kernel clock of every launch (s_memtime / s_memrealtime around the loop, per wave): two scalar reads, and the
    // number bench.py reports next to the roofline fraction (the chip holds 1.4-1.7 GHz of its 2.4 GHz under this load)
    if (p.dbg) { clk0 = stamp(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    // LOAD segment (the partner group is in its MFMA segment): 12 fragment reads, then the 4 DMA pieces of the slice
    // MY_SLOTS - 1 ahead into the slot whose reads retired before the barrier behind us
    auto frag_reads = [&](frag_t (&af)[8], frag_t (&bfr)[4], const char* abase, const char* bbase, bool first_ever) {
      if (!(DBG & 128) || first_ever) {
        if (!(DBG & 16384)) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) bfr[nb] = *reinterpret_cast<const frag_t*>(bbase + b_off + nb * 1024);
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) af[mb] = *reinterpret_cast<const frag_t*>(abase + a_off + mb * 1024);
      }
      if (DBG & 16384) {
        if (first_ever) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) bkeep[nb] = *reinterpret_cast<const frag_t*>(bbase + b_off + nb * 1024);
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          asm volatile("" : "+v"(bkeep[nb]));
          bfr[nb] = bkeep[nb];
        }
      }
      if (DBG & 128) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+v"(bfr[nb]));
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) asm volatile("" : "+v"(af[mb]));
      }
    };
    auto load_segment = [&](frag_t (&af)[8], frag_t (&bfr)[4], bool first_ever) {
      const char* abase = smem + a_rd * SLICE_BYTES;
      const char* bbase = smem + b_rd * SLICE_BYTES;
      if (++a_rd == A_SLOTS) a_rd = 0;
      if (++b_rd == BSL) b_rd = 0;
      frag_reads(af, bfr, abase, bbase, first_ever);
      __builtin_amdgcn_sched_barrier(0);
      issue();
      if (GRP == 1) vm_wait<(BSL - 2) * VM_PER_SLICE>();   // B(S+1) landed before the barrier that opens group 0's LOAD(S+1)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads retired BEFORE the barrier: frees the slots (WAR)
      __builtin_amdgcn_sched_barrier(0);
    };
    // POS: 0 = middle slice, 1 = first slice of a tile (ZC: accumulate onto the constant 0), 2 = last slice of a tile
    // (INTER: the filter's decide step of query block nb follows the MFMAs of block nb + 1, i.e. it issues in their gaps)
    auto one_mfma = [&](frag_t a, frag_t b, f32x4 c) -> f32x4 {
      if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    };
    auto mfma_segment = [&](frag_t (&af)[8], frag_t (&bfr)[4], auto pos_tag) {
      constexpr int POS = decltype(pos_tag)::value;
      constexpr bool zero_c = ZC && POS == 1;
      constexpr bool inter = INTER && POS == 2;
      const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (!(DBG & 2)) {
        if (ORDER == 0) {
#pragma unroll
          for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = one_mfma(af[mb], bfr[nb], zero_c ? zero4 : acc[mb][nb]);
        } else if (ORDER == 1) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) acc[mb][nb] = one_mfma(af[mb], bfr[nb], zero_c ? zero4 : acc[mb][nb]);
        } else if (ORDER == 2) {        // diagnostics: operands swapped (C transposed: results invalid with the filter)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) acc[mb][nb] = one_mfma(bfr[nb], af[mb], zero_c ? zero4 : acc[mb][nb]);
        } else {                        // snake order (every MFMA shares an operand with its predecessor)
          if (inter) decide_fetch();
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int m2 = 0; m2 < 8; ++m2) {
              const int mb = (nb & 1) ? 7 - m2 : m2;
              acc[mb][nb] = one_mfma(af[mb], bfr[nb], zero_c ? zero4 : acc[mb][nb]);
            }
            if (inter && nb > 0) {
              // the maxima of query block nb - 1 (its MFMAs were issued >= 8 instructions ago: no result-latency stall) ride
              // in the issue gaps of block nb's MFMAs: one matrix instruction, then up to three vector instructions
              decide_block(nb - 1);
#pragma unroll
              for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
              }
            }
            if (inter) __builtin_amdgcn_sched_barrier(0);
          }
          if (inter) decide_block(3);
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) asm volatile("" ::"v"(af[mb]));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) asm volatile("" ::"v"(bfr[nb]));
      }
      if (GRP == 0) vm_wait<(A_SLOTS - 2) * VM_PER_SLICE>();   // A(S+1) landed, A(S+2..S+4) may be in flight
      __builtin_amdgcn_sched_barrier(0);
    };
    using pos_mid = std::integral_constant<int, 0>;
    using pos_first = std::integral_constant<int, 1>;
    using pos_last = std::integral_constant<int, 2>;

    uint32_t gt, qt, prev_gt = 0, prev_qt = 0;
    for (uint32_t i = 0; i < my_tiles; ++i) {
      tile_of(i, gt, qt);
      // ---- slice 0 of the tile (peeled: group 0 filters the previous tile between its barrier and its MFMAs).  Group 0
      // reads the fragments of this slice AFTER that filter (their slots are not refilled before the next LOAD segment),
      // so no fragment register is live across the filter: the kernel's VGPR peak is the loop's, not loop + filter.
      {
        frag_t af[8], bfr[4];
        refresh_counts(qt);
        if constexpr (GRP == 0) {
          const char* abase = smem + a_rd * SLICE_BYTES;
          const char* bbase = smem + b_rd * SLICE_BYTES;
          if (++a_rd == A_SLOTS) a_rd = 0;
          if (++b_rd == BSL) b_rd = 0;
          __builtin_amdgcn_sched_barrier(0);
          issue();
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if (i > 0) tile_epilogue(prev_gt, prev_qt);
          __builtin_amdgcn_sched_barrier(0);
          frag_reads(af, bfr, abase, bbase, i == 0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        } else {
          load_segment(af, bfr, i == 0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
        mfma_segment(af, bfr, pos_first{});
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- slices 1 .. KSL-1 (INTER: .. KSL-2): straight-line body
#pragma unroll 1
      for (uint32_t sl = 1; sl < KSL - (INTER ? 1u : 0u); ++sl) {
        frag_t af[8], bfr[4];
        load_segment(af, bfr, false);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfma_segment(af, bfr, pos_mid{});
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (INTER) {
        // ---- last slice of the tile (peeled): its MFMA segment also computes the filter's decide step (KSL >= 2: dp is a
        // multiple of 64).  The thresholds of the tile's query block must be this wave's before the segment reads them.
        frag_t af[8], bfr[4];
        if (qt != thr_qt) load_thresholds(qt);
        load_segment(af, bfr, false);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfma_segment(af, bfr, pos_last{});
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (GRP == 1) tile_epilogue(gt, qt);
      else { prev_gt = gt; prev_qt = qt; }
    }
    if (GRP == 0) {
      tile_epilogue(prev_gt, prev_qt);
      __builtin_amdgcn_s_barrier();                        // balance the stagger barrier
    }
    if (DBG & 32768) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(qdummy[0]), "+v"(qdummy[1]), "+v"(qdummy[2]), "+v"(qdummy[3])::"memory");
    }
    if (p.dbg && lane == 0) {
      unsigned long long* dbgp = p.dbg + (uint64_t)(b * 8 + w) * 8;
      dbgp[3] = my_cnt;                                                              // records this wave emitted
      dbgp[4] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF;   // HW_REG_XCC_ID[3:0]
      dbgp[5] = (unsigned long long)my_tiles * KSL;
      if (DBG & 2048) {                      // filter stamps in slots 0..3, hits << 32 | tiles in 4... (diagnostics)
        dbgp[0] = fdbg[0]; dbgp[1] = fdbg[1]; dbgp[2] = fdbg[2]; dbgp[3] = fdbg[3];
        dbgp[4] = (fdbg[4] << 40) | (fdbg[6] << 20) | fdbg[5];
      }
      dbgp[6] = stamp() - clk0;
      dbgp[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    if (!FIRST && lane == 0) {
      p.rec_cnt[b * 8 + w] = my_cnt < p.rec_cap ? my_cnt : p.rec_cap;
      // (a full segment is no error any more: the records beyond it went straight into their queries' buckets, spill_record.
      // The filter variants of -DMI_KBENCH builds have no spill path and still flag.)
      if ((DBG & (4096 | 8192)) && my_cnt > p.rec_cap) atomicOr(p.st.flags, FLAG_REC_OVERFLOW);
    }
  };
  if (w < 4) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trailing (unused) DMA pieces land before the LDS is released
}

static unsigned persistent_grid() {
  const unsigned g = (unsigned)(current_device_cus() / 8) * 8u;   // one workgroup per CU, whole XCDs
  return g < 8 ? 8 : g;
}

unsigned gemm_select_grid() { return persistent_grid(); }

// One workgroup per wave-private record segment.  The records of a segment belong to few queries (the 64 of the wave
// that wrote it), about ten records each, so positions are handed out in two levels: an LDS counter per query gives the
// rank inside the segment, ONE global atomic per (segment, query) reserves the range in the query's bucket -- a tenth
// of the global atomics of a per-record scheme, which ran at the L2's atomic rate (~80 us for 1.4 M records).
constexpr int SCATTER_THREADS = 256;
constexpr int SCATTER_PER_THREAD = 16;                     // covers rec_cap = 4096 records per segment
__global__ __launch_bounds__(SCATTER_THREADS) void scatter_records_kernel(const SurvRec* __restrict__ rec,
                                                                          const uint32_t* __restrict__ rec_cnt,
                                                                          uint32_t rec_cap, QueryState st,
                                                                          const uint32_t* __restrict__ cond,
                                                                          XccBalance* __restrict__ bal,
                                                                          const unsigned long long* __restrict__ dbg,
                                                                          uint32_t nseg, uint32_t group) {
  __shared__ uint32_t hist[1024];                          // per query of the batch (QB = 1024): count, then base
  if (cond && *cond == 0) return;
  if (bal && blockIdx.x == 0) {
    // XCD shares for the next launch from the loop times of the one that has just finished: speed of label x =
    // slices done by its workgroups / time of its slowest workgroup; new share = half old, half measured
    uint32_t* tmax = hist;                                 // [8]
    uint32_t* work = hist + 8;                             // [8]
    if (threadIdx.x < 16) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t nblk = nseg / 8;
    for (uint32_t b = threadIdx.x; b < nblk; b += SCATTER_THREADS) {
      const unsigned long long* d = dbg + (uint64_t)b * 8 * 8;
      atomicMax(&tmax[b & 7u], (uint32_t)d[7]);
      atomicAdd(&work[b & 7u], (uint32_t)d[5]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float sp[8], tot = 0.f;
      bool ok = true;
      for (int x = 0; x < 8; ++x) {
        ok &= tmax[x] > 0 && work[x] > 0;
        sp[x] = ok ? (float)work[x] / (float)tmax[x] : 0.f;
        tot += sp[x];
      }
      if (ok) {
        float wsum = 0.f, w[8];
        for (int x = 0; x < 8; ++x) {
          w[x] = 0.5f * bal->w[x] + 0.5f * sp[x] / tot;
          wsum += w[x];
        }
        float c = 0.f;
        for (int x = 0; x < 8; ++x) {
          bal->w[x] = w[x] / wsum;
          bal->cum[x] = (uint32_t)(c * (float)XCC_ONE + 0.5f);
          c += w[x] / wsum;
        }
        bal->cum[8] = XCC_ONE;
        bal->launches += 1;
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 16; i += SCATTER_THREADS) hist[i] = 0;
    __syncthreads();
  }
  // `group` consecutive segments (the 8 waves of one scoring workgroup when the batch is small) form ONE list for this
  // workgroup: with few queries every segment holds records of the SAME queries, and one global atomic per (segment, query)
  // would put 2048 adds on each query's counter -- 15-25 us of same-address atomics at 1..128 queries (scripts/timeline.sh)
  __shared__ uint32_t pre[9];                              // prefix of the group's segment counts
  const uint32_t seg0 = blockIdx.x * group;
  const uint32_t CAP = (uint32_t)(SCATTER_THREADS * SCATTER_PER_THREAD);
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t g = 0; g < group; ++g) {
      pre[g] = acc;
      acc += (seg0 + g < nseg) ? min(rec_cnt[seg0 + g], CAP) : 0u;
    }
    pre[group] = acc;
  }
  __syncthreads();
  const uint32_t total = pre[group];
  if (total == 0) return;
  for (uint32_t base = 0; base < total; base += CAP) {     // one round unless the group holds more than 4096 records
    const uint32_t n = min(total - base, CAP);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 1024; i += SCATTER_THREADS) hist[i] = 0;
    __syncthreads();
    SurvRec e[SCATTER_PER_THREAD];
    uint32_t rank[SCATTER_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SCATTER_PER_THREAD; ++j) {
      const uint32_t i = threadIdx.x + j * SCATTER_THREADS;
      if (i < n) {
        const uint32_t v = base + i;
        uint32_t g = 0;
        while (g + 1 < group && v >= pre[g + 1]) ++g;
        e[j] = rec[(uint64_t)(seg0 + g) * rec_cap + (v - pre[g])];
        rank[j] = atomicAdd(&hist[e[j].q & 1023u], 1u);
      }
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < 1024; q += SCATTER_THREADS) {
      const uint32_t c = hist[q];
      if (c) hist[q] = atomicAdd(&st.cnt[q * CNT_STRIDE], c);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SCATTER_PER_THREAD; ++j) {
      const uint32_t i = threadIdx.x + j * SCATTER_THREADS;
      if (i < n) {
        const uint32_t pos = hist[e[j].q & 1023u] + rank[j];
        if (pos < st.cap) st.surv[(uint64_t)e[j].q * st.cap + pos] = pack_entry(e[j].score, e[j].row);
        else atomicOr(st.flags, FLAG_SURV_OVERFLOW);
      }
    }
  }
}

void launch_scatter_records(const SurvRec* rec, const uint32_t* rec_cnt, uint32_t rec_cap, uint32_t nseg,
                            QueryState st, const uint32_t* cond, hipStream_t stream, XccBalance* bal,
                            const unsigned long long* dbg, uint32_t ntiles, int32_t nq) {
  // shares are only re-measured on launches with enough tiles per XCD for the loop time to be a speed
  if (cond || !dbg || ntiles < 8u * 64u) bal = nullptr;
  // batches of <= 128 queries: one workgroup per scoring workgroup (its 8 wave segments as one list), see the kernel
  const uint32_t group = (nq > 0 && nq <= STREAM_MAX_QUERIES) ? 8u : 1u;
  hipLaunchKernelGGL(scatter_records_kernel, dim3((nseg + group - 1) / group), dim3(SCATTER_THREADS), 0, stream, rec, rec_cnt,
                     rec_cap, st, cond, bal, dbg, nseg, group);
}

void init_xcc_balance_host(XccBalance* h) {
  for (int x = 0; x < 8; ++x) {
    h->cum[x] = (uint32_t)x * (XCC_ONE / 8u);
    h->w[x] = 0.125f;
  }
  h->cum[8] = XCC_ONE;
  h->launches = 0;
}

// shares remembered from earlier launches (the prepared-gallery file, another handle of this process on the same device): the
// first launch then starts from them instead of from an even split.  Nonsense input falls back to the even split.
void init_xcc_balance_from(XccBalance* h, const float* w8) {
  float sum = 0.f;
  bool ok = true;
  for (int x = 0; x < 8; ++x) {
    ok &= w8[x] == w8[x] && w8[x] > 0.02f && w8[x] < 0.5f;
    sum += w8[x];
  }
  if (!ok || !(sum > 0.5f && sum < 2.0f)) return init_xcc_balance_host(h);
  float c = 0.f;
  for (int x = 0; x < 8; ++x) {
    h->w[x] = w8[x] / sum;
    h->cum[x] = (uint32_t)(c * (float)XCC_ONE + 0.5f);
    c += w8[x] / sum;
  }
  h->cum[8] = XCC_ONE;
  h->launches = 0;
}

// The product build holds EIGHT instantiations of the tile kernel -- {bootstrap, filtered launch, filtered launch with nt gallery
// pieces (one query tile), conditional repair launch} x {fp16, bf16 image}, structure 2, snake order -- and nothing else.  Every diagnostic (DBG != 0: stages switched off, stamps),
// A/B (ORDER, OPT, POL, the paired-XCD walk) and structure-1 instantiation is compiled under -DMI_KBENCH only, i.e. into
// scripts/kbench.hip's own program (scripts/kbench_build.sh), whose records are under profiles/ (r04a_kbench*, r04c_kbench_rotated,
// r04m_kbench*): several of them return wrong answers by design and none is reachable from the C ABI.
#ifdef MI_KBENCH
template <bool FIRST, int DBG, bool F16, bool REPAIR = false>
static void launch_variant(const ScoreArgs& a, size_t lds, hipStream_t stream) {
  ensure_dynamic_lds((const void*)gemm_select_kernel<FIRST, DBG, F16, REPAIR>);
  hipEvent_t e0, e1;
  take_launch_events(&e0, &e1);
  if (e0 && e1)
    hipExtLaunchKernelGGL((gemm_select_kernel<FIRST, DBG, F16, REPAIR>), dim3(persistent_grid()), dim3(512), lds, stream, e0, e1,
                          0, a);
  else
    hipLaunchKernelGGL((gemm_select_kernel<FIRST, DBG, F16, REPAIR>), dim3(persistent_grid()), dim3(512), lds, stream, a);
}
#endif

template <bool FIRST, int DBG, bool F16, bool REPAIR, int ORDER, int OPT = 0, int POL = 0>
static void launch_tile(const ScoreArgs& a, size_t lds, hipStream_t stream) {
  ensure_dynamic_lds((const void*)gemm_tile_kernel<FIRST, DBG, F16, REPAIR, ORDER, OPT, POL>);
  hipEvent_t e0, e1;
  take_launch_events(&e0, &e1);
  if (e0 && e1)
    hipExtLaunchKernelGGL((gemm_tile_kernel<FIRST, DBG, F16, REPAIR, ORDER, OPT, POL>), dim3(persistent_grid()), dim3(512),
                          lds, stream, e0, e1, 0, a);
  else
    hipLaunchKernelGGL((gemm_tile_kernel<FIRST, DBG, F16, REPAIR, ORDER, OPT, POL>), dim3(persistent_grid()), dim3(512), lds,
                       stream, a);
}

#ifdef MI_KBENCH
// scripts/kbench.hip only: a.debug / a.variant select a diagnostic or A/B instantiation; false = none matches (product kernel)
static bool launch_kbench_variant(ScoreArgs& a, bool first, size_t lds, hipStream_t stream) {
  if (a.variant == 20 || a.variant == 21) a.walk = 1;           // paired-XCD walk (A/B), 21: with nt gallery pieces
  if (a.variant != 1) {
    if (a.cond || first || !a.img_f16) return false;
    switch (a.debug) {
      case 4:
        if (a.variant == 2) return launch_tile<false, 4, true, false, 0>(a, lds, stream), true;
        if (a.variant == 3) return launch_tile<false, 4, true, false, 2>(a, lds, stream), true;
        if (a.variant == 4) return launch_tile<false, 4, true, false, 1>(a, lds, stream), true;
        return launch_tile<false, 4, true, false, 3>(a, lds, stream), true;
      case 5: return launch_tile<false, 5, true, false, 3>(a, lds, stream), true;
      case 512: return launch_tile<false, 512, true, false, 3>(a, lds, stream), true;
      case 1024: return launch_tile<false, 1024, true, false, 3>(a, lds, stream), true;
      case 2048:
        if (a.variant == 6) return launch_tile<false, 2048, true, false, 3, 3>(a, lds, stream), true;
        return launch_tile<false, 2048, true, false, 3>(a, lds, stream), true;
      case 8192: return launch_tile<false, 8192, true, false, 3>(a, lds, stream), true;
      case 8192 + 2048: return launch_tile<false, 8192 + 2048, true, false, 3>(a, lds, stream), true;
      case 4096: return launch_tile<false, 4096, true, false, 3>(a, lds, stream), true;
      case 4096 + 2048: return launch_tile<false, 4096 + 2048, true, false, 3>(a, lds, stream), true;
      case 4096 + 1024: return launch_tile<false, 4096 + 1024, true, false, 3>(a, lds, stream), true;
      case 5 + 128: return launch_tile<false, 5 + 128, true, false, 3>(a, lds, stream), true;
      // round 4: which operand's DMA costs what (gallery pieces / query pieces skipped), and the price list of a query
      // operand that bypasses LDS (query fragments read once; + the fragment-shaped loads into discarded registers)
      case 4 + 32: return launch_tile<false, 4 + 32, true, false, 3>(a, lds, stream), true;
      case 4 + 64: return launch_tile<false, 4 + 64, true, false, 3>(a, lds, stream), true;
      case 4 + 64 + 16384: return launch_tile<false, 4 + 64 + 16384, true, false, 3>(a, lds, stream), true;
      case 4 + 64 + 16384 + 32768: return launch_tile<false, 4 + 64 + 16384 + 32768, true, false, 3>(a, lds, stream), true;
      default:
        // round 4: cache policy of the DMA pieces (10 .. 18; gallery aux | query aux << 8) and the paired-XCD walk (20, 21)
        if (a.variant == 10 || a.variant == 21) return launch_tile<false, 0, true, false, 3, 0, 2>(a, lds, stream), true;   // gallery nt
        if (a.variant == 11) return launch_tile<false, 0, true, false, 3, 0, 16>(a, lds, stream), true;         // gallery sc1
        if (a.variant == 12) return launch_tile<false, 0, true, false, 3, 0, 17>(a, lds, stream), true;         // gallery sc0 sc1
        if (a.variant == 13) return launch_tile<false, 0, true, false, 3, 0, 2 | (2 << 8)>(a, lds, stream), true;   // both nt
        if (a.variant == 14) return launch_tile<false, 0, true, false, 3, 0, 2 << 8>(a, lds, stream), true;     // query nt (control)
        if (a.variant == 15) return launch_tile<false, 0, true, false, 3, 0, 18>(a, lds, stream), true;         // gallery nt sc1
        if (a.variant == 16) return launch_tile<false, 0, true, false, 3, 0, 1>(a, lds, stream), true;          // gallery sc0
        if (a.variant == 17) return launch_tile<false, 0, true, false, 3, 1, 1>(a, lds, stream), true;          // zero-C + gallery sc0
        if (a.variant == 18) return launch_tile<false, 0, true, false, 3, 1, 17>(a, lds, stream), true;         // zero-C + gallery sc0 sc1
        if (a.variant == 2) return launch_tile<false, 0, true, false, 0>(a, lds, stream), true;
        if (a.variant == 4) return launch_tile<false, 0, true, false, 1>(a, lds, stream), true;
        if (a.variant == 5) return launch_tile<false, 0, true, false, 3, 1>(a, lds, stream), true;   // zero-C first slice
        if (a.variant == 6) return launch_tile<false, 0, true, false, 3, 3>(a, lds, stream), true;   // + decide inside the last slice
        if (a.variant == 7) return launch_tile<false, 0, true, false, 3, 2>(a, lds, stream), true;   // decide inside the last slice only
        return false;
    }
  }
  // structure 1 (variant 1)
  if (a.cond) return (a.img_f16 ? launch_variant<false, 0, true, true>(a, lds, stream)
                                : launch_variant<false, 0, false, true>(a, lds, stream)), true;
  if (a.img_f16) {
    if (first) return launch_variant<true, 0, true>(a, lds, stream), true;
    switch (a.debug) {
      case 4: return launch_variant<false, 4, true>(a, lds, stream), true;
      case 5: return launch_variant<false, 5, true>(a, lds, stream), true;
      case 8: case 24: return launch_variant<false, 8, true>(a, lds, stream), true;
      case 4 + 32: return launch_variant<false, 4 + 32, true>(a, lds, stream), true;
      case 4 + 64: return launch_variant<false, 4 + 64, true>(a, lds, stream), true;
      case 4 + 128: return launch_variant<false, 4 + 128, true>(a, lds, stream), true;
      case 5 + 128: return launch_variant<false, 5 + 128, true>(a, lds, stream), true;
      case 5 + 128 + 256: return launch_variant<false, 5 + 128 + 256, true>(a, lds, stream), true;
      case 5 + 256: return launch_variant<false, 5 + 256, true>(a, lds, stream), true;
      default: return launch_variant<false, 0, true>(a, lds, stream), true;
    }
  }
  if (first) return launch_variant<true, 0, false>(a, lds, stream), true;
  return launch_variant<false, 0, false>(a, lds, stream), true;
}
#endif

void launch_gemm_select(const ScoreArgs& a_in, bool first, hipStream_t stream) {
  ScoreArgs a = a_in;
  if (stream_select_applies(a) || (first && stream_bootstrap_applies(a))) return launch_stream_select(a, first, stream);
  const size_t lds = (size_t)RING_BYTES + STAGE_BYTES + 8 * THR_WORDS * 4;      // 162,816 B of the 163,840
#ifdef MI_KBENCH
  if (launch_kbench_variant(a, first, lds, stream)) return;
#endif
  // structure 2, MFMA issue order 3 = query-block-major snake (every MFMA shares an operand with its predecessor; bit-identical
  // results, +0.8 % over plain query-block-major and +4 % over gallery-block-major by the clock the chip holds)
  if (a.cond) return a.img_f16 ? launch_tile<false, 0, true, true, 3>(a, lds, stream)
                               : launch_tile<false, 0, false, true, 3>(a, lds, stream);
  if (first) return a.img_f16 ? launch_tile<true, 0, true, false, 3>(a, lds, stream)
                              : launch_tile<true, 0, false, false, 3>(a, lds, stream);
  // ONE query tile (129 .. 256 queries; the filtered launch only): every gallery tile is read once, by one workgroup, and
  // the launch is paced by HBM, not by the matrix pipe -- its gallery pieces carry the nt policy like the streaming
  // kernel's: 1.038 -> 0.998 ms at 256 queries, 0.977 -> 0.92 at 129 (profiles/r06_one_tile_rings_ab.txt; giving the gallery
  // ring a sixth or seventh slot instead was measured in the same run: 6 + 3 slots 1.059 ms, 7 + 2 1.23 -- the query stream
  // needs its lead).  With more query tiles the second one re-reads the gallery tile from L2: default policy.
  // Round 6 also built a HALF-TILE mode for batches whose last query tile holds <= 128 queries (257 .. 384, ...: 4 x 2 waves of
  // 64 x 64 outputs, 16 MFMAs per wave and slice instead of 32 for the padding): bit-identical answers, and slower -- 384
  // queries 1.775 -> 1.96 ms, 640 2.59 -> 2.82 (profiles/r06g_half_tile_ab.txt; as two instantiations of the slice loop it
  // spilled 161 VGPRs, r06f_half_tile_first_attempt.txt).  A slice of 16 MFMAs per wave is shorter than the LDS round trip
  // and the two barriers that frame it; the half tile costs ~0.8 of a full one and the branches cost the full tiles more.
  if (a.nqt == 1) return a.img_f16 ? launch_tile<false, 0, true, false, 3, 0, 2>(a, lds, stream)
                                   : launch_tile<false, 0, false, false, 3, 0, 2>(a, lds, stream);
  return a.img_f16 ? launch_tile<false, 0, true, false, 3>(a, lds, stream)
                   : launch_tile<false, 0, false, false, 3>(a, lds, stream);
}

}  // namespace mi
