// Small-batch scoring kernel with fused survivor filter (gfx950 / CDNA4): Q <= 128 queries per launch.
//
// The reference's online stage scores ONE query image against the whole database and its test sets have 70
// queries (src/online.py:132-147, src/test_rOP1m.py:144-149).  With so few queries the contraction is bound by
// streaming the 16-bit gallery image from HBM (2 * D bytes per row), not by the matrix pipe: the 256 x 256-tile
// kernel of gemm_select.hip paces itself by 32 MFMAs per wave and slice whatever the batch, which caps it at
// ~5.1 TB/s.  This kernel keeps the gallery side of that design (tile-blocked image, 16 KiB K-slices moved by
// global_load_lds_dwordx4 into an LDS ring that never drains, counted vmcnt, raw s_barrier) and narrows the
// query side to NQB blocks of 16 queries:
//  * persistent grid, one 512-thread workgroup per CU; workgroup b owns gallery tiles b, b + grid, ...;
//  * 8 waves x 32 gallery rows, every wave against all NQB * 16 queries: 2 x NQB MFMAs (16x16x32) per slice;
//  * ONE ring of 6 slots, 5 slices (80 KiB of gallery per CU) in flight, one barrier per slice: the barrier that
//    publishes slice S also retires every wave's reads of slice S-1, whose slot the next DMA pieces then reuse;
//  * per slice a wave issues 2 gallery pieces (1 KiB each) and, waves 0..NQB-1, one query piece;
//  * the filter is the one of the big kernel without its LDS scratch (the matrix pipe is idle most of the time
//    here): ballot/popcount positions into wave-private record segments, no returning atomics.
// Records, thresholds and flags are shared with gemm_select.hip (scatter_records_kernel runs afterwards).
#include <hip/hip_ext.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace mi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

constexpr int SS_DEPTH = 6;                                   // ring slots; SS_DEPTH - 1 slices in flight
constexpr int GAL_AUX = 2;                                    // gallery pieces are read once per launch: nt (aux = 2)

// MODE 0: filtered launch, 1: conditional repair launch (own instantiation, see gemm_select.hip), 2: bootstrap
// launch on the sample image (every score stored, slot = sample row)
template <int NQB, bool F16, int MODE>
__global__ __launch_bounds__(512) void stream_select_kernel(ScoreArgs p) {
  using frag_t = typename std::conditional<F16, f16x8, bf16x8>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BQ_BYTES = NQB * 1024;                        // NQB * 16 query rows of 64 B per slice
  constexpr int B_RING0 = SS_DEPTH * SLICE_BYTES;
  constexpr int THR0 = B_RING0 + SS_DEPTH * BQ_BYTES;
  constexpr bool REPAIR = MODE == 1, FIRST = MODE == 2;
  if (REPAIR && *p.cond == 0) return;
  // bootstrap launches are not persistent: one workgroup per (tile, group of NQB * 16 queries), so that a 1024-query
  // batch spreads its 32 sample tiles over 256 CUs instead of 128 tile-kernel workgroups of a full tile time each
  const uint32_t ntiles = (uint32_t)p.ntiles;
  // bootstrap, small batches: KS workgroups share one (tile, query group), each a contiguous range of K-slices (ksplit)
  const uint32_t KS = FIRST ? (uint32_t)p.ksplit : 1u;
  const uint32_t ks = FIRST ? blockIdx.x % KS : 0u, bid = FIRST ? blockIdx.x / KS : blockIdx.x;
  const uint32_t qg = FIRST ? bid / ntiles : 0u;
  const uint32_t b = FIRST ? bid % ntiles : bid, nwg = FIRST ? ntiles : gridDim.x;
  const uint32_t q0 = qg * NQB * 16;                          // first query of this workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (b >= ntiles) {
    if (!FIRST && lane == 0) p.rec_cnt[b * 8 + w] = 0;
    return;
  }
  const uint32_t my_tiles = (ntiles - b + nwg - 1) / nwg;
  const uint32_t KSL = (uint32_t)p.nslices;
  const uint32_t s0 = ks * (KSL / KS), s1 = FIRST ? s0 + KSL / KS : KSL;      // this workgroup's K-slices of a tile
  const uint32_t T_total = FIRST ? s1 - s0 : my_tiles * KSL;
  const int l15 = lane & 15, lq = lane >> 4;

  float* thr_s = reinterpret_cast<float*>(smem + THR0);       // thresholds of the NQB * 16 queries (+inf for padding)
  if (!FIRST && tid < NQB * 16) thr_s[tid] = p.st.thr[tid];   // published by the first barrier of the loop

  // ---- DMA stream: slice s of my i-th tile, continuous over tile boundaries
  const bool qloader = w < NQB;
  uint32_t pf_i = 0, pf_sl = s0, wr_slot = 0;
  // wave-uniform bases (scalar registers) + one per-lane byte offset: the DMA instructions take the saddr form and
  // the second gallery piece is the instruction's immediate offset (added to the global and the LDS address)
  const char* pfa;
  const char* pfq = reinterpret_cast<const char*>(p.qry_img) + (int64_t)(q0 / TILE) * KSL * SLICE_BYTES +
                    (q0 % TILE) * 64 + w * 1024;
  const uint32_t pf_lane = (uint32_t)lane * 16u;
  auto pf_set = [&](uint32_t i) {
    const uint32_t gt = (uint32_t)p.tile0 + b + (i < my_tiles ? i : my_tiles - 1) * nwg;   // past the end: reload
    pfa = reinterpret_cast<const char*>(p.gal_img) + (int64_t)gt * KSL * SLICE_BYTES + w * 2048;
  };
  pf_set(0);
  pfa += (int64_t)s0 * SLICE_BYTES;
  auto issue = [&]() {
    uint32_t off = pf_lane;
    asm volatile("" : "+v"(off));          // keeps the zero extension in this block (saddr form selection)
    LDS_AS void* dst = (LDS_AS void*)(smem + wr_slot * SLICE_BYTES + w * 2048);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(pfa + off), dst, 16, 0, GAL_AUX);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(pfa + off), dst, 16, 1024, GAL_AUX);
    if (qloader)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(pfq + (int64_t)pf_sl * SLICE_BYTES + off),
                                       (LDS_AS void*)(smem + B_RING0 + wr_slot * BQ_BYTES + w * 1024), 16, 0, 0);
    pfa += SLICE_BYTES;
    if (++pf_sl == KSL) {
      pf_sl = 0;
      pf_set(++pf_i);
    }
    if (++wr_slot == SS_DEPTH) wr_slot = 0;
  };

  f32x4 acc[2][NQB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NQB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment offsets inside a slot ([rows][32 k] 16-bit, 64-byte rows, chunk-swizzled like the big kernel's)
  const uint32_t fsw = (0u - (uint32_t)(l15 >> 2)) & 3u;
  const uint32_t a_off = (uint32_t)(w * 32 + l15) * 64u + ((((uint32_t)lq) ^ fsw) << 4);
  const uint32_t b_off = (uint32_t)B_RING0 + (uint32_t)l15 * 64u + ((((uint32_t)lq) ^ fsw) << 4);

  SurvRec* my_rec = p.rec + (uint64_t)(b * 8 + w) * p.rec_cap;
  uint32_t my_cnt = 0;

#pragma unroll
  for (int d = 0; d < SS_DEPTH - 1; ++d) issue();             // slices 0 .. DEPTH-2 in flight

  uint32_t rd = 0, cur_i = 0, cur_sl = s0;
  for (uint32_t S = 0; S < T_total; ++S) {
    // my pieces of slice S have landed when at most DEPTH-2 later slices of mine are outstanding
    if (qloader) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // slice S published; reads of slice S-1 retired
    const char* abase = smem + rd * SLICE_BYTES;
    const char* bbase = smem + rd * BQ_BYTES;
    if (++rd == SS_DEPTH) rd = 0;
    frag_t af[2], bfr[NQB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) af[mb] = *reinterpret_cast<const frag_t*>(abase + a_off + mb * 1024);
#pragma unroll
    for (int nb = 0; nb < NQB; ++nb) bfr[nb] = *reinterpret_cast<const frag_t*>(bbase + b_off + nb * 1024);
    __builtin_amdgcn_sched_barrier(0);
    issue();                                                  // slice S+DEPTH-1 into the slot of slice S-1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb)
        if constexpr (F16)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
        else
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);

    if (++cur_sl == s1) {
      cur_sl = 0;
      // ---- tile finished: filter.  C layout of 16x16x32: column (query) = lane & 15, row = (lane >> 4) * 4 + reg
      const uint32_t gt = (uint32_t)p.tile0 + b + cur_i * nwg;
      const uint32_t row_base = gt * TILE + w * 32 + lq * 4;  // + mb * 16 + reg
      ++cur_i;
      float thr[NQB];
      bool any = false;
      if constexpr (FIRST) {
        // bootstrap: every score is kept (one tile per workgroup, so this runs once, after the last slice).  The
        // accumulator layout has 16 queries x 4 row quads per store instruction, i.e. 64 different 32-byte sectors of
        // the [query][slot] survivor array; transposing through the (now idle) ring memory turns that into 512
        // contiguous bytes per store instruction.
        constexpr int TSTR = TILE + 4;                                   // floats per query row in LDS
        float* T = reinterpret_cast<float*>(smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // trailing DMA pieces have landed ...
        __builtin_amdgcn_s_barrier();                                    // ... for every wave: the rings are free
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            *reinterpret_cast<f32x4*>(T + (nb * 16 + l15) * TSTR + w * 32 + mb * 16 + lq * 4) = acc[mb][nb];
        __syncthreads();
        const uint32_t row0 = gt * TILE;
        for (int qi = w * (NQB * 2); qi < (w + 1) * (NQB * 2); ++qi) {
          const uint32_t q = q0 + qi;
          if (q >= (uint32_t)p.nq) break;
          uint64_t* dst = p.st.surv + (uint64_t)q * p.st.cap + row0;
          float* dstf = p.samp_out ? p.samp_out + (uint64_t)q * p.samp_ld + row0
                                   : reinterpret_cast<float*>(p.st.surv + (uint64_t)q * p.st.cap) + row0;
#pragma unroll
          for (int it = 0; it < TILE / 64; ++it) {
            const uint32_t rl = it * 64 + lane;
            if (row0 + rl < (uint64_t)p.n) {
              if (p.scores_only) {
                if (KS > 1) atomicAdd(dstf + rl, T[qi * TSTR + rl]);      // partial score of this K range, onto zeros
                else dstf[rl] = T[qi * TSTR + rl];
              } else {
                dst[rl] = pack_entry(T[qi * TSTR + rl], row0 + rl);
              }
            }
          }
        }
      }
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) {
        if constexpr (FIRST) break;
        thr[nb] = thr_s[nb * 16 + l15];
        float m = acc[0][nb][0];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mb][nb][r]);
        any |= m >= thr[nb];
      }
      if (!FIRST && __ballot(any)) {
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          float m = acc[0][nb][0];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mb][nb][r]);
          if (__ballot(m >= thr[nb]) == 0) continue;          // wave-uniform
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const uint32_t row = row_base + mb * 16 + r;
              const float sc = acc[mb][nb][r];
              const bool keep = sc >= thr[nb] && row < (uint64_t)p.n;
              const unsigned long long km = __ballot(keep);
              if (km) {
                const uint32_t pos = my_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                if (keep) {
                  if (__builtin_expect(pos < p.rec_cap, 1)) reinterpret_cast<uint4*>(my_rec)[pos] = make_uint4(__float_as_uint(sc), row, nb * 16 + l15, 0u);
                  else spill_record(p.st, sc, row, nb * 16 + l15);      // segment full: gemm_select.hip
                }
                my_cnt += (uint32_t)__popcll(km);
              }
            }
        }
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  if (!FIRST && lane == 0) {
    p.rec_cnt[b * 8 + w] = my_cnt < p.rec_cap ? my_cnt : p.rec_cap;    // (records beyond the segment went to spill_record)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // trailing (unused) DMA pieces land before the LDS is released
}

template <int NQB, bool F16, int MODE>
static void launch_stream_variant(const ScoreArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)SS_DEPTH * SLICE_BYTES + (size_t)SS_DEPTH * NQB * 1024 + NQB * 16 * 4;
  ensure_dynamic_lds((const void*)stream_select_kernel<NQB, F16, MODE>);
  const unsigned grid = MODE == 2 ? (unsigned)a.ntiles * (unsigned)((a.nq + NQB * 16 - 1) / (NQB * 16)) * (unsigned)a.ksplit
                                  : gemm_select_grid();
  hipEvent_t e0, e1;
  take_launch_events(&e0, &e1);
  if (e0 && e1) hipExtLaunchKernelGGL((stream_select_kernel<NQB, F16, MODE>), dim3(grid), dim3(512), lds, stream, e0, e1, 0, a);
  else hipLaunchKernelGGL((stream_select_kernel<NQB, F16, MODE>), dim3(grid), dim3(512), lds, stream, a);
}

bool stream_select_applies(const ScoreArgs& a) {
  return a.small_batch_kernel && a.nqt == 1 && a.nq <= STREAM_MAX_QUERIES && a.debug == 0;
}

template <int NQB, bool F16>
static void launch_stream_mode(const ScoreArgs& a, bool first, hipStream_t stream) {
  if (first) return launch_stream_variant<NQB, F16, 2>(a, stream);
  if (a.cond) return launch_stream_variant<NQB, F16, 1>(a, stream);
  return launch_stream_variant<NQB, F16, 0>(a, stream);
}

bool stream_bootstrap_applies(const ScoreArgs& a) { return a.small_batch_kernel && a.debug == 0; }

void launch_stream_select(const ScoreArgs& a, bool first, hipStream_t stream) {
  // bootstrap launches are one workgroup per (sample tile, group of NQB * 16 queries): with <= 512 queries the narrower
  // group fills the 256 CUs (32 tiles x 8 groups) where the wider one would leave half of them idle for a whole tile time
  if (first) {
    if (a.nq <= 512) a.img_f16 ? launch_stream_mode<4, true>(a, first, stream) : launch_stream_mode<4, false>(a, first, stream);
    else a.img_f16 ? launch_stream_mode<8, true>(a, first, stream) : launch_stream_mode<8, false>(a, first, stream);
    return;
  }
  // filtered launches: as many blocks of 16 query slots as the batch needs, at least four (the reference's online stage has 1
  // query, its test sets 70: src/online.py:132-147, src/test_rOP1m.py:144-149) -- fewer MFMAs, fragment reads and query
  // DMA per gallery slice in a kernel that should wait for HBM only.  Same box, alternating builds
  // (profiles/r03b_qsweep_ab.txt): 70 queries 0.675 -> 0.637 ms with 5 blocks instead of 8, 100 queries 0.691 -> 0.669 with
  // 7; fewer than four blocks buy nothing (16 queries: 0.611 ms with one block, 0.602 with four)
  switch ((a.nq + 15) / 16) {
#define MI_STREAM_CASE(N)                                                                                              \
  case N:                                                                                                              \
    a.img_f16 ? launch_stream_mode<N, true>(a, first, stream) : launch_stream_mode<N, false>(a, first, stream);       \
    break;
    MI_STREAM_CASE(5)
    MI_STREAM_CASE(6)
    MI_STREAM_CASE(7)
    case 8:
      a.img_f16 ? launch_stream_mode<8, true>(a, first, stream) : launch_stream_mode<8, false>(a, first, stream);
      break;
    default:
      a.img_f16 ? launch_stream_mode<4, true>(a, first, stream) : launch_stream_mode<4, false>(a, first, stream);
#undef MI_STREAM_CASE
  }
}

}  // namespace mi
