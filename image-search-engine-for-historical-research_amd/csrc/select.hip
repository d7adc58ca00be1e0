// Per-query selection kernels around the scoring pass (gfx950): threshold maintenance by radix select,
// candidate extraction with the rigorous error margin, exact f64 re-score, final bitonic sort, and the
// multi-shard merge.  Together they replace the `np.argsort(dist)[:K]` of matching_L2
// (src/utils/nnsearch.py:703) and `np.argsort(-scores, axis=0)` of src/utils/Reranking.py:207 for the
// top-K the callers actually consume (src/online.py:152, src/test_rOP1m.py:157-159).
#include <map>
#include <mutex>
#include <set>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace mi {

// diagnostics (scripts/tailbench.hip): the selection kernels return after phase N of their work; 0 = run to the end (product)
static int g_tail_debug_phase = 0;
void set_tail_debug_phase(int phase) { g_tail_debug_phase = phase; }

int ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  if (done.insert({kernel, dev}).second)
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return dev;
}

int current_device_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 256;
  (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
  cus[dev] = n;
  return n;
}


// ------------------------------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes of a wave by DPP moves (row_shr 1, 2, 4, 8, then row_bcast 15 / 31): six VALU
// steps instead of six ds_bpermute round trips (~100 cycles each) of a __shfl-based scan
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return x;
}

// block-wide K-th largest of n keys (MSB-first 8-bit radix select): keys[0 .. min(n, lds_n)) in LDS, the rest -- degenerate
// data only -- through rest_at(i) (global memory).  lds_n must be a multiple of blockDim.x.
// hist: 1024 uint32 of LDS scratch = one 256-bin histogram PER PASS, all cleared up front, and EVERY wave scans the
// histogram of a pass for itself (same bins, same answer): a pass costs ONE barrier, between its adds and its scan.  The
// cross-lane steps are v_readlane / DPP, not LDS permutes: these kernels are chains of latencies, not throughput
// (scripts/tailbench.hip: the select of 900 keys went 12 -> 4 us).  All threads must call; returns the key.  Requires
// 1 <= K <= n and blockDim.x a multiple of 64.
template <typename RestAt>
__device__ uint32_t block_kth_largest_f(const uint32_t* keys, uint32_t lds_n, RestAt rest_at, uint32_t n, uint32_t K,
                                        uint32_t* hist) {
  uint32_t prefix = 0, mask = 0, remaining = K;
  for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int l = threadIdx.x & 63;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    uint32_t* h = hist + pass * 256;
    // Scores of one query share sign and exponent, so in the leading passes nearly all keys of a wave fall into ONE bin and
    // a per-lane LDS atomic would serialise 64 adds on one address.  The wave first peels off up to four digit values held
    // by many lanes -- one add of the lane count each -- and only the remaining lanes add for themselves.
    auto add_key = [&](bool have, uint32_t k) {
      bool live = have && (k & mask) == prefix;
      const uint32_t digit = (k >> shift) & 255u;
      unsigned long long todo = __ballot(live);
#pragma unroll 1
      for (int peel = 0; peel < 4 && todo; ++peel) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader);
        const unsigned long long same = __ballot(live && digit == dl);
        if (l == leader) atomicAdd(&h[dl], (uint32_t)__popcll(same));
        if (digit == dl) live = false;
        todo &= ~same;
      }
      if (live) atomicAdd(&h[digit], 1u);
    };
    const uint32_t n_lds = min(n, lds_n);
    for (uint32_t i0 = 0; i0 < n_lds; i0 += blockDim.x) {   // i0 is block-uniform: every wave runs the same trip count
      const uint32_t i = i0 + threadIdx.x;
      add_key(i < n_lds, i < n_lds ? keys[i] : 0u);
    }
    for (uint32_t i0 = lds_n; i0 < n; i0 += blockDim.x) {
      const uint32_t i = i0 + threadIdx.x;
      add_key(i < n, i < n ? rest_at(i) : 0u);
    }
    __syncthreads();
    // suffix sums over bins 255..0, by every wave for itself: lane l owns bins 4l..4l+3
    const uint4 hv = *reinterpret_cast<const uint4*>(h + 4 * l);
    const uint32_t tot = hv.x + hv.y + hv.z + hv.w;
    const uint32_t inc = wave_prefix_sum(tot);                                     // bins 0 .. 4l+3
    const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    // bins 4l+3 .. 4l in descending order; exactly one (lane, bin) holds the key of rank `remaining`
    uint32_t c = all - inc;   // keys in bins of higher lanes
    const uint32_t hs[4] = {hv.w, hv.z, hv.y, hv.x};
    uint32_t my_bin = 0, my_rem = 0;
    bool found = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c < remaining && c + hs[e] >= remaining) {
        my_bin = (uint32_t)(4 * l + 3 - e);
        my_rem = remaining - c;
        found = true;
      }
      c += hs[e];
    }
    const int owner = __ffsll((long long)__ballot(found)) - 1;
    prefix |= (uint32_t)__builtin_amdgcn_readlane((int)my_bin, owner) << shift;
    mask |= 255u << shift;
    remaining = (uint32_t)__builtin_amdgcn_readlane((int)my_rem, owner);
  }
  __syncthreads();                      // callers reuse hist right away
  return prefix;
}
// keys[0..n) held in LDS
__device__ uint32_t block_kth_largest(const uint32_t* keys, uint32_t n, uint32_t K, uint32_t* hist) {
  return block_kth_largest_f(keys, 0xFFFFFC00u, [](uint32_t) { return 0u; }, n, K, hist);
}

// ------------------------------------------------------------------------------------------------
__global__ void init_query_state_kernel(const RowStat* __restrict__ qstat, const float* __restrict__ gstat3,
                                        int32_t nq, int32_t qpad, float gamma, int use_img_terms,
                                        uint32_t first_cnt, QueryState st) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0) *st.repair = 0;           // (st.flags is sticky across batches: read and cleared by the host)
  if (q >= qpad) return;
  if (q < nq) {
    const RowStat r = qstat[q];
    const float g_f32 = gstat3[0], g_bf = gstat3[1], g_diff = gstat3[2];
    // |approx - exact| <= gamma*|q^||g^| + |q^||g^ - g| + |q^ - q||g|   (Cauchy-Schwarz)
    float eps;
    if (use_img_terms)
      eps = gamma * r.norm_img * g_bf + r.norm_img * g_diff + r.norm_diff * g_f32;
    else
      eps = gamma * r.norm_f32 * g_f32;
    float margin = 2.0f * eps * 1.001f + 1e-30f;
    float thr0 = -INFINITY;
    if (!(margin == margin)) margin = 0.f;   // NaN query (zero-norm): nothing will match anyway
    if (use_img_terms && !isfinite(r.norm_img) && isfinite(r.norm_f32)) {
      // the 16-bit image of this query overflowed (fp16 range): skip it here, the exact f32 path answers it
      atomicOr(st.flags, FLAG_RANGE);
      margin = 0.f;
      thr0 = INFINITY;
    }
    st.thr[q] = thr0;
    st.thr2[q] = thr0;
    st.qflag[q] = 0;
    st.margin[q] = margin;
    st.cnt[q * CNT_STRIDE] = first_cnt;
  } else {
    st.thr[q] = INFINITY;
    st.thr2[q] = INFINITY;
    st.qflag[q] = 0;
    st.margin[q] = 0.f;
    st.cnt[q * CNT_STRIDE] = 0;
  }
  if (st.lad_tc) {                        // ladder off until a sample-threshold launch turns it on
    st.lad_tc[q] = INFINITY;
    st.lad_pack[q] = 0x7F807F80u;         // (+inf, +inf)
    st.lad_cnt[q] = 0;
  }
}

void launch_init_query_state(const RowStat* qstat, const float* gstat3, int32_t nq, int32_t qpad, float gamma,
                             int use_img_terms, uint32_t first_cnt, QueryState st, hipStream_t stream) {
  hipLaunchKernelGGL(init_query_state_kernel, dim3((qpad + 255) / 256), dim3(256), 0, stream, qstat, gstat3, nq,
                     qpad, gamma, use_img_terms, first_cnt, st);
}

// ------------------------------------------------------------------------------------------------
// maintain: L = K-th largest approximate score among the survivors so far (a lower bound of the final
// K-th largest); threshold <- L - margin; survivors below the new threshold are dropped.
// Only the 4-byte keys live in LDS, and only the first MAINT_LDS_KEYS of them: 32 KiB + 2 KiB per workgroup lets FOUR
// 512-thread workgroups share a CU, i.e. the 1024 queries of a batch run in one round (the keys of all survivor_cap
// entries took 50 KiB: three per CU, two rounds; a descriptor batch has ~900 survivors per query).  Keys beyond that
// (degenerate data only) are recomputed from the entries in global memory by the select passes.  The 8-byte entries stay
// in registers between the read and the in-place compaction (all reads complete before the first write).
constexpr int MAINT_THREADS = 512;
constexpr uint32_t MAINT_LDS_KEYS = 8192;
constexpr int MAINT_PER_THREAD_MAX = 32;        // 512 * 32 = 16384 = largest survivor_cap

// MODE 0 (after the bootstrap chunk).  spec_r > 0: SPECULATIVE threshold for the single remaining scoring launch =
//   the spec_r-th largest sample score (>= K rows above it exist in the whole shard with probability 1 - 1e-6, see
//   api_schedule.hip), and thr2 = the (4*spec_r)-th largest as the looser threshold of the repair pass.  A speculative
//   threshold needs no error margin: it is verified after the scan (MODE 1).
// MODE 1 (after the last launch): topvals / L, and with spec != 0 the verification: the survivors are ALL rows with
//   approximate score >= thr, so if there are >= K of them L is the true K-th largest approximate score, and if also
//   L - margin >= thr every candidate row is among them.  Otherwise the query is flagged for the repair pass
//   (device-side, conditional launches, no host round trip); a query failing again raises FLAG_SPEC_FAIL.
//   repair == 1: the repair pass itself, only flagged queries are processed; repair == 2: no repair pass will follow (small
//   batches, api_schedule.hip), a failed query raises FLAG_SPEC_FAIL at once.  cond: skip the launch when *cond == 0.
// PT = entries per thread kept in registers (PT * 512 >= survivor_cap): 24 instead of 32 at the default cap frees the
// registers for a second workgroup per CU in MODE 1
//   repair == 3 (instantiation SCAN; small batches on the asynchronous entry points, round 4): no repair pass follows EITHER,
//   and nobody may have to read a flag: the workgroup of a failed query repairs it itself -- it scans every stored f32 row of
//   the shard against the query (one wave per row, float64 accumulation: scores at least as good as the image's, so the
//   certificate's bounds hold a fortiori), keeps the rows at or above the looser threshold thr2 and runs the selection again
//   on them.  ~0.1 s for a 1 M-row shard, once per 10^7 queries, instead of three empty launches behind every batch.
template <int MODE, int PT, bool SCAN = false>
__global__ __launch_bounds__(MAINT_THREADS) void select_maintain_kernel(QueryState st, int32_t k,
                                                                        float* __restrict__ topvals,
                                                                        float* __restrict__ l_local,
                                                                        uint64_t* __restrict__ stats2, int32_t spec_r,
                                                                        int32_t spec, int32_t repair,
                                                                        const uint32_t* __restrict__ cond,
                                                                        uint32_t* __restrict__ cand_rows,
                                                                        uint32_t* __restrict__ cand_cnt, uint32_t rcap,
                                                                        int dbg_phase, RepairScan rs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (cond && *cond == 0) return;
  const uint32_t q = blockIdx.x;
  if (MODE == 1 && repair == 1 && st.qflag[q] == 0) return;      // repair pass: flagged queries only
  const uint32_t cap = st.cap;
  const uint32_t lds_keys = min(cap, MAINT_LDS_KEYS);
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);
  uint32_t* hist = keys + ((lds_keys + 3u) & ~3u);    // 1024 words, 16-byte aligned: one histogram per pass (block_kth_largest_f)
  uint32_t* sh = hist + 1024;
  // the per-query words this workgroup needs later are requested up front: each is a global round trip of its own
  // otherwise, in the middle of the barrier chain
  uint32_t cnt_q = st.cnt[q * CNT_STRIDE];
  const float margin_q = st.margin[q];
  float used_thr = st.thr[q];
  const float thr2_q = st.thr2[q];
  const uint32_t lad_cnt_q = st.lad_cnt ? st.lad_cnt[q] : 0u;
  int attempt = 0;                                   // SCAN: 1 = the selection runs again on the rows the scan kept
restart:
  const uint32_t n = min(cnt_q, cap);
  if (MODE == 1 && cnt_q > cap && threadIdx.x == 0) atomicOr(st.flags, FLAG_SURV_OVERFLOW);
  uint64_t* gsurv = st.surv + (uint64_t)q * cap;
  uint64_t ent[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const uint32_t i = threadIdx.x + j * MAINT_THREADS;
    ent[j] = (i < n) ? gsurv[i] : 0ull;
    if (i < n && i < lds_keys) keys[i] = f2key(entry_score(ent[j]));
  }
  auto rest_at = [&](uint32_t i) -> uint32_t { return f2key(entry_score(gsurv[i])); };       // keys beyond the LDS part
  auto key_at = [&](uint32_t i) -> uint32_t { return i < lds_keys ? keys[i] : rest_at(i); };
  if (dbg_phase == 1) { if (ent[0] == 1ull && n == 0xFFFFFFFFu) st.flags[3] = 1; return; }
  if (threadIdx.x == 0) { sh[2] = 0; sh[3] = 0; }
  __syncthreads();                                  // all entries are in registers from here on
  float L = -INFINITY;
  uint32_t keyL = 0;
  if (n >= (uint32_t)k) {
    keyL = block_kth_largest_f(keys, lds_keys, rest_at, n, (uint32_t)k, hist);
    L = key2f(keyL);
  }
  float thr_new = L - margin_q;
  float thr2 = thr_new;
  if (dbg_phase == 2) { if (threadIdx.x == 0) l_local[q] = L; return; }
  if (MODE == 0 && spec_r > 0 && n >= (uint32_t)k) {
    // the spec_r-th and (4 spec_r)-th largest lie among the keys >= keyL (spec_r, 4 spec_r < k): gather those (k plus
    // ties, normally ~k) into the histogram scratch and rank them by counting instead of two more full radix selects
    __syncthreads();
    if (threadIdx.x == 0) sh[3] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
      if (key_at(i) >= keyL) {
        const uint32_t pos = atomicAdd(&sh[3], 1u);
        if (pos < 256) hist[pos] = key_at(i);
      }
    __syncthreads();
    const uint32_t m = sh[3];
    uint32_t want[2] = {(uint32_t)spec_r, (uint32_t)(4 * spec_r)};
    __syncthreads();
    if (m <= 256) {
      if (threadIdx.x < m) {
        const uint32_t me = hist[threadIdx.x];
        uint32_t gt = 0, ge = 0;
        for (uint32_t j = 0; j < m; ++j) { gt += hist[j] > me; ge += hist[j] >= me; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (want[t] < (uint32_t)k && gt < want[t] && want[t] <= ge) sh[4 + t] = me;
      }
      __syncthreads();
      if (spec_r < k) thr_new = fmaxf(thr_new, key2f(sh[4]) - margin_q);
      if (4 * spec_r < k) thr2 = fmaxf(thr2, key2f(sh[5]) - margin_q);
    } else {                                        // a crowd of ties at the K-th score: plain selects
      if (spec_r < k) thr_new = fmaxf(thr_new, key2f(block_kth_largest_f(keys, lds_keys, rest_at, n, (uint32_t)spec_r, hist)) - margin_q);
      if (4 * spec_r < k)
        thr2 = fmaxf(thr2, key2f(block_kth_largest_f(keys, lds_keys, rest_at, n, (uint32_t)(4 * spec_r), hist)) - margin_q);
    }
    if (threadIdx.x == 0) sh[3] = 0;
    __syncthreads();
  }
  bool failed = false;
  if (MODE == 1 && spec) {
    const float used = used_thr;
    // ladder validated: >= K rows with approx >= t_c were emitted, so L >= t_c and every row with approx >= t_c - margin
    // (the tightest threshold any wave applied) is among the survivors: nothing speculative is left to verify
    const bool lad_ok = st.lad_cnt && repair != 1 && attempt == 0 && lad_cnt_q >= (uint32_t)k && n >= (uint32_t)k;
    failed = !lad_ok && (used > -INFINITY) && !(n >= (uint32_t)k && L - margin_q >= used);
    if (failed) thr_new = thr2_q;
  }
  if (SCAN && MODE == 1 && failed && repair == 3 && attempt == 0 && rs.gal_f32) {
    // in-kernel repair of this query: every stored row of the shard against it, rows with score >= thr2 become the survivors
    __syncthreads();
    if (threadIdx.x == 0) {
      sh[6] = 0;
      if (rs.repairs) rs.repairs[q] += 1;           // the one trace of a repair: a sub-millisecond call that took 0.1 .. 1 s
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = MAINT_THREADS / 64;
    const int nvec = rs.dp >> 2;
    const float4* qv = reinterpret_cast<const float4*>(rs.qry_f32 + (uint64_t)q * rs.dp);
    for (int64_t r0 = wv * 2; r0 < rs.n; r0 += 2 * nw) {
      const int64_t r1 = r0 + 1 < rs.n ? r0 + 1 : r0;
      const float4* g0 = reinterpret_cast<const float4*>(rs.gal_f32 + (uint64_t)r0 * rs.dp);
      const float4* g1 = reinterpret_cast<const float4*>(rs.gal_f32 + (uint64_t)r1 * rs.dp);
      double a0 = 0.0, a1 = 0.0;
      for (int v = lane; v < nvec; v += 64) {
        const float4 x = qv[v], y0 = g0[v], y1 = g1[v];
        a0 += (double)x.x * (double)y0.x; a0 += (double)x.y * (double)y0.y;
        a0 += (double)x.z * (double)y0.z; a0 += (double)x.w * (double)y0.w;
        a1 += (double)x.x * (double)y1.x; a1 += (double)x.y * (double)y1.y;
        a1 += (double)x.z * (double)y1.z; a1 += (double)x.w * (double)y1.w;
      }
      for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
      }
      if (lane == 0) {
        if ((float)a0 >= thr2_q) {
          const uint32_t pos = atomicAdd(&sh[6], 1u);
          if (pos < cap) gsurv[pos] = pack_entry((float)a0, (uint32_t)r0);
        }
        if (r1 != r0 && (float)a1 >= thr2_q) {
          const uint32_t pos = atomicAdd(&sh[6], 1u);
          if (pos < cap) gsurv[pos] = pack_entry((float)a1, (uint32_t)r1);
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    cnt_q = sh[6];
    if (cnt_q > cap && threadIdx.x == 0) atomicOr(st.flags, FLAG_SURV_OVERFLOW);
    used_thr = thr2_q;
    attempt = 1;
    __syncthreads();
    goto restart;
  }
  float* tv = topvals + (uint64_t)q * k;
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const uint32_t i = threadIdx.x + j * MAINT_THREADS;
    if (i < n && !failed && !(MODE == 0 && spec)) {
      const float sc = entry_score(ent[j]);
      if (sc >= thr_new) {                                               // compaction (order is irrelevant)
        const uint32_t pos = atomicAdd(&sh[2], 1u);
        gsurv[pos] = ent[j];
        // single-shard search: L is the K-th largest approximate score of the whole gallery, so the compacted survivors
        // (score >= L - margin) ARE the candidate set of the certificate: no separate candidates launch
        if (MODE == 1 && cand_rows && pos < rcap) cand_rows[(uint64_t)q * rcap + pos] = entry_row(ent[j]);
      }
      if (MODE == 1) {
        // the K largest approximate values (unsorted): everything above L, ties of L filled in below
        if (n >= (uint32_t)k) {
          if (f2key(sc) > keyL) tv[atomicAdd(&sh[3], 1u)] = sc;
        } else {
          tv[i] = sc;
        }
      }
    }
  }
  __syncthreads();
  if (dbg_phase == 3) return;
  if (MODE == 1) {
    const uint32_t have = (n >= (uint32_t)k) ? sh[3] : n;
    const float fill = (n >= (uint32_t)k) ? L : -INFINITY;
    for (uint32_t i = have + threadIdx.x; i < (uint32_t)k; i += blockDim.x) tv[i] = fill;
    if (threadIdx.x == 0) l_local[q] = L;
  }
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      st.thr[q] = thr_new;
      st.thr2[q] = thr2;
      st.cnt[q * CNT_STRIDE] = spec ? 0u : sh[2];   // spec (MODE 0): entries came from the sample image, drop them
    } else if (spec) {
      if (failed && repair == 0) {                // repair == 2: no repair pass follows, the failure is final
        st.thr[q] = thr_new;                       // = thr2: the repair pass re-scans every tile for this query
        st.cnt[q * CNT_STRIDE] = 0;
        st.qflag[q] = 1;
        atomicOr(st.repair, 1u);
      } else {
        if (failed) atomicOr(st.flags, FLAG_SPEC_FAIL);
        st.thr[q] = INFINITY;                      // verified: invisible to a repair pass
        st.qflag[q] = 0;
        st.cnt[q * CNT_STRIDE] = sh[2];
      }
    } else {
      st.thr[q] = thr_new;
      st.cnt[q * CNT_STRIDE] = sh[2];
    }
    // statistics: per-query accumulators, written by this workgroup only (an atomic on ONE word from every workgroup of
    // the launch serialises at the memory side: 2 x 1024 x ~12 ns was 15 us of this kernel's 30)
    if (MODE == 1 && stats2 && !failed)        // survivors = entries the filter kept (before the cut at L - margin)
      stats2[2 * q] += (uint64_t)n;
    if (MODE == 1 && cand_rows) {
      // a query whose speculative threshold failed has no candidate list yet (the repair launch writes it; if that fails
      // too the batch is flagged and answered again): its count must still be defined, the re-score reads it
      const uint32_t nc = failed ? 0u : min(sh[2], rcap);
      if (!failed && sh[2] > rcap) atomicOr(st.flags, FLAG_CAND_OVERFLOW);
      cand_cnt[q] = nc;
      if (stats2 && nc) stats2[2 * q + 1] += (uint64_t)nc;
    }
  }
}

void launch_select_maintain(QueryState st, int32_t nq, int32_t k, int mode, float* topvals, float* l_local,
                            uint64_t* stats2, int32_t spec_r, int32_t spec, int32_t repair, const uint32_t* cond,
                            hipStream_t stream, uint32_t* cand_rows, uint32_t* cand_cnt, uint32_t rcap, const RepairScan* scan) {
  const size_t lds = (size_t)std::min<uint32_t>(st.cap, MAINT_LDS_KEYS) * 4 + 16 + 1024 * 4 + 32;     // keys | hist[1024] | sh[8]
  const RepairScan rs = scan ? *scan : RepairScan{};
  auto go = [&](auto kern) {
    ensure_dynamic_lds((const void*)kern);                     // survivor_cap = 16384 needs 66.6 KB
    hipLaunchKernelGGL(kern, dim3(nq), dim3(MAINT_THREADS), lds, stream, st, k, topvals, l_local, stats2, spec_r, spec,
                       repair, cond, cand_rows, cand_cnt, rcap, g_tail_debug_phase, rs);
  };
  const uint32_t per_thread = (st.cap + MAINT_THREADS - 1) / MAINT_THREADS;
  if (mode == 1 && repair == 3 && scan) {                      // in-kernel repair (small batches, asynchronous entry points)
    if (per_thread <= 16) go(select_maintain_kernel<1, 16, true>);
    else if (per_thread <= 24) go(select_maintain_kernel<1, 24, true>);
    else go(select_maintain_kernel<1, MAINT_PER_THREAD_MAX, true>);
    return;
  }
  if (mode == 0) {
    if (per_thread <= 16) go(select_maintain_kernel<0, 16>);
    else if (per_thread <= 24) go(select_maintain_kernel<0, 24>);
    else go(select_maintain_kernel<0, MAINT_PER_THREAD_MAX>);
  } else {
    if (per_thread <= 16) go(select_maintain_kernel<1, 16>);
    else if (per_thread <= 24) go(select_maintain_kernel<1, 24>);
    else go(select_maintain_kernel<1, MAINT_PER_THREAD_MAX>);
  }
}

// ------------------------------------------------------------------------------------------------
// Thresholds from the bootstrap sample (single-launch schedule): only the r-th and the (4r)-th largest of the n_s sample
// scores are needed (r < K; see api_schedule.hip), i.e. the top few dozen of 8192 values -- a full radix select over all keys
// (select_maintain_kernel<0>) spends most of its time funnelling LDS atomics into the 2-4 bins that share the scores'
// sign and exponent.  Here every thread keeps its 16 scores in registers and contributes its maximum; the W-th largest
// of the 512 maxima (W = 4r) is a lower bound of the W-th largest overall, so the values >= it (W plus a few) are
// gathered and ranked exactly by counting.  Falls back to the plain select if more than 256 values qualify (ties).
// The sample entries are dropped afterwards (cnt = 0), like select_maintain_kernel<0> with spec != 0.
// what a sample-threshold kernel leaves for its query: thr / thr2 from the r-th / (4 r)-th largest sample score, the ladder level
__device__ __forceinline__ void store_sample_thresholds(const QueryState& st, uint32_t q, uint32_t key1, uint32_t key2,
                                                        uint32_t key3, int32_t lad_r, float margin_q, float thr_in,
                                                        float order_slack) {
  {
    // order_slack: the sample scores were summed in another order than the scoring launch sums (K-split bootstrap,
    // ScoreArgs::ksplit).  Two f32 summation orders of the same products differ by at most gamma |q^| |g^| <= eps = margin / 2
    // (DESIGN section 4), and the verification below has no slack of its own: with exact duplicates at rank r the threshold
    // would sit an ulp above the scores the scoring launch gives those very rows.  Half a margin more keeps them.
    const float margin = margin_q * (1.0f + order_slack);
    // thr = score(r) - margin (>= the rigorous L_sample - margin since r < K); thr2 = score(min(4r, K)) - margin
    // minus the margin: the verification asks for L - margin >= thr, and L >= score(r) is what the rank guarantees
    const float thr = key2f(key1) - margin, thr2 = key2f(key2) - margin;
    const bool excluded = thr_in == INFINITY;                          // query excluded at init (range overflow)
    st.thr[q] = excluded ? INFINITY : thr;
    st.thr2[q] = excluded ? INFINITY : thr2;
    st.cnt[q * CNT_STRIDE] = 0;
    if (st.lad_tc) {
      // ladder level: t_c = score(lad_r), lad_r < r, so t_c >= score(r) >= every threshold a wave applies and every row
      // with approx >= t_c is emitted and counted; once K are counted, L >= t_c and t_c - margin is a rigorous threshold
      const bool on = lad_r > 0 && !excluded;
      const float tc = on ? key2f(key3) : INFINITY;
      st.lad_tc[q] = tc;
      st.lad_pack[q] = bf16_down(excluded ? INFINITY : thr) | (bf16_down(on ? tc - margin : INFINITY) << 16);
      st.lad_cnt[q] = 0;
    }
  }
}

constexpr int SAMP_THREADS = 512;                          // x SAMP_PER_THREAD (2, 4, 8, 16) = 1024 ... 8192 sample scores; 48 = 24 576
                                                           // (round 6: the sample of shards beyond 160 x 8192 rows; 4-byte scores only)
template <int SAMP_PER_THREAD>
__global__ __launch_bounds__(SAMP_THREADS) void sample_threshold_kernel(QueryState st, int32_t k, int32_t spec_r,
                                                                        int32_t lad_r, int32_t f32_scores, int dbg_phase,
                                                                        float order_slack) {
  constexpr int LDS_PER = SAMP_PER_THREAD < 16 ? SAMP_PER_THREAD : 16;
  __shared__ uint32_t keys[SAMP_THREADS * LDS_PER];           // only used by the fallback (up to 32 KiB; keys beyond: from memory)
  __shared__ __attribute__((aligned(16))) uint32_t hist[1024];   // the select's histograms; first 256 words: gather buffer
  __shared__ uint32_t sh[8];
  const uint32_t q = blockIdx.x;
  const float margin_q = st.margin[q], thr_in = st.thr[q];    // requested up front (thread 0 needs them at the very end)
  const uint32_t n = min(st.cnt[q * CNT_STRIDE], (uint32_t)(SAMP_THREADS * SAMP_PER_THREAD));
  const uint64_t* gsurv = st.surv + (uint64_t)q * st.cap;
  uint32_t kv[SAMP_PER_THREAD];
  uint32_t kmax = 0;
#pragma unroll
  for (int j = 0; j < SAMP_PER_THREAD; ++j) {
    const uint32_t i = threadIdx.x + j * SAMP_THREADS;
    // f32_scores: the bootstrap launch stored bare 4-byte scores (ScoreArgs::scores_only), half the bytes of the entries
    kv[j] = (i < n) ? f2key(f32_scores ? reinterpret_cast<const float*>(gsurv)[i] : entry_score(gsurv[i])) : 0u;
    kmax = max(kmax, kv[j]);
  }
  if (dbg_phase == 1) { if (kmax == 1u && n == 0xFFFFFFFFu) st.flags[3] = 1; return; }
  const uint32_t w1 = (uint32_t)spec_r, w2 = (uint32_t)min(4 * spec_r, k);   // wanted ranks, w1 <= w2 <= 256
  uint32_t* maxima = keys;                                  // 512 keys
  maxima[threadIdx.x] = kmax;
  if (threadIdx.x == 0) { sh[3] = 0; sh[4] = 0; sh[5] = 0; sh[6] = 0; }
  __syncthreads();
  const uint32_t t0 = block_kth_largest(maxima, SAMP_THREADS, w2, hist);
  __syncthreads();
  if (dbg_phase == 2) { if (threadIdx.x == 0 && t0 == 1u) st.flags[3] = 1; return; }
#pragma unroll
  for (int j = 0; j < SAMP_PER_THREAD; ++j)
    if (kv[j] >= t0 && kv[j] != 0u) {
      const uint32_t pos = atomicAdd(&sh[3], 1u);
      if (pos < 256) hist[pos] = kv[j];
    }
  __syncthreads();
  const uint32_t m = sh[3];
  uint32_t key1, key2, key3 = 0;
  if (m <= 256) {
    if (threadIdx.x < m) {
      const uint32_t me = hist[threadIdx.x];
      uint32_t gt = 0, ge = 0;
      for (uint32_t j = 0; j < m; ++j) { gt += hist[j] > me; ge += hist[j] >= me; }
      if (gt < w1 && w1 <= ge) sh[4] = me;
      if (gt < w2 && w2 <= ge) sh[5] = me;
      if (lad_r > 0 && gt < (uint32_t)lad_r && (uint32_t)lad_r <= ge) sh[6] = me;
    }
    __syncthreads();
    key1 = sh[4];
    key2 = sh[5];
    key3 = sh[6];
  } else {                                                  // a crowd of ties: plain selects over all keys
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LDS_PER; ++j) keys[threadIdx.x + j * SAMP_THREADS] = kv[j];
    __syncthreads();
    auto rest_at = [&](uint32_t i) -> uint32_t {            // keys beyond the LDS part (24 576-score samples only)
      return f2key(f32_scores ? reinterpret_cast<const float*>(gsurv)[i] : entry_score(gsurv[i]));
    };
    key1 = block_kth_largest_f(keys, (uint32_t)(SAMP_THREADS * LDS_PER), rest_at, n, w1, hist);
    key2 = block_kth_largest_f(keys, (uint32_t)(SAMP_THREADS * LDS_PER), rest_at, n, w2, hist);
    if (lad_r > 0) key3 = block_kth_largest_f(keys, (uint32_t)(SAMP_THREADS * LDS_PER), rest_at, n, (uint32_t)lad_r, hist);
  }
  if (threadIdx.x == 0) store_sample_thresholds(st, q, key1, key2, key3, lad_r, margin_q, thr_in, order_slack);
}

// The same for the 65 536-row sample of shards beyond 160 x 24 576 rows (round 6: the 10 M-row gallery of BASELINE configs[3]
// on one or two GPUs): the scores do not fit a survivor row, the bootstrap launch writes them to a buffer of their own
// (ScoreArgs::samp_out, [queries][ld] floats), and they are read twice here instead of being held in registers -- once for the
// per-thread maxima, once to gather the few dozen values at or above the (4 r)-th largest maximum.  Same answers' contract:
// the r-th, (4 r)-th and lad_r-th largest of the n sample scores, exactly.
__global__ __launch_bounds__(SAMP_THREADS) void sample_threshold_big_kernel(QueryState st, const float* __restrict__ scores,
                                                                            uint32_t ld, uint32_t n, int32_t k, int32_t spec_r,
                                                                            int32_t lad_r, float order_slack) {
  __shared__ uint32_t keys[SAMP_THREADS];
  __shared__ __attribute__((aligned(16))) uint32_t hist[1024];
  __shared__ uint32_t sh[8];
  const uint32_t q = blockIdx.x;
  const float margin_q = st.margin[q], thr_in = st.thr[q];
  const float* sc = scores + (uint64_t)q * ld;
  uint32_t kmax = 0;
  for (uint32_t i = threadIdx.x; i < n; i += SAMP_THREADS) kmax = max(kmax, f2key(sc[i]));
  const uint32_t w1 = (uint32_t)spec_r, w2 = (uint32_t)min(4 * spec_r, k);
  keys[threadIdx.x] = kmax;
  if (threadIdx.x == 0) { sh[3] = 0; sh[4] = 0; sh[5] = 0; sh[6] = 0; }
  __syncthreads();
  const uint32_t t0 = block_kth_largest(keys, SAMP_THREADS, w2, hist);      // <= the w2-th largest score overall
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += SAMP_THREADS) {
    const uint32_t kv = f2key(sc[i]);
    if (kv >= t0 && kv != 0u) {
      const uint32_t pos = atomicAdd(&sh[3], 1u);
      if (pos < 256) hist[pos] = kv;
    }
  }
  __syncthreads();
  const uint32_t m = sh[3];
  uint32_t key1, key2, key3 = 0;
  if (m <= 256) {
    if (threadIdx.x < m) {
      const uint32_t me = hist[threadIdx.x];
      uint32_t gt = 0, ge = 0;
      for (uint32_t j = 0; j < m; ++j) { gt += hist[j] > me; ge += hist[j] >= me; }
      if (gt < w1 && w1 <= ge) sh[4] = me;
      if (gt < w2 && w2 <= ge) sh[5] = me;
      if (lad_r > 0 && gt < (uint32_t)lad_r && (uint32_t)lad_r <= ge) sh[6] = me;
    }
    __syncthreads();
    key1 = sh[4];
    key2 = sh[5];
    key3 = sh[6];
  } else {                                                  // a crowd of ties: plain selects over all keys, from memory
    __syncthreads();
    auto rest_at = [&](uint32_t i) -> uint32_t { return f2key(sc[i]); };
    key1 = block_kth_largest_f(keys, 0u, rest_at, n, w1, hist);
    key2 = block_kth_largest_f(keys, 0u, rest_at, n, w2, hist);
    if (lad_r > 0) key3 = block_kth_largest_f(keys, 0u, rest_at, n, (uint32_t)lad_r, hist);
  }
  if (threadIdx.x == 0) store_sample_thresholds(st, q, key1, key2, key3, lad_r, margin_q, thr_in, order_slack);
}

void launch_sample_threshold_big(QueryState st, const float* scores, uint32_t ld, uint32_t n, int32_t nq, int32_t k,
                                 int32_t spec_r, int32_t lad_r, float order_slack, hipStream_t stream) {
  hipLaunchKernelGGL(sample_threshold_big_kernel, dim3(nq), dim3(SAMP_THREADS), 0, stream, st, scores, ld, n, k, spec_r, lad_r,
                     order_slack);
}

bool sample_threshold_applies(uint32_t first_cnt, int32_t k, int32_t spec_r) {
  const bool size_ok = first_cnt == SAMP_THREADS * 2u || first_cnt == SAMP_THREADS * 4u || first_cnt == SAMP_THREADS * 8u ||
                       first_cnt == SAMP_THREADS * 16u || first_cnt == SAMP_THREADS * 48u;
  return size_ok && spec_r >= 1 && spec_r < k && 4 * spec_r <= 256 && (int64_t)first_cnt >= k;
}

void launch_sample_threshold(QueryState st, int32_t nq, int32_t k, int32_t spec_r, uint32_t first_cnt, hipStream_t stream,
                             int32_t lad_r, int32_t f32_scores, float order_slack) {
  if (first_cnt == SAMP_THREADS * 2u)
    hipLaunchKernelGGL(sample_threshold_kernel<2>, dim3(nq), dim3(SAMP_THREADS), 0, stream, st, k, spec_r, lad_r, f32_scores,
                       g_tail_debug_phase, order_slack);
  else if (first_cnt == SAMP_THREADS * 4u)
    hipLaunchKernelGGL(sample_threshold_kernel<4>, dim3(nq), dim3(SAMP_THREADS), 0, stream, st, k, spec_r, lad_r, f32_scores,
                       g_tail_debug_phase, order_slack);
  else if (first_cnt == SAMP_THREADS * 8u)
    hipLaunchKernelGGL(sample_threshold_kernel<8>, dim3(nq), dim3(SAMP_THREADS), 0, stream, st, k, spec_r, lad_r, f32_scores,
                       g_tail_debug_phase, order_slack);
  else if (first_cnt == SAMP_THREADS * 16u)
    hipLaunchKernelGGL(sample_threshold_kernel<16>, dim3(nq), dim3(SAMP_THREADS), 0, stream, st, k, spec_r, lad_r, f32_scores,
                       g_tail_debug_phase, order_slack);
  else
    hipLaunchKernelGGL(sample_threshold_kernel<48>, dim3(nq), dim3(SAMP_THREADS), 0, stream, st, k, spec_r, lad_r, f32_scores,
                       g_tail_debug_phase, order_slack);
}

// ------------------------------------------------------------------------------------------------
// candidates: every surviving row whose approximate score is >= L - margin may belong to the exact
// top-K (DESIGN.md "Exactness certificate"); rows below cannot.
__global__ __launch_bounds__(256) void select_candidates_kernel(QueryState st, const float* __restrict__ L,
                                                                uint32_t* __restrict__ cand_rows,
                                                                uint32_t* __restrict__ cand_cnt, uint32_t rcap,
                                                                uint64_t* __restrict__ stats2) {
  __shared__ uint32_t counter;
  const uint32_t q = blockIdx.x;
  const uint32_t n = min(st.cnt[q * CNT_STRIDE], st.cap);
  if (st.cnt[q * CNT_STRIDE] > st.cap && threadIdx.x == 0) atomicOr(st.flags, FLAG_SURV_OVERFLOW);
  if (threadIdx.x == 0) counter = 0;
  __syncthreads();
  const float lim = L[q] - st.margin[q];
  const uint64_t* gsurv = st.surv + (uint64_t)q * st.cap;
  uint32_t* out = cand_rows + (uint64_t)q * rcap;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint64_t e = gsurv[i];
    if (entry_score(e) >= lim) {
      const uint32_t pos = atomicAdd(&counter, 1u);
      if (pos < rcap) out[pos] = entry_row(e);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (counter > rcap) atomicOr(st.flags, FLAG_CAND_OVERFLOW);
    cand_cnt[q] = min(counter, rcap);
    if (stats2) stats2[2 * q + 1] += (uint64_t)min(counter, rcap);      // per-query accumulator (one writer)
  }
}

void launch_select_candidates(QueryState st, int32_t nq, const float* L, uint32_t* cand_rows, uint32_t* cand_cnt,
                              uint32_t rcap, uint64_t* stats2, hipStream_t stream) {
  hipLaunchKernelGGL(select_candidates_kernel, dim3(nq), dim3(256), 0, stream, st, L, cand_rows, cand_cnt, rcap, stats2);
}

// ------------------------------------------------------------------------------------------------
// exact re-score: s = sum_k g[k] * q[k] with f32 inputs, exact f64 products and f64 accumulation
// (HBM-bound gather of 4*dp bytes per candidate row).  One wave per row, two rows per wave pass.

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {          // global_load_dwordx4 ... nt
  const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
constexpr int RESCORE_MAX_THREADS = 256;
template <int UNROLL, int ROWS_PER_WG>
__global__ __launch_bounds__(RESCORE_MAX_THREADS) void rescore_kernel(const float* __restrict__ gal, const float* __restrict__ qry,
                                                      int32_t dp, const uint32_t* __restrict__ cand_rows,
                                                      const uint32_t* __restrict__ cand_cnt, uint32_t rcap,
                                                      double* __restrict__ cand_score, uint32_t last_row) {
  const uint32_t q = blockIdx.y;
  // the count is clamped to the list's capacity and every row id to the shard's last row, like the sibling kernels: a
  // stale or undefined count / id ("the abort of round 2") can then cost a wrong candidate, never a
  // read outside cand_rows or the gallery
  const uint32_t nc = min(cand_cnt[q], rcap);
  if (blockIdx.x * ROWS_PER_WG >= nc) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float4* qv = reinterpret_cast<const float4*>(qry + (uint64_t)q * dp);
  const int nvec = dp >> 2;            // float4 per row (dp is a multiple of 64 -> nvec multiple of 16)
  const uint32_t* rows = cand_rows + (uint64_t)q * rcap;
  double* outs = cand_score + (uint64_t)q * rcap;
  // grid.x covers RESCORE_GRID_X * ROWS_PER_WG candidates per sweep (the usual count fits one sweep; a launch sized for
  // rcap would consist almost entirely of workgroups that exit at once)
  for (uint32_t c0 = blockIdx.x * ROWS_PER_WG; c0 < nc; c0 += gridDim.x * ROWS_PER_WG) {
  const uint32_t cend = min(nc, c0 + ROWS_PER_WG);
  for (uint32_t c = c0 + w * 2; c < cend; c += 2 * (blockDim.x >> 6)) {
    const bool two = (c + 1 < cend);
    const float4* g0 = reinterpret_cast<const float4*>(gal + (uint64_t)min(rows[c], last_row) * dp);
    const float4* g1 = reinterpret_cast<const float4*>(gal + (uint64_t)min(rows[two ? c + 1 : c], last_row) * dp);
    double a0 = 0.0, a1 = 0.0;
    int v = lane;
    // 4 column steps at a time with all 12 loads issued before the first use (the rolled loop keeps only 3 loads of
    // 16 B per lane in flight and pays one memory round trip per step)
    for (; v + 192 < nvec; v += 256) {
      float4 x[4], y0[4], y1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = qv[v + 64 * u];
        // candidate rows are read once per batch (the query row is shared by all its candidates: default policy)
        y0[u] = nt_load4(g0 + v + 64 * u);
        y1[u] = nt_load4(g1 + v + 64 * u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 += (double)x[u].x * (double)y0[u].x; a0 += (double)x[u].y * (double)y0[u].y;
        a0 += (double)x[u].z * (double)y0[u].z; a0 += (double)x[u].w * (double)y0[u].w;
        a1 += (double)x[u].x * (double)y1[u].x; a1 += (double)x[u].y * (double)y1[u].y;
        a1 += (double)x[u].z * (double)y1[u].z; a1 += (double)x[u].w * (double)y1[u].w;
      }
    }
#pragma unroll UNROLL
    for (; v < nvec; v += 64) {
      const float4 x = qv[v];
      const float4 y0 = g0[v];
      const float4 y1 = g1[v];
      a0 += (double)x.x * (double)y0.x; a0 += (double)x.y * (double)y0.y;
      a0 += (double)x.z * (double)y0.z; a0 += (double)x.w * (double)y0.w;
      a1 += (double)x.x * (double)y1.x; a1 += (double)x.y * (double)y1.y;
      a1 += (double)x.z * (double)y1.z; a1 += (double)x.w * (double)y1.w;
    }
    for (int o = 32; o > 0; o >>= 1) {
      a0 += __shfl_xor(a0, o);
      a1 += __shfl_xor(a1, o);
    }
    if (lane == 0) {
      outs[c] = a0;
      if (two) outs[c + 1] = a1;
    }
  }
  }
}

// One wave per workgroup, one 2-row pass per sweep: the finest granularity balances best.  Measured on 1024 x ~127
// candidate rows of 8 KiB (same box): 201 us with 2 rows / 64 threads, 207 us with 4 rows / 128 or 256 threads,
// 266 us with 8 rows / 256 threads, 447 us with 16 rows; the unroll factor of the column loop does not matter.
constexpr int RESCORE_ROWS_PER_WG = 2;
constexpr int RESCORE_THREADS = 64;
constexpr uint32_t RESCORE_GRID_X = 64;     // 128 candidates per query and sweep

// Resident variant for the asynchronous tail: ONE 256-thread workgroup per CU (its four waves land on the four SIMDs), so
// that the launch occupies no more than the 80 VGPRs per SIMD lane the tile kernel (2 x 216) leaves free and the NEXT batch's scoring
// launch finds room on every CU at once (a grid of 64 k one-wave workgroups fills the SIMDs with re-score waves first
// and the persistent kernel then waits for them to drain).  Query q belongs to SUB consecutive waves, which take its
// candidate pairs round-robin; same arithmetic as rescore_kernel.
__global__ __launch_bounds__(256, 6) void rescore_resident_kernel(const float* __restrict__ gal, const float* __restrict__ qry,
                                                               int32_t dp, int32_t nq, const uint32_t* __restrict__ cand_rows,
                                                               const uint32_t* __restrict__ cand_cnt, uint32_t rcap,
                                                               double* __restrict__ cand_score, uint32_t sub,
                                                               uint32_t last_row, uint32_t dbg) {
#ifndef MI_RESIDENT_PROBE
  dbg = 0u;                                                  // product build: the probe branches below are compiled out
#endif
  const int lane = threadIdx.x & 63;
  const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int nvec = dp >> 2;
  for (uint32_t t = gw; t < (uint32_t)nq * sub; t += nwaves) {
    const uint32_t q = t / sub, part = t % sub;
    const uint32_t nc = min(cand_cnt[q], rcap);
    const float4* qv = reinterpret_cast<const float4*>(qry + (uint64_t)q * dp);
    const uint32_t* rows = cand_rows + (uint64_t)q * rcap;
    double* outs = cand_score + (uint64_t)q * rcap;
    for (uint32_t c = part * 2; c < nc; c += 2 * sub) {
      const bool two = (c + 1 < nc);
      uint32_t r0 = min(rows[c], last_row), r1 = min(rows[two ? c + 1 : c], last_row);
      if (dbg & 2u) { r0 = c & 63u; r1 = (c + 1) & 63u; }                 // diagnostics: no gather (64 rows, cache-resident)
      const float4* g0 = reinterpret_cast<const float4*>(gal + (uint64_t)r0 * dp);
      const float4* g1 = reinterpret_cast<const float4*>(gal + (uint64_t)r1 * dp);
      double a0 = 0.0, a1 = 0.0;
      int v = lane;
      if (dbg & 1u) {                                                       // diagnostics: the loads without the arithmetic
        float f0 = 0.f, f1 = 0.f;
        for (; v < nvec; v += 64) {
          const float4 y0 = nt_load4(g0 + v), y1 = nt_load4(g1 + v);
          f0 += y0.x + y0.w;
          f1 += y1.x + y1.w;
        }
        a0 = f0; a1 = f1;
      }
      // two 16-byte loads per row in flight per lane (4 KiB per wave): the launch must stay within the 80 VGPRs the tile
      // kernel leaves free on a SIMD (2 x 216 of 512), and it has a whole scoring launch to move its bytes in
      for (; v + 64 < nvec; v += 128) {
        float4 x[2], y0[2], y1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          x[u] = qv[v + 64 * u];
          y0[u] = nt_load4(g0 + v + 64 * u);
          y1[u] = nt_load4(g1 + v + 64 * u);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          a0 += (double)x[u].x * (double)y0[u].x; a0 += (double)x[u].y * (double)y0[u].y;
          a0 += (double)x[u].z * (double)y0[u].z; a0 += (double)x[u].w * (double)y0[u].w;
          a1 += (double)x[u].x * (double)y1[u].x; a1 += (double)x[u].y * (double)y1[u].y;
          a1 += (double)x[u].z * (double)y1[u].z; a1 += (double)x[u].w * (double)y1[u].w;
        }
      }
      for (; v < nvec; v += 64) {
        const float4 x = qv[v];
        const float4 y0 = g0[v];
        const float4 y1 = g1[v];
        a0 += (double)x.x * (double)y0.x; a0 += (double)x.y * (double)y0.y;
        a0 += (double)x.z * (double)y0.z; a0 += (double)x.w * (double)y0.w;
        a1 += (double)x.x * (double)y1.x; a1 += (double)x.y * (double)y1.y;
        a1 += (double)x.z * (double)y1.z; a1 += (double)x.w * (double)y1.w;
      }
      for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
      }
      if (lane == 0) {
        outs[c] = a0;
        if (two) outs[c + 1] = a1;
      }
    }
  }
}

void launch_rescore_resident(const float* gal_f32, const float* qry_f32, int32_t dp, int32_t nq, const uint32_t* cand_rows,
                             const uint32_t* cand_cnt, uint32_t rcap, double* cand_score, hipStream_t stream,
                             uint32_t last_row) {
  unsigned grid = (unsigned)current_device_cus();
  // The probes of scripts/resident_probe.sh (1 = loads without the f64 arithmetic, 2 = arithmetic without the gather, 4 = a
  // quarter of the CUs host the tail; results wrong by construction) exist only in a library built with -DMI_RESIDENT_PROBE:
  // the product build has no path on which an environment variable could change the exact re-score.
#ifdef MI_RESIDENT_PROBE
  static const uint32_t dbg = [] { const char* e = getenv("MI_RESIDENT_DEBUG"); return e ? (uint32_t)atoi(e) : 0u; }();
#else
  constexpr uint32_t dbg = 0u;
#endif
  if (dbg & 4u) grid /= 4;
  uint32_t sub = 1;
  while ((uint64_t)nq * sub * 2 <= (uint64_t)grid * 4) sub *= 2;      // every wave of the grid gets a share
  hipLaunchKernelGGL(rescore_resident_kernel, dim3(grid), dim3(256), 0, stream, gal_f32, qry_f32, dp, nq, cand_rows,
                     cand_cnt, rcap, cand_score, sub, last_row, dbg);
}

// grid_x: workgroups (of 2 candidates) per query and sweep, 0 = default.  Workgroups beyond a query's count exit at once and
// queries with more candidates than one sweep covers take further sweeps.
void launch_rescore(const float* gal_f32, const float* qry_f32, int32_t dp, int32_t nq, const uint32_t* cand_rows,
                    const uint32_t* cand_cnt, uint32_t rcap, double* cand_score, hipStream_t stream, uint32_t grid_x,
                    uint32_t last_row) {
  uint32_t gx = grid_x ? grid_x : RESCORE_GRID_X;
  gx = std::max<uint32_t>(1u, std::min<uint32_t>(gx, (rcap + RESCORE_ROWS_PER_WG - 1) / RESCORE_ROWS_PER_WG));
  hipLaunchKernelGGL((rescore_kernel<1, RESCORE_ROWS_PER_WG>), dim3(gx, nq), dim3(RESCORE_THREADS), 0, stream, gal_f32,
                     qry_f32, dp, cand_rows, cand_cnt, rcap, cand_score, last_row);
}

// ------------------------------------------------------------------------------------------------
// Final order of one query's candidates: (exact f64 score desc, NaN last, ties to the lower row id) -- a strict total
// order, since the row ids of a list are distinct -- and the first k of it.  The place of a candidate is the number of
// candidates that come before it, so the threads COUNT places against the list staged in LDS (tiles of EMIT_TILE
// entries) and write every candidate straight to its output slot: no sorting network, two barriers, 6 KiB of LDS per
// workgroup whatever rescore_cap is.  Scores are staged as order-preserving 64-bit integer keys (NaN -> 0, i.e. last;
// -0.0 counted as +0.0), so one comparison is two integer compares, and with <= 128 candidates (~127 at K = 100) two
// threads share a candidate, each counting against half of the list.  Same order, bit for bit, as the bitonic sort this
// replaces (which staged rcap * 12 bytes -- 24 KiB -- per workgroup); 17 -> 8 us per 1024-query batch in isolation
// (scripts/tailbench.hip).
constexpr uint32_t EMIT_TILE = 512;
__device__ __forceinline__ uint64_t emit_key(double s) {
  if (s != s) return 0ull;                                    // NaN: after everything
  if (s == 0.0) return 0x8000000000000000ull;                 // +0.0 and -0.0 compare equal
  const uint64_t b = (uint64_t)__double_as_longlong(s);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ __launch_bounds__(256) void emit_kernel(const uint32_t* __restrict__ cand_rows,
                                                   const uint32_t* __restrict__ cand_cnt,
                                                   const double* __restrict__ cand_score, uint32_t rcap, int32_t k,
                                                   int64_t row_offset, int64_t* __restrict__ out_idx,
                                                   float* __restrict__ out_score, double* __restrict__ out_score64) {
  __shared__ __attribute__((aligned(16))) uint64_t tk[EMIT_TILE];
  __shared__ __attribute__((aligned(16))) uint32_t ti[EMIT_TILE];
  __shared__ uint32_t partial[256];
  const uint32_t q = blockIdx.x;
  const double* qs = cand_score + (uint64_t)q * rcap;
  const uint32_t* qi = cand_rows + (uint64_t)q * rcap;
  // the count and this thread's first list entry are requested together (the list has rcap >= 64 slots whatever the count:
  // reading slot threadIdx.x is in bounds; what it holds is ignored when the slot is beyond the count) -- one global round
  // trip instead of two in a kernel that is nothing but a chain of them
  const uint32_t nc_raw = cand_cnt[q];
  const bool pre_ok = threadIdx.x < rcap;
  const double s_pre = pre_ok ? qs[threadIdx.x] : 0.0;
  const uint32_t i_pre = pre_ok ? qi[threadIdx.x] : 0u;
  const uint32_t nc = min(nc_raw, rcap);
  // slots no candidate claims (fewer than k candidates): padding
  for (uint32_t i = nc + threadIdx.x; i < (uint32_t)k; i += blockDim.x) {
    out_idx[(uint64_t)q * k + i] = -1;
    if (out_score) out_score[(uint64_t)q * k + i] = -INFINITY;
    if (out_score64) out_score64[(uint64_t)q * k + i] = -INFINITY;
  }
  const uint32_t parts = (nc <= 128u) ? 2u : 1u;              // threads per candidate
  const uint32_t per = 256u / parts;                          // candidates per sweep
  const uint32_t local = threadIdx.x & (per - 1u), part = threadIdx.x / per;
  for (uint32_t i0 = 0; i0 < nc; i0 += per) {                 // one sweep unless there are more than 256 candidates
    const uint32_t i = i0 + local;
    const bool mine = i < nc;
    double a = 0.0;
    uint32_t ia = 0u;
    uint64_t ka = 0ull;
    bool have_a = false;
    uint32_t place = 0;
    for (uint32_t t0 = 0; t0 < nc; t0 += EMIT_TILE) {
      const uint32_t tn = min(EMIT_TILE, nc - t0), tn4 = (tn + 3u) & ~3u;
      __syncthreads();                                        // the previous tile (and `partial`) has been read by everybody
      for (uint32_t e = threadIdx.x; e < tn4; e += blockDim.x) {
        // padding to a multiple of four never counts: key 0 with the largest id comes before nothing
        const bool reg = (t0 == 0 && e == threadIdx.x);       // the first 256 entries of the list are in registers
        tk[e] = e < tn ? emit_key(reg ? s_pre : qs[t0 + e]) : 0ull;
        ti[e] = e < tn ? (reg ? i_pre : qi[t0 + e]) : 0xFFFFFFFFu;
      }
      partial[threadIdx.x] = 0;
      __syncthreads();
      if (mine && !have_a) {                                  // my candidate: from the staged tile when it lies in it
        if (i >= t0 && i < t0 + tn) { ka = tk[i - t0]; ia = ti[i - t0]; }
        else { ka = emit_key(qs[i]); ia = qi[i]; }
        have_a = true;
      }
      if (mine) {
        // this thread's share of the tile: `parts` contiguous chunks of whole 4-entry groups; four entries per step as three
        // 16-byte LDS reads (broadcasts), two steps in flight
        const uint32_t groups = tn4 >> 2, g0 = groups * part / parts, g1 = groups * (part + 1u) / parts;
#pragma unroll 2
        for (uint32_t e = g0 * 4u; e < g1 * 4u; e += 4) {
          const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(tk + e), k23 = *reinterpret_cast<const ulonglong2*>(tk + e + 2);
          const uint4 id4 = *reinterpret_cast<const uint4*>(ti + e);
          const uint64_t kb[4] = {k01.x, k01.y, k23.x, k23.y};
          const uint32_t ib[4] = {id4.x, id4.y, id4.z, id4.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) place += ((kb[u] > ka) || (kb[u] == ka && ib[u] < ia)) ? 1u : 0u;   // "b comes before a"
        }
      }
    }
    if (parts == 2u) {                                        // (block-uniform) combine the two halves of a candidate
      if (mine && part == 1u) partial[local] = place;
      __syncthreads();
      if (mine && part == 0u) place += partial[local];
    }
    if (mine && part == 0u && place < (uint32_t)k) {
      a = (i < 256u && i == threadIdx.x) ? s_pre : qs[i];
      out_idx[(uint64_t)q * k + place] = row_offset + (int64_t)ia;
      if (out_score) out_score[(uint64_t)q * k + place] = (float)a;
      if (out_score64) out_score64[(uint64_t)q * k + place] = a;
    }
  }
}

// (A one-launch re-score + order kernel for batches of <= 128 queries was built in round 3 and measured slower -- 39 us against
// 19.5 + 9.0 us at 70 queries, profiles/r03*: one workgroup per query gathers its candidate rows at a fraction of the rate of 64
// two-row workgroups -- and removed in round 5.)

void launch_emit(const uint32_t* cand_rows, const uint32_t* cand_cnt, const double* cand_score, uint32_t rcap,
                 int32_t nq, int32_t k, int64_t row_offset, int64_t* out_idx, float* out_score, double* out_score64,
                 hipStream_t stream) {
  hipLaunchKernelGGL(emit_kernel, dim3(nq), dim3(256), 0, stream, cand_rows, cand_cnt, cand_score, rcap, k, row_offset,
                     out_idx, out_score, out_score64);
}

// ------------------------------------------------------------------------------------------------
// K-th largest of the gathered per-shard top-K approximate values (the global top-K is contained in the
// union of the shards' top-K, so this is the exact K-th largest approximate score of the whole gallery).
__global__ __launch_bounds__(256) void kth_of_gathered_kernel(const float* __restrict__ gathered, int32_t nshards,
                                                              int64_t nq, int32_t k, float* __restrict__ out_L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t q = blockIdx.x;
  const uint32_t n = (uint32_t)nshards * (uint32_t)k;
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);
  uint32_t* hist = keys + ((n + 3u) & ~3u);               // 16-byte aligned (the select reads its bins as uint4)
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t s = i / k, e = i % k;
    keys[i] = f2key(gathered[((uint64_t)s * nq + q) * k + e]);
  }
  __syncthreads();
  const uint32_t key = block_kth_largest(keys, n, (uint32_t)k, hist);
  if (threadIdx.x == 0) out_L[q] = key2f(key);
}

void launch_kth_of_gathered(const float* gathered, int32_t nshards, int64_t nq, int32_t k, float* out_L,
                            hipStream_t stream) {
  const size_t lds = (size_t)nshards * k * 4 + 16 + 1024 * 4 + 16;
  ensure_dynamic_lds((const void*)kth_of_gathered_kernel);
  hipLaunchKernelGGL(kth_of_gathered_kernel, dim3((unsigned)nq), dim3(256), lds, stream, gathered, nshards, nq, k,
                     out_L);
}

// ------------------------------------------------------------------------------------------------
// merge of the shards' exact top-K lists by (score64 desc, idx asc); padded entries carry idx -1 / -inf.
// Every list arrives sorted in exactly that order (emit_kernel) and the ids of different shards are disjoint, so the
// place of an entry in the merged order is its place in its own list plus, for every other list, the number of entries
// that come before it there -- one binary search per (entry, other list), no sorting network.  One workgroup per query;
// the lists are staged in LDS when they fit (nshards * k * 16 bytes <= 48 KiB), else searched where they lie.
__device__ __forceinline__ bool merged_before(double sa, long long ia, double sb, long long ib) {
  // "a comes before b": higher score first, NaN last, padding (id < 0) last of all, ties to the lower id
  if (ia < 0 || ib < 0) return ib < 0 && ia >= 0;
  const bool a_nan = (sa != sa), b_nan = (sb != sb);
  if (a_nan || b_nan) return (!a_nan) || (b_nan && ia < ib);
  return (sa > sb) || (sa == sb && ia < ib);
}

template <bool IN_LDS>
__global__ __launch_bounds__(1024) void merge_kernel(const double* __restrict__ score64, const int64_t* __restrict__ idx,
                                                    int32_t nshards, int64_t nq, int32_t k, int64_t shard_stride,
                                                    int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t q = blockIdx.x;
  const uint32_t n = (uint32_t)nshards * (uint32_t)k;
  double* ls = reinterpret_cast<double*>(smem);
  long long* li = reinterpret_cast<long long*>(smem + (IN_LDS ? (size_t)n * 8 : 0));
  if (IN_LDS) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      const uint32_t sh = i / k, e = i % k;
      const uint64_t src = (uint64_t)sh * shard_stride + (uint64_t)q * k + e;
      ls[i] = score64[src];
      li[i] = idx[src];
    }
    __syncthreads();
  }
  auto entry = [&](uint32_t sh, uint32_t e, double& sc, long long& id) {
    if (IN_LDS) {
      sc = ls[sh * k + e];
      id = li[sh * k + e];
    } else {
      const uint64_t src = (uint64_t)sh * shard_stride + (uint64_t)q * k + e;
      sc = score64[src];
      id = idx[src];
    }
  };
  // positions nobody claims (fewer than k valid entries in all lists together) read as padding
  for (uint32_t i = threadIdx.x; i < (uint32_t)k; i += blockDim.x) {
    out_idx[(uint64_t)q * k + i] = -1;
    if (out_score) out_score[(uint64_t)q * k + i] = -INFINITY;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t sh = i / k, e = i % k;
    double sc;
    long long id;
    entry(sh, e, sc, id);
    if (id < 0) continue;
    uint32_t rank = e;
    for (uint32_t t = 0; t < (uint32_t)nshards && rank < (uint32_t)k; ++t) {
      if (t == sh) continue;
      uint32_t lo = 0, hi = (uint32_t)k;                   // first entry of list t that does NOT come before mine
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        double so;
        long long io;
        entry(t, mid, so, io);
        if (merged_before(so, io, sc, id)) lo = mid + 1;
        else hi = mid;
      }
      rank += lo;
    }
    if (rank < (uint32_t)k) {
      out_idx[(uint64_t)q * k + rank] = (int64_t)id;
      if (out_score) out_score[(uint64_t)q * k + rank] = (float)sc;
    }
  }
}

void launch_merge(const double* score64, const int64_t* idx, int32_t nshards, int64_t nq, int32_t k, int64_t shard_stride,
                  int64_t* out_idx, float* out_score, hipStream_t stream) {
  const size_t bytes = (size_t)nshards * (size_t)k * 16;
  // one entry per thread where possible: a thread's work is a chain of dependent look-ups (latency, not throughput)
  const unsigned threads = (unsigned)std::min<size_t>(1024, std::max<size_t>(256, round_up((size_t)nshards * (size_t)k, 64)));
  if (bytes <= 48 * 1024)
    hipLaunchKernelGGL(merge_kernel<true>, dim3((unsigned)nq), dim3(threads), bytes, stream, score64, idx, nshards, nq, k,
                       shard_stride, out_idx, out_score);
  else
    hipLaunchKernelGGL(merge_kernel<false>, dim3((unsigned)nq), dim3(threads), 0, stream, score64, idx, nshards, nq, k,
                       shard_stride, out_idx, out_score);
}

}  // namespace mi
