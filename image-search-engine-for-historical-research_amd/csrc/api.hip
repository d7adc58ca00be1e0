// Host side of the C ABI (include/mi355_retrieval.h): handle management, workspace, the chunked
// scoring schedule and the two-phase exact top-K protocol.  No torch, no numpy: plain pointers.
#include "../../include/mi355_retrieval.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "kernels.h"

using namespace mi;

static thread_local std::string g_err;
static std::atomic<int> g_default_img_f16{1};   // mi_set_global_option("image_dtype", 0 = bf16 | 1 = fp16); read at gallery creation
// mi_set_global_option("host_ingest", ...): how mi_gallery_create moves a HOST array to the device.  1 (default) = row blocks of
// ~32 MiB copied straight from the caller's (pageable) array by the runtime into two alternating device blocks, the copy of
// block i + 1 under the ingest of block i, no staging the size of the gallery; 0 = one copy of the whole array into a same-size
// staging allocation, then one ingest (rounds 1-4).  Measured at 1 005 994 x 2048 float32 (profiles/r05f_host_ingest_modes.json):
// 53.1 and 53.6 GB/s = 0.96 of the box's pinned H2D rate -- the runtime's pageable path is as fast as a pinned copy here.  A
// third mode, blocks through two pinned buffers filled by 2-8 host threads (what VERDICT r04 prescribed), reached 30-35 GB/s
// and was removed again.
static std::atomic<int> g_host_ingest{1};
// XCD shares of the tile kernel as the last handle on a device left them (common.h XccBalance): a handle created later -- or
// loaded from a file written without shares -- starts from these instead of from an even split
static std::mutex g_bal_mu;
static std::map<int, std::vector<float>> g_bal_cache;
// mi_set_global_option("keep_buffers", 0 | 1): a caller that prepares a gallery per call -- create, search, destroy: what a
// stateless matching_<method>(K, train, test) is (src/utils/nnsearch.py:687-706) -- pays hipMalloc + hipFree of the gallery's
// buffers every time (12.4 GB at 1 005 994 x 2048: 1-6 ms, more than half an ingest, and an idle GPU meanwhile).  With 1
// (default) mi_gallery_destroy hands the four buffers of a gallery of up to 16 GiB to ONE spare slot per process instead of
// freeing them, and the next gallery of exactly the same sizes on the same device takes them (every byte a search reads is
// written by the ingest or by an explicit memset; nothing depends on fresh memory).  0 frees the spare and stops keeping.
struct SpareBuffers {
  int device = -1;
  size_t f32_bytes = 0, img_bytes = 0, stat_bytes = 0;
  float* gal_f32 = nullptr;
  void* gal_img = nullptr;
  RowStat* rowstat = nullptr;
  float* gstat3 = nullptr;
};
static std::mutex g_spare_mu;
static SpareBuffers g_spare;
struct SpareArena {           // the search workspace of the handle destroyed last (one carved allocation, ~200 MB)
  int device = -1;
  size_t bytes = 0;
  void* p = nullptr;
};
static SpareArena g_spare_ws;
static std::atomic<int> g_keep_buffers{1};
static const size_t SPARE_MAX_BYTES = (size_t)16 << 30;
static void spare_release_locked() {
  if (g_spare.device < 0) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(g_spare.device);
  (void)hipFree(g_spare.gal_f32);
  (void)hipFree(g_spare.gal_img);
  (void)hipFree(g_spare.rowstat);
  (void)hipFree(g_spare.gstat3);
  (void)hipSetDevice(cur);
  g_spare = SpareBuffers();
}
static void spare_ws_release_locked() {
  if (!g_spare_ws.p) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(g_spare_ws.device);
  (void)hipFree(g_spare_ws.p);
  (void)hipSetDevice(cur);
  g_spare_ws = SpareArena();
}
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPC(expr)                                                                                   \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return fail(_e == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP,                             \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                                \
  } while (0)
#define REQUIRE(cond, msg) \
  do {                     \
    if (!(cond)) return fail(MI_ERR_INVALID, msg); \
  } while (0)

namespace {
constexpr int QB = 1024;  // queries per batch (workspace size)

struct Workspace {
  int32_t qcap = 0, kcap = 0;
  uint32_t cap = 0, rcap = 0;
  float* q_f32 = nullptr;
  void* q_img = nullptr;
  RowStat* q_stat = nullptr;
  float *thr = nullptr, *margin = nullptr, *thr2 = nullptr;
  uint32_t* qflag = nullptr;
  float* lad_tc = nullptr;
  uint32_t *lad_pack = nullptr, *lad_cnt = nullptr;
  uint32_t* cnt = nullptr;
  uint64_t* surv = nullptr;
  uint32_t* flags = nullptr;
  uint32_t* repair = nullptr;       // this workspace's own 'repair needed' word (never aliased)
  float *topvals = nullptr, *L = nullptr;
  uint32_t *cand_rows = nullptr, *cand_cnt = nullptr;
  double* cand_score = nullptr;
  // second set of the buffers the tail of a search (exact re-score + emit) reads, for the asynchronous tail: the tail of
  // batch i runs on its own stream beside the scoring launch of batch i + 1, which refills the other set
  float* q_f32_set[2] = {nullptr, nullptr};
  uint32_t *cand_rows_set[2] = {nullptr, nullptr}, *cand_cnt_set[2] = {nullptr, nullptr};
  double* cand_score_set[2] = {nullptr, nullptr};
  uint64_t* stats2 = nullptr;
  SurvRec* rec = nullptr;
  uint32_t* rec_cnt = nullptr;
  unsigned long long* dbg = nullptr;
  XccBalance* bal = nullptr;      // measured XCD shares of the tile kernel (device memory)
  uint32_t rec_cap = 4096, nseg = 0;
  std::vector<void*> allocs;
  size_t arena_bytes = 0;   // allocs[0] is one carved allocation of this size on device arena_device (ws_ensure)
  int arena_device = -1;
};
struct TmpAlloc {
  std::vector<void*> v;
  ~TmpAlloc() { for (void* p : v) (void)hipFree(p); }
  template <typename T> T* get(size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, count * sizeof(T) + 256) != hipSuccess) return nullptr;
    v.push_back(p);
    return reinterpret_cast<T*>(p);
  }
};
// what phase 1 of one batch will do (pure arithmetic on the shapes; plan_phase1)
struct P1Plan {
  int32_t nq = 0, k = 0, qpad = 0;
  bool exact = false;
  int64_t ntiles = 0, t0 = 0;
  int32_t samp_r = 0;              // > 0: single-launch schedule on the threshold sample, speculative rank
  uint32_t first_cnt = 0;
  float gamma = 0.f;
  int32_t boot_ksplit = 1;
  bool sample_f32 = false;         // the bootstrap launch stores bare 4-byte scores
  bool thr_kernel = false;         // sample_threshold_kernel takes the thresholds (else select_maintain mode 0)
  int32_t lad_r = 0;               // ladder level (sample rank), 0 = off
  bool zero_scores = false;        // the query ingest writes zeros for the K-split bootstrap to add onto
};
}  // namespace

struct mi_gallery {
  int device = 0;
  int64_t n = 0, npad = 0, row_offset = 0;
  int64_t cap = 0;          // allocated rows (== n unless created with mi_gallery_create_empty)
  int32_t d = 0, dp = 0, norm_mode = 0;
  int img_f16 = 1;          // 16-bit image element type of the gallery AND of the query batches searched on it
  float* gal_f32 = nullptr;
  void* gal_img = nullptr;
  RowStat* rowstat = nullptr;
  float* gstat3 = nullptr;
  // bootstrap sample image of the speculative schedule (built lazily, rebuilt when rows were appended)
  void* samp_img = nullptr;
  int64_t samp_tiles = 0, samp_for_n = -1;
  int64_t hbm_bytes = 0;
  size_t buf_bytes[3] = {0, 0, 0};                 // gal_f32 / gal_img / rowstat as allocated (what a spare slot is matched by)
  // XCD shares read from the prepared-gallery file (MI355GAL trailer) / snapshotted for the next save
  float file_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool file_w_valid = false;
  hipStream_t stream = nullptr;
  Workspace ws;
  // option "workspace_slot": the phase API of batch i + 1 may run in the other workspace while batch i waits for its
  // collectives (sharded search, two batches in flight).  `ws` is always the active one; `ws_alt` the parked one.  The
  // sticky flags, the statistics, the kernel clocks and the XCD shares are ONE set: the second workspace aliases them.
  Workspace ws_alt;
  int ws_slot = 0;
  // options
  int chunk0_tiles = 0 /* 0 = default, bootstrap_tiles() */, chunk_growth = 8, exact_fallback = 1, force_exact = 0,
      speculative = 1, rescore_grid_x = 0, spec_max_ratio = 160;
  int device_repair = -1;       // -1 = by batch size (off for <= 128 queries), 0 / 1 = never / always launch the conditional repair pass
  int small_batch_kernel = 1;   // batches of <= 128 queries are scored by stream_select.hip (HBM-bound kernel)
  int xcc_balance = 1;          // split the gallery tiles over the XCDs by their measured speed (common.h XccBalance)
  int ladder = 1;               // in-launch threshold ladder of the tile kernel (common.h QueryState::lad_*): 0 = off, 1 = on
  int boot_ksplit = 1;          // small batches: K-split bootstrap launch (kernels.h ScoreArgs::ksplit); 0 = one workgroup per tile
  int stream_tail = 1;          // host entry points with more than one batch of queries: deferred tail between their batches
  // asynchronous tail (option "async_tail", device entry point mi_knn_search_device only): the exact re-score + emit of a
  // batch run on tail_stream behind an event, beside the scoring launch of the NEXT batch (the tile kernel leaves 80
  // VGPRs per SIMD lane and no LDS: exactly one 70-register re-score wave per SIMD fits next to its two); results are
  // valid after mi_search_join
  int async_tail = 0, tail_set = 0;
  hipStream_t tail_stream = nullptr;
  hipEvent_t ev_p1[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
  bool ev_tail_valid[2] = {false, false};
  hipEvent_t gate_before_scoring = nullptr;   // async_tail 2: the filtered scoring launch of a batch waits for this event
  // async_tail 3 ("deferred"): the tail of batch i is ENQUEUED by the search call of batch i + 1, after that batch's query
  // ingest / bootstrap / threshold launches and right before its scoring launch, so that the re-score gather shares the
  // device with the power-bound scoring launch only -- not with the bootstrap, which wants the same memory system
  // (mode 1 starts the tail as soon as phase 1 is done, i.e. beside the next batch's bootstrap: 63 -> 214 us).
  struct PendingTail {
    bool valid = false;
    int32_t b = 0, k = 0;
    int set = 0;
    int slot = 0;                             // workspace slot (ws_slot) the batch ran in
    int64_t* out_idx = nullptr;
    float* out_score = nullptr;
    double* out_score64 = nullptr;
  } pending;
  hipEvent_t ev_pre = nullptr;                // recorded on the caller's stream right before the scoring launch
  int qnorm_override = -1;  // device entry points: normalise queries with this mi_norm instead of the gallery's (-1 = off)
  uint32_t surv_cap = 12288, rescore_cap = 2048;
  // stats
  mi_search_stats stats{};
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  std::vector<float> launch_ms_log;   // duration of every timed scoring launch since the last statistics reset (capped)
  hipStream_t ev_stream = nullptr;
  // grow-only device staging of the host entry point mi_knn_search (queries in, results out): a hipMalloc / hipFree
  // pair per call costs more than a single-query search
  void* io_buf[3] = {nullptr, nullptr, nullptr};
  size_t io_cap[3] = {0, 0, 0};
  // diffusion state (offline matrix rows kept on the device for the online stage)
  int32_t* dif_ids = nullptr;
  float* dif_vals = nullptr;
  int32_t dif_T = 0;
  std::mutex mu;
};

static int ws_free(Workspace& ws) {
  if (ws.allocs.size() == 1 && ws.arena_bytes && g_keep_buffers.load()) {
    // hipFree waits for the device before it gives memory back; so does this: the next owner clears the allocation, and
    // launches of this handle on a caller's (non-blocking) stream may still be reading it when a workspace is re-built
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(g_spare_mu);
    spare_ws_release_locked();
    g_spare_ws.device = ws.arena_device, g_spare_ws.bytes = ws.arena_bytes, g_spare_ws.p = ws.allocs[0];
  } else {
    for (void* p : ws.allocs) (void)hipFree(p);
  }
  ws = Workspace();
  return MI_OK;
}

static int flush_pending_tail(mi_gallery* g, hipStream_t s, bool beside_scoring);

static int ws_ensure(mi_gallery* g, int32_t k) {
  Workspace& ws = g->ws;
  if (ws.qcap >= QB && ws.kcap >= k && ws.cap == g->surv_cap && ws.rcap == g->rescore_cap) return MI_OK;
  if (g->pending.valid) {          // a deferred tail (async_tail 3) still reads the buffers about to be rebuilt
    const int rc = flush_pending_tail(g, nullptr, false);
    if (rc != MI_OK) return rc;
    HIPC(hipStreamSynchronize(g->tail_stream));
  }
  const int32_t kcap = std::max<int32_t>(k, std::max(ws.kcap, g->ws_alt.kcap));
  // (re)allocation of the active workspace.  The parked one survives only if this is the active one's FIRST allocation
  // (then it owns the shared flags / statistics and the new one aliases them); any other re-allocation may free what
  // the parked one aliases, so it is dropped too and rebuilt when it is next switched in.
  const bool first_alloc = ws.allocs.empty();
  ws_free(ws);
  if (!first_alloc || g->ws_alt.kcap < kcap || g->ws_alt.cap != g->surv_cap || g->ws_alt.rcap != g->rescore_cap)
    ws_free(g->ws_alt);
  ws.qcap = QB;
  ws.kcap = kcap;
  ws.cap = g->surv_cap;
  ws.rcap = g->rescore_cap;
  // ONE allocation, carved: ~30 hipMalloc / hipFree pairs per handle cost a caller that prepares a gallery per call (create,
  // search, destroy) 4-5 ms, several times its search.  Two passes over the same list: sizes first, pointers second.  The
  // buffers that must start as zeros come first (kept together; the whole allocation is cleared anyway).
  ws.rec_cap = 4096;          // records per wave segment and launch (K = 1000 at 1M rows needs ~1800)
  ws.nseg = gemm_select_grid() * 8;
  char* base = nullptr;
  size_t total = 0;
  for (int pass = 0; pass < 2; ++pass) {
    total = 0;
    auto carve = [&](auto** ptr, size_t count) {
      using T = typename std::remove_pointer<typename std::remove_pointer<decltype(ptr)>::type>::type;
      if (base) *ptr = reinterpret_cast<T*>(base + total);
      total += (count * sizeof(T) + 511) / 256 * 256;
    };
#define A(ptr, count) carve(&ws.ptr, (count))
    A(flags, 4);
    A(repair, 4);
    A(stats2, 3 * (size_t)QB);     // per query: (survivors, candidates) accumulators, then [2 QB ..) in-kernel repairs -- one writer each, no atomics
    A(dbg, (size_t)ws.nseg * 8);
    A(cand_cnt, QB);
    A(cand_cnt_set[1], QB);
    A(q_f32, (size_t)QB * g->dp);
    {
      __hip_bfloat16* tmp = nullptr;
      carve(&tmp, (size_t)QB * g->dp);
      ws.q_img = tmp;
    }
    A(q_stat, QB);
    A(thr, QB);
    A(margin, QB);
    A(thr2, QB);
    A(qflag, QB);
    A(lad_tc, QB);
    A(lad_pack, QB);
    A(lad_cnt, QB);
    A(cnt, (size_t)QB * CNT_STRIDE);
    A(surv, (size_t)QB * ws.cap);
    A(topvals, (size_t)QB * kcap);
    A(L, QB);
    A(cand_rows, (size_t)QB * ws.rcap);
    A(cand_score, (size_t)QB * ws.rcap);
    A(rec, (size_t)ws.nseg * ws.rec_cap);
    A(rec_cnt, ws.nseg);
    A(bal, 1);
    A(q_f32_set[1], (size_t)QB * g->dp);
    A(cand_rows_set[1], (size_t)QB * ws.rcap);
    A(cand_score_set[1], (size_t)QB * ws.rcap);
#undef A
    if (pass == 0) {
      void* v = nullptr;
      {
        std::lock_guard<std::mutex> lock(g_spare_mu);
        if (g_spare_ws.p && g_spare_ws.device == g->device && g_spare_ws.bytes == total) {
          v = g_spare_ws.p;
          g_spare_ws = SpareArena();
        }
      }
      if (!v) HIPC(hipMalloc(&v, total));
      ws.allocs.push_back(v);
      ws.arena_bytes = total;
      ws.arena_device = g->device;
      base = reinterpret_cast<char*>(v);
    }
  }
  ws.q_f32_set[0] = ws.q_f32;
  ws.cand_rows_set[0] = ws.cand_rows;
  ws.cand_cnt_set[0] = ws.cand_cnt;
  ws.cand_score_set[0] = ws.cand_score;
  // the whole allocation starts as zeros, recycled or fresh (~0.05 ms): nothing may depend on what a previous handle left
  HIPC(hipMemset(base, 0, total));
  {
    // the XCD shares start from what is known: the file's, else this process's last ones on the device, else an even split
    XccBalance hb;
    init_xcc_balance_host(&hb);
    if (g->file_w_valid) {
      init_xcc_balance_from(&hb, g->file_w);
    } else {
      std::lock_guard<std::mutex> l(g_bal_mu);
      auto it = g_bal_cache.find(g->device);
      if (it != g_bal_cache.end()) init_xcc_balance_from(&hb, it->second.data());
    }
    HIPC(hipMemcpy(ws.bal, &hb, sizeof hb, hipMemcpyHostToDevice));
  }
  if (g->ws_alt.flags) {                   // one set of flags / statistics / clocks / XCD shares per handle
    ws.flags = g->ws_alt.flags;
    ws.stats2 = g->ws_alt.stats2;
    ws.dbg = g->ws_alt.dbg;
    ws.bal = g->ws_alt.bal;
  }
  return MI_OK;
}

// the shares as the launches so far left them -> process cache (and out_w8, if given); false = none measured yet
static bool snapshot_balance(const mi_gallery* g, float* out_w8) {
  const Workspace& sw = g->ws.bal ? g->ws : g->ws_alt;
  if (!sw.bal) return false;
  XccBalance hb;
  if (hipMemcpy(&hb, sw.bal, sizeof hb, hipMemcpyDeviceToHost) != hipSuccess || hb.launches == 0) return false;
  {
    std::lock_guard<std::mutex> l(g_bal_mu);
    g_bal_cache[g->device].assign(hb.w, hb.w + 8);
  }
  if (out_w8) memcpy(out_w8, hb.w, sizeof hb.w);
  return true;
}

static QueryState make_state(const Workspace& ws) {
  QueryState st;
  st.thr = ws.thr;
  st.margin = ws.margin;
  st.cnt = ws.cnt;
  st.surv = ws.surv;
  st.flags = ws.flags;
  st.repair = ws.repair;
  st.thr2 = ws.thr2;
  st.qflag = ws.qflag;
  st.lad_tc = ws.lad_tc;
  st.lad_pack = ws.lad_pack;
  st.lad_cnt = ws.lad_cnt;
  st.cap = ws.cap;
  return st;
}

static void prof_begin(mi_gallery* g, hipStream_t s, size_t* slot) {
  *slot = (size_t)-1;
  if (!g->profile) return;
  if (g->ev_used == g->ev_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g->ev_pool.emplace_back(a, b);
  }
  *slot = g->ev_used++;
  g->ev_stream = s;
  // the pair rides in the dispatch packet of the scoring launch itself (hipExtLaunchKernelGGL): begin / end timestamps of
  // that kernel, no barrier packets of their own on the stream
  set_launch_events(g->ev_pool[*slot].first, g->ev_pool[*slot].second);
}
static void prof_end(mi_gallery* g, hipStream_t s, size_t slot) {
  (void)g; (void)s; (void)slot;
  hipEvent_t a, b;
  take_launch_events(&a, &b);      // a launcher that did not take them (f32 scorer) leaves nothing behind
}
static void prof_collect(mi_gallery* g) {
  for (size_t i = 0; i < g->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g->ev_pool[i].second) == hipSuccess &&
        hipEventElapsedTime(&ms, g->ev_pool[i].first, g->ev_pool[i].second) == hipSuccess) {
      g->stats.gemm_ms += ms;
      g->stats.gemm_launches += 1;
      if (g->launch_ms_log.size() < (size_t)1 << 16) g->launch_ms_log.push_back(ms);
    }
  }
  g->ev_used = 0;
}

// Rank of the sample score the speculative threshold is taken from.  The number of the shard's K best rows that fall
// into a uniform sample of n_s of its N rows is Binomial(K, n_s / N) ~ Poisson(lambda = K n_s / N); the r-th largest
// sample score exceeds the shard's K-th best (and the threshold fails its verification) iff that number is >= r.
// r = the smallest rank whose tail probability is <= 1e-7 per query (1e-4 per 1024-query batch: a failed query costs
// a repair pass, never a wrong answer).  The kernels take `score(r) - margin` as the threshold, so the certificate's
// band below the K-th score is covered whatever the image type.
static int32_t spec_rank(double lambda) {
  double pmf = std::exp(-lambda), cdf = pmf;
  int32_t r = 1;
  while (1.0 - cdf > 1e-7 && r < (1 << 20)) {
    pmf *= lambda / r;
    cdf += pmf;
    ++r;
  }
  return r;
}

// Rows of the bootstrap chunk / threshold sample, in tiles of 256: 8192 rows unless option "chunk0_tiles" says otherwise
// (2048 and 4096 rows have their own threshold kernels).  Measured on the 125 750-row shards of an 8-way 1M gallery, same
// box: a 2048-row sample saves 21 us of bootstrap + selection and costs 38 us in the scoring launch (twice the survivors
// in its filter); 4096 and 8192 rows tie.  So the size does not follow the shard size.
static int64_t bootstrap_tiles(const mi_gallery* g) { return g->chunk0_tiles > 0 ? g->chunk0_tiles : 32; }

// (re)build the bootstrap sample image for the current number of rows
static int ensure_sample(mi_gallery* g, int64_t tiles, hipStream_t s) {
  if (g->samp_img && g->samp_tiles == tiles && g->samp_for_n == g->n) return MI_OK;
  if (!g->samp_img || g->samp_tiles != tiles) {
    (void)hipFree(g->samp_img);
    g->samp_img = nullptr;
    HIPC(hipMalloc(&g->samp_img, (size_t)tiles * TILE * g->dp * 2 + 256));
    g->samp_tiles = tiles;
  }
  launch_build_sample(g->gal_img, g->samp_img, g->n, tiles * TILE, g->dp, s);
  HIPC(hipGetLastError());
  g->samp_for_n = g->n;
  return MI_OK;
}

// ---- phase 1 for one batch (nq <= QB): query ingest, chunked scoring + threshold maintenance ----------
// In three parts: a PLAN (pure arithmetic: which schedule, which sample rank), the PRE part -- everything that depends on the
// queries but not on a finished scoring launch: query ingest, bootstrap launch on the sample image, thresholds -- and the MAIN
// part (scoring launches, scatter, maintain, repair).  (Round 4 ran the pre part of batch i + 1 on a stream of its own beside
// the main part of batch i -- "lookahead": measured at no gain, profiles/r04*_lookahead*; removed in round 5.)
static int flush_pending_tail(mi_gallery* g, hipStream_t s, bool beside_scoring);


static P1Plan plan_phase1(const mi_gallery* g, const Workspace& ws, int32_t nq, int32_t k, bool exact) {
  P1Plan pl;
  pl.nq = nq;
  pl.k = k;
  pl.exact = exact;
  pl.qpad = (int32_t)round_up(nq, TILE);
  pl.ntiles = g->npad / TILE;
  // bootstrap chunk: stored completely (no threshold yet); must hold >= K rows and fit the survivor buffer
  int64_t t0 = std::max<int64_t>(bootstrap_tiles(g), (2 * (int64_t)k + TILE - 1) / TILE);
  t0 = std::min<int64_t>(t0, ws.cap / TILE);
  t0 = std::min<int64_t>(t0, pl.ntiles);
  pl.t0 = t0;
  // Single-launch schedule?  The speculative threshold is an order statistic of the scores of a SAMPLE: t0 * 256
  // rows drawn evenly (one hashed draw per stratum) into their own small image, so that the order in which the shard
  // was ingested cannot bias it.  With n_s sampled rows the shard's K-th largest score sits near sample rank
  // lambda = K * n_s / N; the r-th largest sample score with r = spec_rank(lambda) lies below it except with
  // probability 1e-7 per query (Poisson tail) and keeps the expected survivors at r * N / n_s.  The sample entries
  // are dropped once the threshold is taken (the scoring launch visits every tile, sample rows included).
  if (g->speculative && !exact && pl.ntiles >= 2 * t0 && g->n / (t0 * TILE) <= g->spec_max_ratio) {
    const double lambda = (double)k * (double)(t0 * TILE) / (double)g->n;
    const int32_t r = spec_rank(lambda);
    if (r < k) pl.samp_r = r;
  }
  pl.first_cnt = (uint32_t)(pl.samp_r > 0 ? t0 * TILE : std::min<int64_t>(g->n, t0 * TILE));
  pl.gamma = 2.0f * (float)g->dp * 5.9604645e-08f;  // 2 * dp * 2^-24 (f32 accumulation, doubled)
  pl.thr_kernel = pl.samp_r > 0 && sample_threshold_applies(pl.first_cnt, k, pl.samp_r);
  // small batches on the sample schedule: the bootstrap launch splits K over several workgroups per (sample tile, query
  // group) and adds its partial scores onto zeros that the query ingest writes (ScoreArgs::ksplit)
  if (pl.samp_r > 0 && !exact && g->boot_ksplit && g->small_batch_kernel && pl.thr_kernel && g->dp <= 4096) {
    const int64_t wgs = t0 * ((nq + 63) / 64);                       // bootstrap workgroups of a batch of <= 512 queries
    if (nq <= 512)
      while (pl.boot_ksplit < 8 && wgs * pl.boot_ksplit * 2 <= 256 && (g->dp / SLICE_K) % (pl.boot_ksplit * 2) == 0)
        pl.boot_ksplit *= 2;
  }
  // sample-based schedule with the dedicated threshold kernel: the bootstrap launch (stream_select MODE 2) stores bare
  // 4-byte scores, all that kernel reads (half the bytes written and read back; ScoreArgs::scores_only)
  pl.sample_f32 = pl.thr_kernel && g->small_batch_kernel;
  // in-launch ladder: a tighter sample order statistic (rank j < r) becomes a rigorous threshold once K rows above it have
  // been counted during the launch.  In units of N / n_s rows: score(j) has expected rank j in the shard and is validated
  // after the fraction lambda / j of the rows (lambda = K n_s / N), so the survivors are ~ (lambda / j) r + (1 - lambda / j) j,
  // smallest at j = sqrt(lambda r) (3 at N = 1M, K = 100: 1500 -> 900 survivors per query)
  if (g->ladder && pl.samp_r > 1 && pl.thr_kernel) {
    const double lambda = (double)k * (double)(t0 * TILE) / (double)g->n;
    int32_t lr = (int32_t)std::lround(std::sqrt(lambda * pl.samp_r));
    pl.lad_r = std::max<int32_t>(1, std::min<int32_t>(lr, pl.samp_r - 1));
  }
  return pl;
}

// one scoring launch (+ the scatter of its records) of a phase-1 schedule
static void p1_score_launch(mi_gallery* g, Workspace& ws, const QueryState& st, const P1Plan& pl, int64_t tile_from, int64_t ntile,
                            bool first_chunk, const uint32_t* cond, bool profile_it, bool on_sample, bool ladder_on,
                            hipStream_t s);

// PRE part.  With pl.samp_r > 0: query ingest + per-query state, bootstrap launch on the sample image, thresholds (+ ladder
// levels).  Otherwise (chunk schedules): the query ingest alone.
static int p1_pre(mi_gallery* g, Workspace& ws, const P1Plan& pl, const void* q_src, int q_dtype, int64_t q_rs, int64_t q_cs,
                  int q_norm, hipStream_t s) {
  QueryState st = make_state(ws);
  if (pl.samp_r > 0) {
    const int rc = ensure_sample(g, pl.t0, s);
    if (rc != MI_OK) return rc;
  }
  int32_t boot_ksplit = pl.boot_ksplit;
  // query ingest (normalise, f32 rows, 16-bit image, rounding norms) and the per-query search state in one launch
  if (!launch_ingest_queries(q_src, q_dtype, pl.nq, g->d, q_rs, q_cs, q_norm, ws.q_f32, ws.q_img, g->img_f16, ws.q_stat, g->dp,
                             pl.qpad, g->gstat3, pl.gamma, pl.exact ? 0 : 1, pl.first_cnt, st, s,
                             boot_ksplit > 1 ? pl.first_cnt : 0u)) {
    boot_ksplit = 1;
    launch_ingest(q_src, q_dtype, pl.nq, g->d, q_rs, q_cs, q_norm, ws.q_f32, ws.q_img, g->img_f16, ws.q_stat, g->dp, pl.qpad, s);
    launch_init_query_state(ws.q_stat, g->gstat3, pl.nq, pl.qpad, pl.gamma, pl.exact ? 0 : 1, pl.first_cnt, st, s);
  }
  if (pl.samp_r > 0) {
    P1Plan p2 = pl;
    p2.boot_ksplit = boot_ksplit;
    p1_score_launch(g, ws, st, p2, 0, pl.t0, true, nullptr, false, true, false, s);     // bootstrap on the sample image
    if (pl.thr_kernel)
      launch_sample_threshold(st, pl.nq, pl.k, pl.samp_r, pl.first_cnt, s, pl.lad_r, pl.sample_f32 ? 1 : 0,
                              (pl.sample_f32 && boot_ksplit > 1) ? 0.5f : 0.f);
    else
      launch_select_maintain(st, pl.nq, pl.k, 0, ws.topvals, ws.L, ws.stats2, pl.samp_r, 1, 0, nullptr, s);
  }
  HIPC(hipGetLastError());
  return MI_OK;
}

static void p1_score_launch(mi_gallery* g, Workspace& ws, const QueryState& st, const P1Plan& pl, int64_t tile_from, int64_t ntile,
                            bool first_chunk, const uint32_t* cond, bool profile_it, bool on_sample, bool ladder_on,
                            hipStream_t s) {
  const int64_t rows0 = tile_from * TILE, rows1 = std::min<int64_t>(g->n, (tile_from + ntile) * TILE);
  if (pl.exact) {
    ExactArgs a;
    a.gal_f32 = g->gal_f32;
    a.qry_f32 = ws.q_f32;
    a.dp = g->dp;
    a.row0 = rows0;
    a.row1 = rows1;
    a.n = g->n;
    a.nq = pl.nq;
    a.st = st;
    launch_exact_select(a, first_chunk, s);
    return;
  }
  ScoreArgs a;
  a.gal_img = on_sample ? g->samp_img : g->gal_img;
  a.qry_img = ws.q_img;
  a.img_f16 = g->img_f16;
  a.nslices = g->dp / SLICE_K;
  a.tile0 = (int32_t)tile_from;
  a.ntiles = (int32_t)ntile;
  a.nqt = pl.qpad / TILE;
  a.n = on_sample ? ntile * TILE : g->n;
  a.nq = pl.nq;
  a.small_batch_kernel = g->small_batch_kernel;
  a.rec = ws.rec;
  a.rec_cnt = ws.rec_cnt;
  a.rec_cap = ws.rec_cap;
  a.cond = cond;
  a.bal = g->xcc_balance ? ws.bal : nullptr;
  a.lad_k = ladder_on ? pl.k : 0;
  a.scores_only = (on_sample && first_chunk && pl.sample_f32) ? 1 : 0;
  a.ksplit = a.scores_only ? pl.boot_ksplit : 1;
  a.dbg = ws.dbg;
  a.st = st;
  profile_it = profile_it && !first_chunk;      // the roofline is quoted on the filtered scoring launches only
  size_t slot = (size_t)-1;
  if (profile_it) prof_begin(g, s, &slot);
  launch_gemm_select(a, first_chunk, s);
  if (profile_it) prof_end(g, s, slot);
  if (profile_it && g->profile) {
    const double rows = (double)(rows1 - rows0);
    g->stats.gemm_flops += 2.0 * pl.nq * rows * g->d;
    g->stats.gemm_bytes += rows * g->d * 2.0 + (double)pl.nq * g->d * 2.0;
  }
  if (!first_chunk)
    launch_scatter_records(ws.rec, ws.rec_cnt, ws.rec_cap, ws.nseg, st, cond, s,
                           (g->xcc_balance && !on_sample && !stream_select_applies(a)) ? ws.bal : nullptr, ws.dbg,
                           (uint32_t)ntile, pl.nq);
}

// MAIN part: the filtered scoring launch(es) with their scatter / maintain launches, the conditional repair pass.
// fuse_cand: the final maintain launch also writes the candidate lists (single-shard search: its L is the global one).
// caller_checks_flags: the caller synchronises, reads the sticky flags itself and answers a flagged batch again (the host
// entry points): small batches then launch no device-side repair pass.  The asynchronous device / phase entry points pass
// false -- their callers may never look at the flags, so a failed speculative threshold is repaired on the device.
static int p1_main(mi_gallery* g, Workspace& ws, const P1Plan& pl, hipStream_t s, bool fuse_cand, bool caller_checks_flags) {
  uint32_t* fc_rows = fuse_cand ? ws.cand_rows : nullptr;
  uint32_t* fc_cnt = fuse_cand ? ws.cand_cnt : nullptr;
  QueryState st = make_state(ws);
  const int32_t nq = pl.nq, k = pl.k;
  const int64_t ntiles = pl.ntiles, t0 = pl.t0;
  const bool exact = pl.exact;
  if (pl.samp_r > 0) {
    if (g->gate_before_scoring) {               // the previous batch's tail ran beside this batch's ingest / bootstrap
      HIPC(hipStreamWaitEvent(s, g->gate_before_scoring, 0));
      g->gate_before_scoring = nullptr;
    }
    if (g->pending.valid) {                     // async_tail 3: the previous batch's tail starts beside THIS scoring launch
      const int rc = flush_pending_tail(g, s, /*beside_scoring=*/true);
      if (rc != MI_OK) return rc;
    }
    // every tile, one launch
    p1_score_launch(g, ws, st, pl, 0, ntiles, false, nullptr, true, false, pl.lad_r > 0 && pl.thr_kernel, s);
    // repair pass for queries whose speculative threshold failed verification (1e-7 per query): conditional on the device
    // word *ws.repair, i.e. three early-exit launches in the (overwhelmingly) common case, and no host round trip.  Batches
    // of <= 128 queries -- the reference's own shapes, one query online and 70 per test set, where three empty launches
    // are 1 % of the batch and a failure has probability <= 1e-5 -- do without.  A HOST entry point synchronises anyway, sees
    // FLAG_SPEC_FAIL and answers the batch again by the rigorous schedule (mode 2).  The asynchronous device and phase entry
    // points -- whose callers (sharded protocol, pipelined streams, alpha-QE re-search) need not read flags to get a
    // complete answer -- take mode 3 below.  Option "device_repair" overrides.
    // Round 4, second half: small batches on the asynchronous entry points launch no repair kernels either -- the workgroup
    // of a failed query repairs it inside the maintain launch (repair mode 3, select.hip SCAN: a scan of the shard's stored
    // rows by that one workgroup, ~0.1 s per 1 M rows, once per 10^7 queries).  "device_repair" = 1 still forces the launches.
    const bool small = nq <= STREAM_MAX_QUERIES;     // (in-kernel repair up to 1024 queries: measured slower, profiles/r04q_*)
    const bool repair_pass = g->device_repair < 0 ? !small : g->device_repair != 0;
    const int rep_mode = repair_pass ? 0 : ((small && !caller_checks_flags && g->device_repair < 0) ? 3 : 2);
    const RepairScan scan{g->gal_f32, ws.q_f32, g->dp, g->n, ws.stats2 + 2 * (size_t)QB};
    launch_select_maintain(st, nq, k, 1, ws.topvals, ws.L, ws.stats2, 0, 1, rep_mode, nullptr, s, fc_rows, fc_cnt, ws.rcap,
                           rep_mode == 3 ? &scan : nullptr);
    if (repair_pass) {
      const uint32_t* cond = ws.repair;
      p1_score_launch(g, ws, st, pl, 0, ntiles, false, cond, false, false, false, s);
      launch_select_maintain(st, nq, k, 1, ws.topvals, ws.L, ws.stats2, 0, 1, 1, cond, s, fc_rows, fc_cnt, ws.rcap);
    }
    HIPC(hipGetLastError());
    return MI_OK;
  }
  if (g->gate_before_scoring) {                 // other schedules: no bootstrap worth overlapping, wait up front
    HIPC(hipStreamWaitEvent(s, g->gate_before_scoring, 0));
    g->gate_before_scoring = nullptr;
  }
  // chunk schedule (shards too small or too large for the sample-based single launch): thresholds come from the rows
  // scored and kept so far; once those are >= 1/160 of the shard the rest goes out as one speculative launch
  // chunk boundaries (cumulative tiles): t0, t0*g, then x max(2, g/2) per step (thresholds keep tightening as the
  // sample grows; a 10M-row shard needs more steps than a 1M-row one); a tail shorter than half a step is merged
  const int64_t gr = std::max(1, g->chunk_growth);
  const int64_t gr2 = std::max<int64_t>(2, gr / 2);
  int64_t bound = t0;
  int64_t t = 0, len = t0;
  bool first = true;
  bool spec_next = false, spec_cur = false;
  while (t < ntiles) {
    int64_t cur = std::min<int64_t>(len, ntiles - t);
    if (spec_next) cur = ntiles - t;
    spec_cur = spec_next;
    p1_score_launch(g, ws, st, pl, t, cur, first, nullptr, true, false, false, s);
    t += cur;
    if (t >= ntiles) {
      launch_select_maintain(st, nq, k, 1, ws.topvals, ws.L, ws.stats2, 0, spec_cur ? 1 : 0, 0, nullptr, s, fc_rows, fc_cnt,
                             ws.rcap);
      if (spec_cur) {
        const uint32_t* cond = ws.repair;
        p1_score_launch(g, ws, st, pl, 0, ntiles, false, cond, false, false, false, s);
        launch_select_maintain(st, nq, k, 1, ws.topvals, ws.L, ws.stats2, 0, 1, 1, cond, s, fc_rows, fc_cnt, ws.rcap);
      }
    } else {
      int32_t spec_r = 0;
      const int64_t n_seen = std::min<int64_t>(g->n, t * TILE);
      if (g->speculative && !exact && g->n / n_seen <= 160) {
        const double lambda = (double)k * (double)n_seen / (double)g->n;
        const int32_t r = spec_rank(lambda);
        if (r < k) spec_r = r;
      }
      spec_next = spec_r > 0;
      launch_select_maintain(st, nq, k, 0, ws.topvals, ws.L, ws.stats2, spec_r, 0, 0, nullptr, s);
    }
    first = false;
    if (gr == 1) {
      len = t0;
    } else {
      bound = (bound == t0) ? t0 * gr : bound * gr2;
      len = std::max<int64_t>(1, bound - t);
      if (ntiles - (t + len) < len / 2) len = ntiles - t;
    }
  }
  HIPC(hipGetLastError());
  return MI_OK;
}

static int phase1_batch(mi_gallery* g, const void* q_src, int q_dtype, int64_t q_rs, int64_t q_cs, int q_norm,
                        int32_t nq, int32_t k, bool exact, hipStream_t s, bool fuse_cand = false,
                        bool caller_checks_flags = false) {
  Workspace& ws = g->ws;
  const P1Plan pl = plan_phase1(g, ws, nq, k, exact);
  const int rc = p1_pre(g, ws, pl, q_src, q_dtype, q_rs, q_cs, q_norm, s);
  if (rc != MI_OK) return rc;
  return p1_main(g, ws, pl, s, fuse_cand, caller_checks_flags);
}

// ---- phase 2 for one batch: candidates within the margin of L, exact f64 re-score, sorted emit --------
static int phase2_batch(mi_gallery* g, int32_t nq, int32_t k, const float* L_dev, int64_t* out_idx, float* out_score,
                        double* out_score64, hipStream_t s, bool have_cand = false, bool resident = false,
                        Workspace* wsp = nullptr) {
  Workspace& ws = wsp ? *wsp : g->ws;
  QueryState st = make_state(ws);
  if (!have_cand) launch_select_candidates(st, nq, L_dev, ws.cand_rows, ws.cand_cnt, ws.rcap, ws.stats2, s);
  const uint32_t last_row = (uint32_t)std::max<int64_t>(0, g->n - 1);
  if (resident) launch_rescore_resident(g->gal_f32, ws.q_f32, g->dp, nq, ws.cand_rows, ws.cand_cnt, ws.rcap, ws.cand_score, s,
                                        last_row);
  else launch_rescore(g->gal_f32, ws.q_f32, g->dp, nq, ws.cand_rows, ws.cand_cnt, ws.rcap, ws.cand_score, s,
                      (uint32_t)g->rescore_grid_x, last_row);
  launch_emit(ws.cand_rows, ws.cand_cnt, ws.cand_score, ws.rcap, nq, k, g->row_offset, out_idx, out_score,
              out_score64, s);
  HIPC(hipGetLastError());
  return MI_OK;
}

// async_tail 3: enqueue the deferred tail (exact re-score + emit of the batch recorded in g->pending) on the tail stream.
// beside_scoring: called from phase1_batch right before the scoring launch of the NEXT batch on `s` -- the tail then also
// waits for everything enqueued on `s` so far (that batch's query ingest, bootstrap, thresholds), so that it runs beside the
// scoring launch and nothing else.  Otherwise (join, a call that cannot defer) it only waits for its own phase 1.
static int flush_pending_tail(mi_gallery* g, hipStream_t s, bool beside_scoring) {
  if (!g->pending.valid) return MI_OK;
  const mi_gallery::PendingTail p = g->pending;
  g->pending.valid = false;
  Workspace& ws = (p.slot == g->ws_slot) ? g->ws : g->ws_alt;     // the workspace the batch ran in (option "workspace_slot")
  float* q_keep = ws.q_f32;
  uint32_t* rows_keep = ws.cand_rows;
  uint32_t* cnt_keep = ws.cand_cnt;
  double* sc_keep = ws.cand_score;
  ws.q_f32 = ws.q_f32_set[p.set];
  ws.cand_rows = ws.cand_rows_set[p.set];
  ws.cand_cnt = ws.cand_cnt_set[p.set];
  ws.cand_score = ws.cand_score_set[p.set];
  int rc = MI_OK;
  hipError_t e = hipStreamWaitEvent(g->tail_stream, g->ev_p1[p.set], 0);
  if (e == hipSuccess && beside_scoring) {
    e = hipEventRecord(g->ev_pre, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(g->tail_stream, g->ev_pre, 0);
  }
  if (e != hipSuccess) rc = fail(MI_ERR_HIP, hipGetErrorString(e));
  if (rc == MI_OK)
    rc = phase2_batch(g, p.b, p.k, ws.L, p.out_idx, p.out_score, p.out_score64, g->tail_stream, /*have_cand=*/true,
                      /*resident=*/true, &ws);
  if (rc == MI_OK) {
    e = hipEventRecord(g->ev_tail[p.set], g->tail_stream);
    if (e != hipSuccess) rc = fail(MI_ERR_HIP, hipGetErrorString(e));
    else g->ev_tail_valid[p.set] = true;
  }
  ws.q_f32 = q_keep;
  ws.cand_rows = rows_keep;
  ws.cand_cnt = cnt_keep;
  ws.cand_score = sc_keep;
  return rc;
}

// a batch whose sticky flags were raised: a buffer overflow (or fp16 range) and a failed speculative threshold are counted apart
static void count_flagged_batch(mi_gallery* g, uint32_t flags) {
  if (flags & ~(uint32_t)FLAG_SPEC_FAIL) g->stats.overflow_batches += 1;
  else g->stats.spec_retries += 1;
}

static int check_k(const mi_gallery* g, int32_t k) {
  REQUIRE(k >= 1, "k must be >= 1");
  if ((int64_t)k > g->n)
    return fail(MI_ERR_INVALID, "k > number of gallery rows (the reference fails in numpy broadcasting here, "
                                "src/utils/nnsearch.py:703)");
  if ((uint32_t)k * 4 > g->surv_cap || (uint32_t)k > g->rescore_cap)
    return fail(MI_ERR_UNSUPPORTED, "k too large for the top-K path (needs k <= survivor_cap/4 and k <= rescore_cap)");
  return MI_OK;
}

// full search of up to any nq on device inputs (strided, any dtype), outputs on device
static int search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t q_rs, int64_t q_cs, int q_norm,
                         int64_t nq, int32_t k, int64_t* out_idx, float* out_score, double* out_score64, bool exact,
                         hipStream_t s, bool allow_async = false, bool caller_checks_flags = false) {
  int rc = check_k(g, k);
  if (rc != MI_OK) return rc;
  const bool async = g->async_tail != 0 && allow_async;
  if (g->pending.valid && !(async && g->async_tail == 3 && g->pending.k == k)) {
    // a deferred tail is waiting and this call cannot carry it (another mode, another k: the workspace may be rebuilt):
    // run it now and make this call's stream wait for it
    const int set = g->pending.set;
    if ((rc = flush_pending_tail(g, s, false)) != MI_OK) return rc;
    HIPC(hipStreamWaitEvent(s, g->ev_tail[set], 0));
  }
  if ((rc = ws_ensure(g, k)) != MI_OK) return rc;
  const size_t esz = q_dtype == MI_F32 ? 4 : 8;
  if (async && !g->tail_stream) {
    HIPC(hipStreamCreateWithFlags(&g->tail_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIPC(hipEventCreateWithFlags(&g->ev_p1[i], hipEventDisableTiming));
      HIPC(hipEventCreateWithFlags(&g->ev_tail[i], hipEventDisableTiming));
    }
    HIPC(hipEventCreateWithFlags(&g->ev_pre, hipEventDisableTiming));
  }
  Workspace& ws = g->ws;
  for (int64_t q0 = 0; q0 < nq; q0 += QB) {
    const int32_t b = (int32_t)std::min<int64_t>(QB, nq - q0);
    const char* src = (const char*)q_src + (size_t)q0 * q_rs * esz;
    hipStream_t tail = s;
    int set = 0;
    if (async) set = g->tail_set;
    if (async) {
      g->tail_set ^= 1;
      tail = g->tail_stream;
      if (g->ev_tail_valid[set]) HIPC(hipStreamWaitEvent(s, g->ev_tail[set], 0));   // the tail that last read this set is done
      // mode 2: the previous batch's tail (other set) may run beside this batch's query ingest and bootstrap -- both
      // latency-bound, the board far below its power cap -- but not beside its scoring launch (power-capped: nothing to gain)
      if (g->async_tail == 2 && g->ev_tail_valid[set ^ 1]) g->gate_before_scoring = g->ev_tail[set ^ 1];
    }
    ws.q_f32 = ws.q_f32_set[set];
    ws.cand_rows = ws.cand_rows_set[set];
    ws.cand_cnt = ws.cand_cnt_set[set];
    ws.cand_score = ws.cand_score_set[set];
    rc = phase1_batch(g, src, q_dtype, q_rs, q_cs, q_norm, b, k, exact, s, /*fuse_cand=*/true, caller_checks_flags);
    if (rc != MI_OK) return rc;
    if (async && g->async_tail == 3) {
      // deferred: a schedule without the single filtered launch (chunked, f32-scored) has not picked the previous tail up
      if (g->pending.valid && (rc = flush_pending_tail(g, s, false)) != MI_OK) return rc;
      HIPC(hipEventRecord(g->ev_p1[set], s));
      if (b > STREAM_MAX_QUERIES) {
        g->pending.valid = true;
        g->pending.b = b;
        g->pending.k = k;
        g->pending.set = set;
        g->pending.slot = g->ws_slot;
        g->pending.out_idx = out_idx + q0 * k;
        g->pending.out_score = out_score ? out_score + q0 * k : nullptr;
        g->pending.out_score64 = out_score64 ? out_score64 + q0 * k : nullptr;
        g->stats.searches += 1;
        g->stats.queries += b;
        continue;
      }
    }
    if (async) {
      HIPC(hipEventRecord(g->ev_p1[set], s));
      HIPC(hipStreamWaitEvent(tail, g->ev_p1[set], 0));
    }
    if ((rc = phase2_batch(g, b, k, g->ws.L, out_idx + q0 * k, out_score ? out_score + q0 * k : nullptr,
                           out_score64 ? out_score64 + q0 * k : nullptr, tail, /*have_cand=*/true,
                           /*resident=*/async && g->async_tail == 1 && b > STREAM_MAX_QUERIES)) != MI_OK)
      return rc;
    if (async) {
      HIPC(hipEventRecord(g->ev_tail[set], tail));
      g->ev_tail_valid[set] = true;
    }
    g->stats.searches += 1;
    g->stats.queries += b;
  }
  return MI_OK;
}

// makes `s` wait for every tail enqueued so far (no-op in the synchronous mode)
static int join_tails(mi_gallery* g, hipStream_t s) {
  if (g->pending.valid) {
    const int rc = flush_pending_tail(g, s, false);
    if (rc != MI_OK) return rc;
  }
  for (int i = 0; i < 2; ++i)
    if (g->ev_tail_valid[i]) HIPC(hipStreamWaitEvent(s, g->ev_tail[i], 0));
  return MI_OK;
}

static int strided_extent(int64_t n, int64_t d, int64_t rs, int64_t cs, int64_t* elems) {
  REQUIRE(rs >= 0 && cs >= 0, "negative strides are not supported");
  *elems = (n > 0 && d > 0) ? (n - 1) * rs + (d - 1) * cs + 1 : 0;
  return MI_OK;
}

// =====================================================================================================
extern "C" {

const char* mi_last_error(void) { return g_err.c_str(); }

int mi_device_count(int* count) {
  REQUIRE(count, "null");
  HIPC(hipGetDeviceCount(count));
  return MI_OK;
}

int mi_gallery_destroy(mi_gallery* g) {
  if (!g) return MI_OK;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->tail_stream) (void)hipStreamSynchronize(g->tail_stream);
  (void)hipDeviceSynchronize();
  (void)snapshot_balance(g, nullptr);
  for (int i = 0; i < 2; ++i) {
    if (g->ev_p1[i]) (void)hipEventDestroy(g->ev_p1[i]);
    if (g->ev_tail[i]) (void)hipEventDestroy(g->ev_tail[i]);
  }
  if (g->ev_pre) (void)hipEventDestroy(g->ev_pre);
  if (g->tail_stream) (void)hipStreamDestroy(g->tail_stream);
  ws_free(g->ws);
  ws_free(g->ws_alt);
  for (auto& e : g->ev_pool) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  {
    // (every stream of the handle is drained above: nothing in flight touches these buffers any more)
    std::lock_guard<std::mutex> lock(g_spare_mu);
    const size_t total = g->buf_bytes[0] + g->buf_bytes[1] + g->buf_bytes[2];
    if (g_keep_buffers.load() && g->gal_f32 && g->gal_img && g->rowstat && g->gstat3 && total <= SPARE_MAX_BYTES) {
      spare_release_locked();
      g_spare.device = g->device;
      g_spare.f32_bytes = g->buf_bytes[0], g_spare.img_bytes = g->buf_bytes[1], g_spare.stat_bytes = g->buf_bytes[2];
      g_spare.gal_f32 = g->gal_f32, g_spare.gal_img = g->gal_img, g_spare.rowstat = g->rowstat, g_spare.gstat3 = g->gstat3;
    } else {
      (void)hipFree(g->gal_f32);
      (void)hipFree(g->gal_img);
      (void)hipFree(g->rowstat);
      (void)hipFree(g->gstat3);
    }
  }
  (void)hipFree(g->samp_img);
  for (void* b : g->io_buf) (void)hipFree(b);
  (void)hipFree(g->dif_ids);
  (void)hipFree(g->dif_vals);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return MI_OK;
}

// Rows [row_base, row_base + m) of the gallery from a device source of any strides (m_pad >= m: trailing rows of the last tile
// to be written as zeros).  Layouts the ingest kernels take in one pass go straight in; anything else -- a [D, N] layout at a
// width other than 2048, fully strided sources -- is copied, 65 536 rows at a time, into a row-major scratch block first: the
// values are untouched, the sums are the row kernel's, so the gallery is the same bits whatever the layout.
static int ingest_rows_any_layout(mi_gallery* g, const void* src, int dtype, int64_t m, int64_t m_pad, int64_t rs, int64_t cs,
                                  int64_t row_base, hipStream_t s) {
  if (ingest_takes_layout(g->d, rs, cs)) {
    launch_ingest(src, dtype, m, g->d, rs, cs, g->norm_mode, g->gal_f32, g->gal_img, g->img_f16, g->rowstat, g->dp, m_pad, s,
                  row_base);
    HIPC(hipGetLastError());
    return MI_OK;
  }
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  const int64_t chunk = 65536;
  TmpAlloc tmp;
  char* scratch = tmp.get<char>((size_t)std::min<int64_t>(chunk, m) * g->d * esz);
  if (!scratch) return fail(MI_ERR_NOMEM, "ingest scratch block");
  for (int64_t r0 = 0; r0 < m_pad; r0 += chunk) {
    const int64_t rows = std::max<int64_t>(0, std::min<int64_t>(chunk, m - r0));
    const int64_t rows_pad = std::min<int64_t>(chunk, m_pad - r0);
    if (rows > 0) launch_transpose_rows((const char*)src + (size_t)r0 * rs * esz, dtype, rows, g->d, rs, cs, scratch, s);
    launch_ingest(scratch, dtype, rows, g->d, g->d, 1, g->norm_mode, g->gal_f32, g->gal_img, g->img_f16, g->rowstat, g->dp,
                  rows_pad, s, row_base + r0);
  }
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(s));           // the scratch block is freed on return
  return MI_OK;
}

static int gallery_alloc(mi_gallery* g) {
  g->dp = (int32_t)round_up(g->d, BK);
  g->npad = round_up(g->n, TILE);
  if (g->cap < g->n) g->cap = g->n;
  const int64_t cap_pad = round_up(g->cap, TILE);
  HIPC(hipSetDevice(g->device));
  HIPC(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  const size_t f32_bytes = (size_t)g->cap * g->dp * 4, bf_bytes = (size_t)cap_pad * g->dp * 2;
  const size_t stat_bytes = (size_t)cap_pad * sizeof(RowStat);
  g->buf_bytes[0] = f32_bytes, g->buf_bytes[1] = bf_bytes, g->buf_bytes[2] = stat_bytes;
  g->hbm_bytes = (int64_t)(f32_bytes + bf_bytes + stat_bytes);
  {
    std::lock_guard<std::mutex> lock(g_spare_mu);
    if (g_spare.device == g->device && g_spare.f32_bytes == f32_bytes && g_spare.img_bytes == bf_bytes &&
        g_spare.stat_bytes == stat_bytes) {                      // the buffers of the gallery destroyed last: same sizes
      g->gal_f32 = g_spare.gal_f32, g->gal_img = g_spare.gal_img, g->rowstat = g_spare.rowstat, g->gstat3 = g_spare.gstat3;
      g_spare = SpareBuffers();
      return MI_OK;
    }
    if (f32_bytes + bf_bytes + stat_bytes > SPARE_MAX_BYTES / 2) spare_release_locked();   // a big gallery of other sizes: make room
  }
  HIPC(hipMalloc((void**)&g->gal_f32, f32_bytes + 256));
  HIPC(hipMalloc(&g->gal_img, bf_bytes + 256));
  HIPC(hipMalloc((void**)&g->rowstat, stat_bytes));
  HIPC(hipMalloc((void**)&g->gstat3, 16));
  return MI_OK;
}

// Host array -> gallery, in row blocks.  The reference's layouts: rows contiguous (cs == 1: a block is m whole rows) or the [D, N]
// layout (rs == 1: a block is d runs of m consecutive rows, packed [d][m] on the device and read with strides (1, m)).  Block i + 1
// crosses PCIe while block i is ingested; nothing the size of the gallery is allocated besides the gallery.
static int gallery_ingest_host_blocks(mi_gallery* g, const void* data, int dtype, int64_t n, int64_t rs, int64_t cs) {
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  const int32_t d = g->d;
  const bool by_cols = rs == 1 && cs != 1;                     // [D, N] layout
  const int64_t m_blk = std::max<int64_t>(TILE, ((int64_t)32 << 20) / ((int64_t)d * (int64_t)esz) / TILE * TILE);   // ~32 MiB, whole tiles
  const size_t blk_bytes = (size_t)m_blk * d * esz;
  TmpAlloc tmp;
  char* dev[2] = {tmp.get<char>(blk_bytes), tmp.get<char>(blk_bytes)};
  if (!dev[0] || !dev[1]) return fail(MI_ERR_NOMEM, "host ingest device blocks");
  hipStream_t cs_stream = nullptr;
  hipEvent_t copied[2] = {nullptr, nullptr}, ingested[2] = {nullptr, nullptr};
  auto release = [&](int code) {
    if (cs_stream) { (void)hipStreamSynchronize(cs_stream); (void)hipStreamDestroy(cs_stream); }
    (void)hipStreamSynchronize(g->stream);
    for (int i = 0; i < 2; ++i) {
      if (copied[i]) (void)hipEventDestroy(copied[i]);
      if (ingested[i]) (void)hipEventDestroy(ingested[i]);
    }
    return code;
  };
#define HIPR(expr)                                                                                           \
  do {                                                                                                       \
    hipError_t _e = (expr);                                                                                  \
    if (_e != hipSuccess) return release(fail(MI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  HIPR(hipStreamCreateWithFlags(&cs_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    HIPR(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
    HIPR(hipEventCreateWithFlags(&ingested[i], hipEventDisableTiming));
  }
  const char* base = (const char*)data;
  int64_t nblk = 0;
  for (int64_t r0 = 0; r0 < n; r0 += m_blk, ++nblk) {
    const int slot = (int)(nblk & 1);
    const int64_t m = std::min<int64_t>(m_blk, n - r0);
    const bool last = r0 + m >= n;
    if (nblk >= 2) HIPR(hipStreamWaitEvent(cs_stream, ingested[slot], 0));     // the device block is free again
    // the runtime copies from the caller's pageable pages (it stages / pins them itself); these calls return when the bytes
    // have left the host, so block i + 1 is still copied while block i is ingested on the handle's stream
    if (by_cols) HIPR(hipMemcpy2DAsync(dev[slot], (size_t)m * esz, base + (size_t)r0 * esz, (size_t)cs * esz, (size_t)m * esz,
                                       (size_t)d, hipMemcpyHostToDevice, cs_stream));
    else HIPR(hipMemcpy2DAsync(dev[slot], (size_t)d * esz, base + (size_t)r0 * rs * esz, (size_t)rs * esz, (size_t)d * esz,
                               (size_t)m, hipMemcpyHostToDevice, cs_stream));
    HIPR(hipEventRecord(copied[slot], cs_stream));
    HIPR(hipStreamWaitEvent(g->stream, copied[slot], 0));
    const int64_t m_pad = last ? g->npad - r0 : m;
    const int rc = by_cols ? ingest_rows_any_layout(g, dev[slot], dtype, m, m_pad, 1, m, r0, g->stream)
                           : ingest_rows_any_layout(g, dev[slot], dtype, m, m_pad, d, 1, r0, g->stream);
    if (rc != MI_OK) return release(rc);
    HIPR(hipEventRecord(ingested[slot], g->stream));
  }
#undef HIPR
  return release(MI_OK);
}

int mi_gallery_create(const void* data, int64_t n, int32_t d, int dtype, int64_t row_stride, int64_t col_stride,
                      int memspace, int norm_mode, int device, int64_t row_offset, mi_gallery** out) {
  REQUIRE(data && out, "null pointer");
  REQUIRE(n >= 1 && d >= 1, "empty gallery");
  REQUIRE(n < (int64_t)1 << 32, "a shard holds at most 2^32-1 rows");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  REQUIRE(norm_mode >= 0 && norm_mode <= 2, "bad norm_mode");
  int64_t elems;
  int rc = strided_extent(n, d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  mi_gallery* g = new mi_gallery();
  g->device = device;
  g->n = n;
  g->d = d;
  g->norm_mode = norm_mode;
  g->row_offset = row_offset;
  if ((rc = gallery_alloc(g)) != MI_OK) {
    mi_gallery_destroy(g);
    return rc;
  }
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  void* staged = nullptr;
  const void* src = data;
  auto cleanup = [&](int code) {
    if (staged) (void)hipFree(staged);
    mi_gallery_destroy(g);
    return code;
  };
  // host arrays in one of the reference's two layouts: block pipeline (no staging the size of the gallery)
  const int host_mode = g_host_ingest.load();
  const bool host_blocks = memspace == MI_HOST && host_mode != 0 && (col_stride == 1 || row_stride == 1) && d > 1 &&
                           (col_stride == 1 ? row_stride >= d : col_stride >= n);
  if (memspace == MI_HOST && !host_blocks) {
    hipError_t e = hipMalloc(&staged, (size_t)elems * esz + 256);
    if (e != hipSuccess) return cleanup(fail(MI_ERR_NOMEM, std::string("staging hipMalloc: ") + hipGetErrorString(e)));
    e = hipMemcpy(staged, data, (size_t)elems * esz, hipMemcpyHostToDevice);
    if (e != hipSuccess) return cleanup(fail(MI_ERR_HIP, std::string("H2D copy: ") + hipGetErrorString(e)));
    src = staged;
  }
  g->img_f16 = g_default_img_f16.load();
  hipError_t e = hipSuccess;
  for (int pass = 0; pass < 2; ++pass) {
    rc = host_blocks ? gallery_ingest_host_blocks(g, data, dtype, n, row_stride, col_stride)
                     : ingest_rows_any_layout(g, src, dtype, n, g->npad, row_stride, col_stride, 0, g->stream);
    if (rc != MI_OK) return cleanup(rc);
    launch_rowstat_max(g->rowstat, n, g->gstat3, g->stream);
    e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess || !g->img_f16 || norm_mode != MI_NORM_NONE) break;
    // raw (un-normalised) rows: fp16 only if they sit comfortably inside its range, otherwise re-ingest as bf16
    float gs[3] = {0, 0, 0};
    e = hipMemcpy(gs, g->gstat3, 12, hipMemcpyDeviceToHost);
    if (e != hipSuccess || (gs[0] <= 4.0f && std::isfinite(gs[1]))) break;
    g->img_f16 = 0;
  }
  if (e != hipSuccess) return cleanup(fail(MI_ERR_HIP, std::string("ingest: ") + hipGetErrorString(e)));
  if (staged) (void)hipFree(staged);
  *out = g;
  return MI_OK;
}

int mi_gallery_create_empty(int64_t capacity, int32_t d, int norm_mode, int device, int64_t row_offset,
                            mi_gallery** out) {
  REQUIRE(out, "null pointer");
  REQUIRE(capacity >= 1 && capacity < ((int64_t)1 << 32) && d >= 1, "bad sizes");
  REQUIRE(norm_mode >= 0 && norm_mode <= 2, "bad norm_mode");
  mi_gallery* g = new mi_gallery();
  g->device = device;
  g->n = 0;
  g->cap = capacity;
  g->d = d;
  g->norm_mode = norm_mode;
  g->row_offset = row_offset;
  g->img_f16 = (norm_mode == MI_NORM_NONE) ? 0 : g_default_img_f16.load();   // raw rows of unknown range: bf16 image
  int rc = gallery_alloc(g);
  if (rc == MI_OK) {
    hipError_t e = hipMemset(g->gal_img, 0, (size_t)round_up(capacity, TILE) * g->dp * 2);
    if (e == hipSuccess) e = hipMemset(g->gstat3, 0, 12);
    if (e != hipSuccess) rc = fail(MI_ERR_HIP, std::string("memset: ") + hipGetErrorString(e));
  }
  if (rc != MI_OK) {
    mi_gallery_destroy(g);
    return rc;
  }
  *out = g;
  return MI_OK;
}

int mi_gallery_append_device(mi_gallery* g, const float* rows_dev, int64_t m, void* stream) {
  REQUIRE(g && rows_dev, "null pointer");
  REQUIRE(m >= 1, "nothing to append");
  REQUIRE(g->n + m <= g->cap, "gallery capacity exceeded");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  hipStream_t s = (hipStream_t)stream;
  // rows [n, n+m): normalise like the gallery, write f32 rows + 16-bit image + rounding norms at their final place
  launch_ingest(rows_dev, MI_F32, m, g->d, g->d, 1, g->norm_mode, g->gal_f32, g->gal_img, g->img_f16, g->rowstat, g->dp, m,
                s, g->n);
  launch_rowstat_max(g->rowstat + g->n, m, g->gstat3, s, /*reset=*/false);
  HIPC(hipGetLastError());
  g->n += m;
  g->npad = round_up(g->n, TILE);
  return MI_OK;
}

int mi_gallery_append(mi_gallery* g, const void* data, int64_t m, int dtype, int64_t row_stride, int64_t col_stride,
                      int memspace) {
  REQUIRE(g && data, "null pointer");
  REQUIRE(m >= 1, "nothing to append");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  REQUIRE(g->n + m <= g->cap, "gallery capacity exceeded");
  int64_t elems;
  int rc = strided_extent(m, g->d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  hipStream_t s = g->stream;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  TmpAlloc tmp;
  const void* src = data;
  int64_t rs = row_stride, cs = col_stride;
  if (memspace == MI_HOST) {
    // Only the bytes of these m rows cross PCIe.  A column block of a [D, N] array (the reference's layout: callers pass
    // vecs.T, src/test_rOP1m.py:156) is d runs of m contiguous elements: a 2-D copy packs it to [d][m] on the device and
    // the ingest kernel reads it with strides (1, m) -- no host transpose, no float64 promotion.
    if (row_stride == 1 && col_stride >= m) {
      char* st = tmp.get<char>((size_t)g->d * m * esz);
      if (!st) return fail(MI_ERR_NOMEM, "append staging");
      HIPC(hipMemcpy2D(st, (size_t)m * esz, data, (size_t)col_stride * esz, (size_t)m * esz, (size_t)g->d,
                       hipMemcpyHostToDevice));
      src = st;
      rs = 1;
      cs = m;
    } else {
      char* st = tmp.get<char>((size_t)elems * esz);
      if (!st) return fail(MI_ERR_NOMEM, "append staging");
      HIPC(hipMemcpy(st, data, (size_t)elems * esz, hipMemcpyHostToDevice));
      src = st;
    }
  }
  if ((rc = ingest_rows_any_layout(g, src, dtype, m, m, rs, cs, g->n, s)) != MI_OK) return rc;
  launch_rowstat_max(g->rowstat + g->n, m, g->gstat3, s, /*reset=*/false);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(s));          // the staging buffer is freed on return
  g->n += m;
  g->npad = round_up(g->n, TILE);
  return MI_OK;
}

int mi_desc_tail_device(const float* feat_dev, int32_t b, int32_t c, int32_t hw, float p, float eps,
                        const float* whiten_w_dev, const float* whiten_b_dev, int32_t c_out, float* scratch_dev,
                        float* out_dev, void* stream) {
  REQUIRE(feat_dev && out_dev, "null pointer");
  REQUIRE(b >= 1 && c >= 1 && hw >= 1, "bad sizes");
  REQUIRE(!whiten_w_dev || (scratch_dev && c_out >= 1), "whitening needs a [b][c] scratch buffer and c_out");
  REQUIRE(!whiten_w_dev || (size_t)8 * c * 4 <= 160 * 1024 - 1024, "c too large for the whitening kernel");
  launch_desc_tail(feat_dev, b, c, hw, p, eps, whiten_w_dev, whiten_b_dev, c_out, scratch_dev, out_dev,
                   (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_desc_ms_accumulate_device(float* acc_dev, const float* desc_dev, int64_t count, float msp, int first,
                                 void* stream) {
  REQUIRE(acc_dev && desc_dev && count >= 1, "bad arguments");
  launch_ms_accumulate(acc_dev, desc_dev, count, msp, first, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_desc_ms_finish_device(float* acc_dev, int32_t b, int32_t d, int32_t nscales, float msp, void* stream) {
  REQUIRE(acc_dev && b >= 1 && d >= 1 && nscales >= 1, "bad arguments");
  launch_ms_finish(acc_dev, b, d, nscales, msp, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_gallery_info(const mi_gallery* g, int64_t* n, int32_t* d, int32_t* norm_mode, int32_t* device,
                    int64_t* row_offset, int64_t* hbm_bytes) {
  REQUIRE(g, "null handle");
  if (n) *n = g->n;
  if (d) *d = g->d;
  if (norm_mode) *norm_mode = g->norm_mode;
  if (device) *device = g->device;
  if (row_offset) *row_offset = g->row_offset;
  if (hbm_bytes) *hbm_bytes = g->hbm_bytes;
  return MI_OK;
}

int mi_gallery_get_rows(const mi_gallery* g, int64_t row0, int64_t nrows, float* out_host) {
  REQUIRE(g && out_host, "null");
  REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= g->n, "row range out of bounds");
  HIPC(hipSetDevice(g->device));
  HIPC(hipMemcpy2D(out_host, (size_t)g->d * 4, g->gal_f32 + row0 * g->dp, (size_t)g->dp * 4, (size_t)g->d * 4,
                   (size_t)nrows, hipMemcpyDeviceToHost));
  return MI_OK;
}

// ---- persistence -----------------------------------------------------------------------------------
// Prepared-gallery file "MI355GAL" v2: header | f32 rows [n][dp] | 16-bit image [npad][dp] | row
// norms [npad].  The header carries a checksum per section and one of itself; sections move through two pinned host
// buffers so that the disk and the PCIe copy overlap (the v1 loader went through one pageable 64 MB buffer), and the
// section checksums are recomputed on the device after the copy.
namespace {
struct FileHeader {
  char magic[8];
  int64_t version, n, npad, row_offset;
  int32_t d, dp, norm_mode, img_f16;
  float gstat3[3];
  uint32_t reserved;
  uint64_t section_sum[3];     // f32 rows, image, row norms
  uint64_t header_sum;         // of all bytes above
};
uint64_t host_sum(const void* p, size_t bytes) {       // FNV-1a, header only
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < bytes; ++i) h = (h ^ ((const unsigned char*)p)[i]) * 0x100000001b3ull;
  return h;
}
struct BalanceTrailer {        // optional, after the last section
  char magic[8];               // "MIXCCBAL"
  float w[8];
  uint64_t sum;                // FNV-1a of the bytes above
};
struct FileSegment {
  size_t file_off;
  char* dev;
  size_t bytes;
};
// device -> file: 32 MiB chunks cross PCIe into a ring of eight pinned buffers; four writer threads pwrite() each chunk at its
// place as soon as its copy is done.  (Measured: 12 GB reach the page cache at 11 GB/s with one writer and with four -- the
// kernel's dirty-page throttling, not the copy; callers write the file behind their call, nnsearch._save_behind.)
int copy_dev_to_file_parallel(int fd, int device, const FileSegment* seg, int nseg) {
  constexpr size_t CH = (size_t)32 << 20;
  constexpr int WRITERS = 4, RING = 8;
  struct Chunk { size_t off; char* dev; size_t len; };
  std::vector<Chunk> ch;
  for (int i = 0; i < nseg; ++i)
    for (size_t o = 0; o < seg[i].bytes; o += CH) ch.push_back({seg[i].file_off + o, seg[i].dev + o, std::min(CH, seg[i].bytes - o)});
  const size_t n = ch.size();
  if (n == 0) return MI_OK;
  void* buf[RING] = {};
  hipEvent_t ev[RING] = {};
  hipStream_t s = nullptr;
  hipError_t he = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int ring = (int)std::min<size_t>(RING, n);
  for (int i = 0; i < ring && he == hipSuccess; ++i) {
    he = hipHostMalloc(&buf[i], CH, hipHostMallocDefault);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  std::vector<std::atomic<int>> issued(n), written(n);    // the copy of chunk i is enqueued / chunk i is in the file
  for (size_t i = 0; i < n; ++i) issued[i] = 0, written[i] = 0;
  std::atomic<size_t> next{0};
  std::atomic<int> io_err{0}, hip_err{0}, stop{0};
  std::vector<std::thread> writers;
  if (he == hipSuccess)
    for (int t = 0; t < std::min<int>(WRITERS, ring); ++t)
      writers.emplace_back([&] {
        (void)hipSetDevice(device);
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n) return;
          while (!issued[i].load(std::memory_order_acquire)) {
            if (stop.load()) return;
            std::this_thread::yield();
          }
          if (hipEventSynchronize(ev[i % ring]) != hipSuccess) hip_err = 1;
          size_t put = 0;
          while (put < ch[i].len && !hip_err.load() && !io_err.load()) {
            const ssize_t r = pwrite(fd, (const char*)buf[i % ring] + put, ch[i].len - put, (off_t)(ch[i].off + put));
            if (r <= 0) {
              io_err = 1;
              break;
            }
            put += (size_t)r;
          }
          written[i].store(1, std::memory_order_release);
        }
      });
  for (size_t i = 0; i < n && he == hipSuccess && !io_err.load() && !hip_err.load(); ++i) {
    if (i >= (size_t)ring)
      while (!written[i - ring].load(std::memory_order_acquire)) std::this_thread::yield();   // its buffer is free again
    he = hipMemcpyAsync(buf[i % ring], ch[i].dev, ch[i].len, hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipEventRecord(ev[i % ring], s);
    if (he == hipSuccess) issued[i].store(1, std::memory_order_release);
  }
  if (he != hipSuccess || io_err.load() || hip_err.load()) stop = 1;
  for (auto& t : writers) t.join();
  if (s) (void)hipStreamSynchronize(s);
  for (int i = 0; i < ring; ++i) {
    if (buf[i]) (void)hipHostFree(buf[i]);
    if (ev[i]) (void)hipEventDestroy(ev[i]);
  }
  if (s) (void)hipStreamDestroy(s);
  if (he != hipSuccess || hip_err.load()) return fail(MI_ERR_HIP, std::string("device -> gallery file: ") + hipGetErrorString(he));
  if (io_err.load()) return fail(MI_ERR_IO, "short write");
  return MI_OK;
}
// file -> device: four reader threads pread() 32 MiB chunks into a ring of eight pinned buffers, every chunk crosses PCIe as soon
// as it is read.  A cached 12 GB file reaches the device at the pinned H2D rate of the box this way (55 GB/s; one reader and
// two buffers: 17 GB/s, one kernel memcpy stream; a read-only mapping copied by the runtime as pageable memory: 30-42 GB/s incl.
// the unmapping -- scripts/mapload_probe.hip, profiles/r05r_mapload_probe.txt).  Nothing the size of the file is resident.
int copy_file_to_dev_parallel(int fd, const FileSegment* seg, int nseg) {
  constexpr size_t CH = (size_t)32 << 20;
  constexpr int READERS = 4, RING = 8;
  struct Chunk { size_t off; char* dev; size_t len; };
  std::vector<Chunk> ch;
  for (int i = 0; i < nseg; ++i)
    for (size_t o = 0; o < seg[i].bytes; o += CH) ch.push_back({seg[i].file_off + o, seg[i].dev + o, std::min(CH, seg[i].bytes - o)});
  const size_t n = ch.size();
  if (n == 0) return MI_OK;
  void* buf[RING] = {};
  hipEvent_t ev[RING] = {};
  hipStream_t s = nullptr;
  hipError_t he = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int ring = (int)std::min<size_t>(RING, n);
  for (int i = 0; i < ring && he == hipSuccess; ++i) {
    he = hipHostMalloc(&buf[i], CH, hipHostMallocDefault);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  std::vector<std::atomic<int>> filled(n), freed(n);      // chunk i is in its buffer / its copy has left the buffer
  for (size_t i = 0; i < n; ++i) filled[i] = 0, freed[i] = 0;
  std::atomic<size_t> next{0};
  std::atomic<int> io_err{0}, stop{0};
  std::vector<std::thread> readers;
  if (he == hipSuccess)
    for (int t = 0; t < std::min<int>(READERS, ring); ++t)
      readers.emplace_back([&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n) return;
          if (i >= (size_t)ring)
            while (!freed[i - ring].load(std::memory_order_acquire)) {
              if (stop.load()) return;
              std::this_thread::yield();
            }
          size_t got = 0;
          while (got < ch[i].len && !stop.load()) {
            const ssize_t r = pread(fd, (char*)buf[i % ring] + got, ch[i].len - got, (off_t)(ch[i].off + got));
            if (r <= 0) {
              io_err = 1;
              break;
            }
            got += (size_t)r;
          }
          filled[i].store(1, std::memory_order_release);
        }
      });
  size_t released = 0;
  for (size_t i = 0; i < n && he == hipSuccess && !io_err.load(); ++i) {
    while (!filled[i].load(std::memory_order_acquire)) std::this_thread::yield();
    if (io_err.load()) break;
    he = hipMemcpyAsync(ch[i].dev, buf[i % ring], ch[i].len, hipMemcpyHostToDevice, s);
    if (he == hipSuccess) he = hipEventRecord(ev[i % ring], s);
    // hand buffers back in order once their copies are done; keep half a ring of copies in flight
    while (he == hipSuccess && released <= i && (i - released >= (size_t)ring / 2 || i + 1 == n)) {
      he = hipEventSynchronize(ev[released % ring]);
      freed[released].store(1, std::memory_order_release);
      ++released;
    }
  }
  stop = (he != hipSuccess || io_err.load()) ? 1 : 0;
  for (auto& t : readers) t.join();
  if (s) (void)hipStreamSynchronize(s);
  for (int i = 0; i < ring; ++i) {
    if (buf[i]) (void)hipHostFree(buf[i]);
    if (ev[i]) (void)hipEventDestroy(ev[i]);
  }
  if (s) (void)hipStreamDestroy(s);
  if (he != hipSuccess) return fail(MI_ERR_HIP, std::string("gallery file -> device: ") + hipGetErrorString(he));
  if (io_err.load()) return fail(MI_ERR_IO, "short read (truncated gallery file)");
  return MI_OK;
}
int section_sums(const mi_gallery* g, uint64_t out[3]) {
  unsigned long long* d = nullptr;
  HIPC(hipMalloc((void**)&d, 24));
  launch_checksum(g->gal_f32, (size_t)g->n * g->dp * 4, d + 0, g->stream);
  launch_checksum(g->gal_img, (size_t)g->npad * g->dp * 2, d + 1, g->stream);
  launch_checksum(g->rowstat, (size_t)g->npad * sizeof(RowStat), d + 2, g->stream);
  hipError_t e = hipStreamSynchronize(g->stream);
  if (e == hipSuccess) e = hipMemcpy(out, d, 24, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MI_ERR_HIP, std::string("checksum: ") + hipGetErrorString(e));
  return MI_OK;
}
}  // namespace

int mi_gallery_save(const mi_gallery* g, const char* path) {
  REQUIRE(g && path, "null");
  HIPC(hipSetDevice(g->device));
  HIPC(hipStreamSynchronize(g->stream));
  FileHeader h{};
  memcpy(h.magic, "MI355GAL", 8);
  h.version = 2;
  h.n = g->n;
  h.npad = g->npad;
  h.row_offset = g->row_offset;
  h.d = g->d;
  h.dp = g->dp;
  h.norm_mode = g->norm_mode;
  h.img_f16 = g->img_f16;
  HIPC(hipMemcpy(h.gstat3, g->gstat3, 12, hipMemcpyDeviceToHost));
  int rc = section_sums(g, h.section_sum);
  if (rc != MI_OK) return rc;
  h.header_sum = host_sum(&h, offsetof(FileHeader, header_sum));
  FILE* f = fopen(path, "wb");
  if (!f) return fail(MI_ERR_IO, std::string("cannot open for writing: ") + path);
  if (fwrite(&h, sizeof h, 1, f) != 1 || fflush(f) != 0) rc = fail(MI_ERR_IO, "short write");
  const size_t sec[3] = {(size_t)g->n * g->dp * 4, (size_t)g->npad * g->dp * 2, (size_t)g->npad * sizeof(RowStat)};
  char* src[3] = {(char*)g->gal_f32, (char*)g->gal_img, (char*)g->rowstat};
  FileSegment seg[3];
  size_t off = sizeof h;
  for (int i = 0; i < 3; ++i) {
    seg[i] = {off, src[i], sec[i]};
    off += sec[i];
  }
  if (rc == MI_OK) rc = copy_dev_to_file_parallel(fileno(f), g->device, seg, 3);
  if (rc == MI_OK && fseeko(f, (off_t)off, SEEK_SET) != 0) rc = fail(MI_ERR_IO, "seek failed");
  if (rc == MI_OK) {
    // optional trailer: the XCD shares of the tile kernel as measured so far (or as loaded), so that `load -> first search`
    // starts calibrated; files without it (no large launch has run yet) are complete
    BalanceTrailer tr{};
    memcpy(tr.magic, "MIXCCBAL", 8);
    bool have = snapshot_balance(g, tr.w);
    if (!have && g->file_w_valid) { memcpy(tr.w, g->file_w, sizeof tr.w); have = true; }
    tr.sum = host_sum(&tr, offsetof(BalanceTrailer, sum));
    if (have && fwrite(&tr, sizeof tr, 1, f) != 1) rc = fail(MI_ERR_IO, "short write");
  }
  if (fclose(f) != 0 && rc == MI_OK) rc = fail(MI_ERR_IO, "close failed");
  return rc;
}

int mi_gallery_load(const char* path, int device, mi_gallery** out) {
  REQUIRE(path && out, "null");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(MI_ERR_IO, std::string("cannot open: ") + path);
  FileHeader h{};
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "MI355GAL", 8) != 0 || h.version != 2) {
    fclose(f);
    return fail(MI_ERR_IO, "not a MI355GAL v2 file (files of the v1 layout carry no checksums: rebuild with ifgenerate)");
  }
  if (h.header_sum != host_sum(&h, offsetof(FileHeader, header_sum))) {
    fclose(f);
    return fail(MI_ERR_IO, "gallery file header checksum mismatch");
  }
  if (h.n < 1 || h.d < 1 || h.n >= ((int64_t)1 << 32) || h.norm_mode < 0 || h.norm_mode > 2) {
    fclose(f);
    return fail(MI_ERR_IO, "inconsistent header");
  }
  mi_gallery* g = new mi_gallery();
  g->device = device;
  g->n = h.n;
  g->d = h.d;
  g->norm_mode = h.norm_mode;
  g->img_f16 = h.img_f16;
  g->row_offset = h.row_offset;
  int rc = gallery_alloc(g);
  if (rc == MI_OK && (g->dp != h.dp || g->npad != h.npad)) rc = fail(MI_ERR_IO, "inconsistent header");
  if (rc == MI_OK) {
    const size_t sec[3] = {(size_t)g->n * g->dp * 4, (size_t)g->npad * g->dp * 2, (size_t)g->npad * sizeof(RowStat)};
    void* dst[3] = {g->gal_f32, g->gal_img, g->rowstat};
    FileSegment seg[3];
    size_t off = sizeof(FileHeader);
    for (int i = 0; i < 3; ++i) {
      seg[i] = {off, (char*)dst[i], sec[i]};
      off += sec[i];
    }
    rc = copy_file_to_dev_parallel(fileno(f), seg, 3);
    if (rc == MI_OK && fseeko(f, (off_t)off, SEEK_SET) != 0) rc = fail(MI_ERR_IO, "seek failed");
  }
  if (rc == MI_OK) {
    BalanceTrailer tr{};
    const size_t got = fread(&tr, 1, sizeof tr, f);
    if (got == sizeof tr && memcmp(tr.magic, "MIXCCBAL", 8) == 0 && tr.sum == host_sum(&tr, offsetof(BalanceTrailer, sum)) &&
        fgetc(f) == EOF) {
      memcpy(g->file_w, tr.w, sizeof tr.w);
      g->file_w_valid = true;
    } else if (got != 0) {
      rc = fail(MI_ERR_IO, "trailing bytes after the last section");
    }
  }
  fclose(f);
  if (rc == MI_OK && hipMemcpy(g->gstat3, h.gstat3, 12, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(MI_ERR_HIP, "gstat3 copy failed");
  if (rc == MI_OK) {
    uint64_t sums[3];
    rc = section_sums(g, sums);
    static const char* names[3] = {"f32 rows", "16-bit image", "row norms"};
    for (int i = 0; i < 3 && rc == MI_OK; ++i)
      if (sums[i] != h.section_sum[i])
        rc = fail(MI_ERR_IO, std::string("gallery file checksum mismatch in section: ") + names[i]);
  }
  if (rc != MI_OK) {
    mi_gallery_destroy(g);
    return rc;
  }
  *out = g;
  return MI_OK;
}

// ---- search ----------------------------------------------------------------------------------------
static int read_and_clear_flags(mi_gallery* g, uint32_t* flags) {
  HIPC(hipMemcpy(flags, g->ws.flags, 4, hipMemcpyDeviceToHost));
  if (*flags) HIPC(hipMemset(g->ws.flags, 0, 4));
  return MI_OK;
}

static int dense64_search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t rs, int64_t cs, int q_norm,
                                 int64_t nq, int32_t k, int64_t* out_idx_dev, float* out_score_dev, double* out_score64_dev,
                                 hipStream_t s);

// synchronous search with overflow handling: 16-bit MFMA pass, then (if buffers overflowed) the f32-scored filter pass,
// then (massive ties: more rows within the margin of the K-th score than the buffers hold) the dense f64 path: every
// score at full precision, exact top-k of the dense matrix -- the same contract as the filtered paths, at any data
static int search_sync(mi_gallery* g, const void* q_dev, int q_dtype, int64_t rs, int64_t cs, int q_norm, int64_t nq,
                       int32_t k, int64_t* idx_dev, float* score_dev, double* score64_dev) {
  hipStream_t s = g->stream;
  int rc = check_k(g, k);
  if (rc != MI_OK) return rc;
  if ((rc = ws_ensure(g, k)) != MI_OK) return rc;
  const size_t esz = q_dtype == MI_F32 ? 4 : 8;
  if (nq > QB && g->stream_tail && g->async_tail == 0 && !g->force_exact) {
    // More than one batch: the N x N callers of the path (AQE / DBA / k-reciprocal re-ranking, src/utils/Reranking.py:314-432,
    // the diffusion re-search) and query sets beyond 1024.  Their batches run as a stream with the DEFERRED tail (async_tail 3):
    // the exact re-score + final order of batch i is enqueued right before the scoring launch of batch i + 1 and runs beside
    // it on the handle's own stream (+2 ... 7 % queries/s); a failed threshold is repaired on the device (no host
    // round trip between the batches) and the sticky flags are read ONCE, after the last batch.  A raised flag -- an
    // overflow somewhere in the stream -- sends the whole call through the verified per-batch loop below.
    g->async_tail = 3;
    rc = search_device(g, q_dev, q_dtype, rs, cs, q_norm, nq, k, idx_dev, score_dev, score64_dev, false, s,
                       /*allow_async=*/true, /*caller_checks_flags=*/false);
    const int rc2 = join_tails(g, s);
    g->async_tail = 0;
    if (rc != MI_OK) return rc;
    if (rc2 != MI_OK) return rc2;
    HIPC(hipStreamSynchronize(s));
    uint32_t flags = 0;
    if ((rc = read_and_clear_flags(g, &flags)) != MI_OK) return rc;
    if (!flags) return MI_OK;
    count_flagged_batch(g, flags);
  }
  for (int64_t q0 = 0; q0 < nq; q0 += QB) {
    const int64_t b = std::min<int64_t>(QB, nq - q0);
    const char* src = (const char*)q_dev + (size_t)q0 * rs * esz;
    bool exact = g->force_exact != 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
      if (attempt == 2) {
        if ((rc = dense64_search_device(g, src, q_dtype, rs, cs, q_norm, b, k, idx_dev + q0 * k,
                                        score_dev ? score_dev + q0 * k : nullptr,
                                        score64_dev ? score64_dev + q0 * k : nullptr, s)) != MI_OK)
          return rc;
        break;
      }
      if ((rc = search_device(g, src, q_dtype, rs, cs, q_norm, b, k, idx_dev + q0 * k,
                              score_dev ? score_dev + q0 * k : nullptr,
                              score64_dev ? score64_dev + q0 * k : nullptr, exact, s, /*allow_async=*/false,
                              /*caller_checks_flags=*/true)) != MI_OK)
        return rc;
      HIPC(hipStreamSynchronize(s));
      uint32_t flags = 0;
      if ((rc = read_and_clear_flags(g, &flags)) != MI_OK) return rc;
      if (!flags) break;
      count_flagged_batch(g, flags);
      if (exact && g->exact_fallback && k <= 4096) continue;      // -> dense f64 path
      if (exact || !g->exact_fallback)
        return fail(MI_ERR_OVERFLOW,
                    "candidate buffers overflowed (more than survivor_cap / rescore_cap rows within the error margin "
                    "of the K-th score); raise the caps with mi_set_option");
      exact = true;
    }
  }
  return MI_OK;
}

int mi_knn_search(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                  int32_t k, int64_t* out_idx, float* out_score, double* out_seconds) {
  REQUIRE(g && q && out_idx, "null pointer");
  REQUIRE(nq >= 1, "no queries");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  const auto t0 = std::chrono::steady_clock::now();
  int64_t elems;
  int rc = strided_extent(nq, g->d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  auto stage = [&](int slot, size_t bytes) -> void* {
    if (g->io_cap[slot] < bytes) {
      (void)hipFree(g->io_buf[slot]);
      g->io_buf[slot] = nullptr;
      g->io_cap[slot] = 0;
      const size_t want = bytes + bytes / 4 + 256;
      if (hipMalloc(&g->io_buf[slot], want) != hipSuccess) return nullptr;
      g->io_cap[slot] = want;
    }
    return g->io_buf[slot];
  };
  if ((rc = check_k(g, k)) != MI_OK) return rc;            // before sizing buffers by k
  void* qd = stage(0, (size_t)elems * esz);
  int64_t* idx_d = (int64_t*)stage(1, (size_t)nq * k * 8);
  float* sc_d = (float*)stage(2, (size_t)nq * k * 4);
  auto done = [&](int code) { return code; };
  if (!qd || !idx_d || !sc_d) return fail(MI_ERR_NOMEM, "staging buffers of mi_knn_search");
  if (hipMemcpy(qd, q, (size_t)elems * esz, hipMemcpyHostToDevice) != hipSuccess)
    return done(fail(MI_ERR_HIP, "H2D query copy failed"));
  rc = search_sync(g, qd, dtype, row_stride, col_stride, g->norm_mode, nq, k, idx_d, sc_d, nullptr);
  if (rc != MI_OK) return done(rc);
  if (hipMemcpy(out_idx, idx_d, (size_t)nq * k * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return done(fail(MI_ERR_HIP, "D2H idx copy failed"));
  if (out_score && hipMemcpy(out_score, sc_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return done(fail(MI_ERR_HIP, "D2H score copy failed"));
  if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return done(MI_OK);
}

int mi_knn_search_device(mi_gallery* g, const float* q_dev, int64_t nq, int32_t k, int64_t* out_idx_dev,
                         float* out_score_dev, double* out_score64_dev, void* stream) {
  REQUIRE(g && q_dev && out_idx_dev, "null pointer");
  REQUIRE(nq >= 1, "no queries");
  HIPC(hipSetDevice(g->device));
  return search_device(g, q_dev, MI_F32, g->d, 1, g->qnorm_override >= 0 ? g->qnorm_override : g->norm_mode, nq, k,
                       out_idx_dev, out_score_dev, out_score64_dev, g->force_exact != 0, (hipStream_t)stream,
                       /*allow_async=*/true, /*caller_checks_flags=*/false);
}

int mi_gallery_calibrate(mi_gallery* g, int32_t launches, void* stream) {
  REQUIRE(g, "null handle");
  REQUIRE(launches >= 0 && launches <= 64, "launches must be in [0, 64]");
  HIPC(hipSetDevice(g->device));
  // only the tile kernel's weighted split has something to measure: a launch over >= 512 gallery tiles (launch_scatter_records)
  if (launches == 0 || !g->xcc_balance || g->npad / TILE < 8 * 64 || g->force_exact) return MI_OK;
  const int32_t nq = (int32_t)std::min<int64_t>(QB, g->n);
  const int32_t k = (int32_t)std::min<int64_t>(100, g->n);
  int rc = check_k(g, k);
  if (rc != MI_OK) return rc;
  if ((rc = ws_ensure(g, k)) != MI_OK) return rc;
  // the calibration launches reuse the query buffers of the searches before them: an asynchronous tail (deferred or already
  // running on the handle's own stream) must be done with them first
  if (g->pending.valid && (rc = flush_pending_tail(g, (hipStream_t)stream, false)) != MI_OK) return rc;
  for (int i = 0; i < 2; ++i)
    if (g->ev_tail_valid[i]) HIPC(hipStreamWaitEvent((hipStream_t)stream, g->ev_tail[i], 0));
  // queries = the first stored rows of the gallery itself (resident, already in the gallery's own normalisation): what the
  // launches score is irrelevant, every workgroup's loop time is what block 0 of the scatter kernel turns into shares
  for (int32_t i = 0; i < launches; ++i)
    if ((rc = phase1_batch(g, g->gal_f32, MI_F32, g->dp, 1, MI_NORM_NONE, nq, k, false, (hipStream_t)stream)) != MI_OK) return rc;
  // the answers are discarded, and so is whatever these launches flagged (a gallery that starts with duplicate rows can
  // overflow their candidate lists): sticky flags raised from here on belong to real searches again
  HIPC(hipMemsetAsync(g->ws.flags, 0, 4, (hipStream_t)stream));
  return MI_OK;
}

int mi_search_join(mi_gallery* g, void* stream) {
  REQUIRE(g, "null handle");
  HIPC(hipSetDevice(g->device));
  return join_tails(g, (hipStream_t)stream);
}

int mi_knn_phase1_device(mi_gallery* g, const float* q_dev, int64_t nq, int32_t k, float* out_approx_dev,
                         void* stream) {
  REQUIRE(g && q_dev && out_approx_dev, "null pointer");
  REQUIRE(nq >= 1 && nq <= QB, "phase API handles one batch of at most 1024 queries");
  HIPC(hipSetDevice(g->device));
  REQUIRE(k >= 1, "k must be >= 1");
  // a shard may hold fewer than k rows: clamp the local k, pad the tail with -inf
  const int32_t kl = (int32_t)std::min<int64_t>(k, g->n);
  int rc = check_k(g, kl);
  if (rc != MI_OK) return rc;
  if ((rc = ws_ensure(g, k)) != MI_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = phase1_batch(g, q_dev, MI_F32, g->d, 1, g->qnorm_override >= 0 ? g->qnorm_override : g->norm_mode,
                         (int32_t)nq, kl, g->force_exact != 0, s)) != MI_OK)
    return rc;
  if (kl == k) {
    HIPC(hipMemcpyAsync(out_approx_dev, g->ws.topvals, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, s));
  } else {
    std::vector<float> ninf((size_t)nq * k, -INFINITY);
    HIPC(hipMemcpyAsync(out_approx_dev, ninf.data(), ninf.size() * 4, hipMemcpyHostToDevice, s));
    HIPC(hipStreamSynchronize(s));
    HIPC(hipMemcpy2DAsync(out_approx_dev, (size_t)k * 4, g->ws.topvals, (size_t)kl * 4, (size_t)kl * 4, (size_t)nq,
                          hipMemcpyDeviceToDevice, s));
  }
  g->stats.searches += 1;
  g->stats.queries += nq;
  return MI_OK;
}

int mi_kth_of_gathered_device(const float* gathered_dev, int32_t nshards, int64_t nq, int32_t k, float* out_L_dev,
                              void* stream) {
  REQUIRE(gathered_dev && out_L_dev, "null pointer");
  REQUIRE(nshards >= 1 && (int64_t)nshards * k <= 8192, "nshards * k too large (the merge takes at most 8192 entries per query)");
  launch_kth_of_gathered(gathered_dev, nshards, nq, k, out_L_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_knn_phase2_device(mi_gallery* g, int64_t nq, int32_t k, const float* L_dev, int64_t* out_idx_dev,
                         float* out_score_dev, double* out_score64_dev, void* stream) {
  REQUIRE(g && L_dev && out_idx_dev, "null pointer");
  REQUIRE(nq >= 1 && nq <= QB, "phase API handles one batch of at most 1024 queries");
  REQUIRE(g->ws.qcap > 0, "phase 2 without phase 1");
  HIPC(hipSetDevice(g->device));
  return phase2_batch(g, (int32_t)nq, k, L_dev, out_idx_dev, out_score_dev, out_score64_dev, (hipStream_t)stream);
}

int mi_topk_merge_device(const double* score64_dev, const int64_t* idx_dev, int32_t nshards, int64_t nq, int32_t k,
                         int64_t* out_idx_dev, float* out_score_dev, void* stream) {
  REQUIRE(score64_dev && idx_dev && out_idx_dev, "null pointer");
  REQUIRE(nshards >= 1 && (int64_t)nshards * k <= 8192, "nshards * k too large");
  launch_merge(score64_dev, idx_dev, nshards, nq, k, nq * (int64_t)k, out_idx_dev, out_score_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_topk_merge_strided_device(const double* score64_dev, const int64_t* idx_dev, int64_t shard_stride, int32_t nshards,
                                 int64_t nq, int32_t k, int64_t* out_idx_dev, float* out_score_dev, void* stream) {
  REQUIRE(score64_dev && idx_dev && out_idx_dev, "null pointer");
  REQUIRE(nshards >= 1 && (int64_t)nshards * k <= 8192, "nshards * k too large");
  REQUIRE(shard_stride >= nq * (int64_t)k, "shard_stride smaller than one shard's list");
  launch_merge(score64_dev, idx_dev, nshards, nq, k, shard_stride, out_idx_dev, out_score_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

// ---- alpha query expansion ---------------------------------------------------------------------------
int mi_aqe_partial_device(mi_gallery* g, const int64_t* ranks_dev, int64_t rank_stride_j, int64_t rank_stride_q,
                          int64_t nq, int32_t k_qe, double w, double* out_sum_dev, void* stream) {
  REQUIRE(g && ranks_dev && out_sum_dev, "null pointer");
  REQUIRE(nq >= 1 && k_qe >= 1, "bad sizes");
  HIPC(hipSetDevice(g->device));
  launch_aqe_partial(g->gal_f32, g->dp, g->d, g->n, g->row_offset, ranks_dev, rank_stride_j, rank_stride_q, nq, k_qe,
                     w, nullptr, out_sum_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_aqe_rows_device(mi_gallery* g, const int64_t* ranks_dev, int64_t rank_stride_j, int64_t rank_stride_q, int64_t nq,
                       int32_t k_qe, float* out_rows_dev, void* stream) {
  REQUIRE(g && ranks_dev && out_rows_dev, "null pointer");
  REQUIRE(nq >= 1 && k_qe >= 1 && k_qe <= 65535, "bad sizes");
  HIPC(hipSetDevice(g->device));
  launch_aqe_rows(g->gal_f32, g->dp, g->d, g->n, g->row_offset, ranks_dev, rank_stride_j, rank_stride_q, nq, k_qe,
                  out_rows_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_aqe_combine_device(const float* rows_dev, int64_t nq, int32_t d, int32_t k_qe, double w, double* out_sum_dev,
                          void* stream) {
  REQUIRE(rows_dev && out_sum_dev, "null pointer");
  REQUIRE(nq >= 1 && k_qe >= 1 && d >= 1, "bad sizes");
  launch_aqe_combine(rows_dev, nq, d, k_qe, w, nullptr, out_sum_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_aqe_finish_device(const double* sum_dev, int64_t nq, int32_t d, double eps, float* out_q_dev,
                         double* out_q64_dev, void* stream) {
  REQUIRE(sum_dev && out_q_dev, "null pointer");
  launch_aqe_finish(sum_dev, nq, d, eps, out_q_dev, out_q64_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_aqe_search(mi_gallery* g, const int64_t* ranks, int64_t rank_stride_j, int64_t rank_stride_q, int64_t nq,
                  int32_t k_qe, double w, double eps, int32_t k, int64_t* out_idx, float* out_score, double* out_qexp,
                  double* out_seconds) {
  REQUIRE(g && ranks && out_idx, "null pointer");
  REQUIRE(nq >= 1 && k_qe >= 1, "bad sizes");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  const auto t0 = std::chrono::steady_clock::now();
  int64_t elems;
  int rc = strided_extent(k_qe, nq, rank_stride_j, rank_stride_q, &elems);
  if (rc != MI_OK) return rc;
  std::vector<void*> tmp;
  auto alloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes + 256) != hipSuccess) return nullptr;
    tmp.push_back(p);
    return p;
  };
  auto done = [&](int code) {
    for (void* p : tmp) (void)hipFree(p);
    return code;
  };
  int64_t* ranks_d = (int64_t*)alloc((size_t)elems * 8);
  double* sum_d = (double*)alloc((size_t)nq * g->d * 8);
  double* q64_d = (double*)alloc((size_t)nq * g->d * 8);
  float* q_d = (float*)alloc((size_t)nq * g->d * 4);
  int64_t* idx_d = (int64_t*)alloc((size_t)nq * k * 8);
  float* sc_d = (float*)alloc((size_t)nq * k * 4);
  if (!ranks_d || !sum_d || !q64_d || !q_d || !idx_d || !sc_d) return done(fail(MI_ERR_NOMEM, "aqe buffers"));
  // validate the row ids on the host: an out-of-range id must not become a wild gather
  for (int64_t j = 0; j < k_qe; ++j)
    for (int64_t q = 0; q < nq; ++q) {
      const int64_t v = ranks[j * rank_stride_j + q * rank_stride_q] - g->row_offset;
      if (v < 0 || v >= g->n) return done(fail(MI_ERR_INVALID, "rank id outside the gallery"));
    }
  if (hipMemcpy(ranks_d, ranks, (size_t)elems * 8, hipMemcpyHostToDevice) != hipSuccess)
    return done(fail(MI_ERR_HIP, "H2D ranks copy failed"));
  hipStream_t s = g->stream;
  launch_aqe_partial(g->gal_f32, g->dp, g->d, g->n, g->row_offset, ranks_d, rank_stride_j, rank_stride_q, nq, k_qe, w,
                     nullptr, sum_d, s);
  launch_aqe_finish(sum_d, nq, g->d, eps, q_d, q64_d, s);
  // the expanded query is used as is (no second normalisation), like `np.dot(vecs.T, qvecs_qe)`
  rc = search_sync(g, q_d, MI_F32, g->d, 1, MI_NORM_NONE, nq, k, idx_d, sc_d, nullptr);
  if (rc != MI_OK) return done(rc);
  if (hipMemcpy(out_idx, idx_d, (size_t)nq * k * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return done(fail(MI_ERR_HIP, "D2H idx copy failed"));
  if (out_score && hipMemcpy(out_score, sc_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return done(fail(MI_ERR_HIP, "D2H score copy failed"));
  if (out_qexp && hipMemcpy(out_qexp, q64_d, (size_t)nq * g->d * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return done(fail(MI_ERR_HIP, "D2H qexp copy failed"));
  if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return done(MI_OK);
}

}  // extern "C"

// ---- dense exact kNN (k a large fraction of N) and truncated graph diffusion ---------------------------------------
namespace {

// exact f32 inner products of every stored row with nq (device, strided) queries -> top-k, all on `s`
int dense_search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t rs, int64_t cs, int q_norm, int64_t nq,
                        int32_t k, int64_t* out_idx_dev, float* out_score_dev, hipStream_t s) {
  REQUIRE(k >= 1 && (int64_t)k <= g->n, "k must be in [1, N]");
  REQUIRE(k <= 4096, "dense top-k supports k <= 4096");
  int rc = ws_ensure(g, std::min<int32_t>(k, 1024));
  if (rc != MI_OK) return rc;
  // these launches overwrite the workspace's query buffers: a deferred / running asynchronous tail of an earlier device-API
  // batch (async_tail 1..3) must be done with them first
  if ((rc = join_tails(g, s)) != MI_OK) return rc;
  Workspace& ws = g->ws;
  int64_t qb = std::min<int64_t>(QB, std::max<int64_t>(64, ((int64_t)1 << 28) / g->n / 64 * 64));
  TmpAlloc tmp;
  float* dense = tmp.get<float>((size_t)qb * g->n);
  if (!dense) return fail(MI_ERR_NOMEM, "dense score buffer");
  const size_t esz = q_dtype == MI_F32 ? 4 : 8;
  for (int64_t q0 = 0; q0 < nq; q0 += qb) {
    const int32_t b = (int32_t)std::min<int64_t>(qb, nq - q0);
    const int32_t qpad = (int32_t)round_up(b, TILE);
    launch_ingest((const char*)q_src + (size_t)q0 * rs * esz, q_dtype, b, g->d, rs, cs, q_norm, ws.q_f32, ws.q_img,
                  g->img_f16, ws.q_stat, g->dp, qpad, s);
    ExactArgs a;
    a.gal_f32 = g->gal_f32;
    a.qry_f32 = ws.q_f32;
    a.dp = g->dp;
    a.row0 = 0;
    a.row1 = g->n;
    a.n = g->n;
    a.nq = b;
    a.st = make_state(ws);
    a.dense_out = dense;
    a.dense_ld = g->n;
    launch_exact_select(a, false, s);
    launch_dense_topk(dense, g->n, g->n, b, k, g->row_offset, out_idx_dev + q0 * k,
                      out_score_dev ? out_score_dev + q0 * k : nullptr, s);
    HIPC(hipGetLastError());
  }
  HIPC(hipStreamSynchronize(s));      // `dense` is freed on return
  return MI_OK;
}
}  // namespace

// the same at full precision (search_sync's last resort): dense f64 scores of a sub-batch of queries (<= 2 GiB of scores
// at a time), exact top-k by (f64 score desc, idx asc)
static int dense64_search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t rs, int64_t cs, int q_norm,
                                 int64_t nq, int32_t k, int64_t* out_idx_dev, float* out_score_dev, double* out_score64_dev,
                                 hipStream_t s) {
  REQUIRE(k >= 1 && (int64_t)k <= g->n && k <= 4096, "dense top-k supports k <= min(N, 4096)");
  int rc = ws_ensure(g, std::min<int32_t>(k, 1024));
  if (rc != MI_OK) return rc;
  // these launches overwrite the workspace's query buffers: a deferred / running asynchronous tail of an earlier device-API
  // batch (async_tail 1..3) must be done with them first
  if ((rc = join_tails(g, s)) != MI_OK) return rc;
  Workspace& ws = g->ws;
  const int64_t qb = std::min<int64_t>(QB, std::max<int64_t>(16, ((int64_t)1 << 28) / g->n));
  TmpAlloc tmp;
  double* dense = tmp.get<double>((size_t)qb * g->n);
  if (!dense) return fail(MI_ERR_NOMEM, "dense f64 score buffer");
  const size_t esz = q_dtype == MI_F32 ? 4 : 8;
  for (int64_t q0 = 0; q0 < nq; q0 += qb) {
    const int32_t b = (int32_t)std::min<int64_t>(qb, nq - q0);
    const int32_t qpad = (int32_t)round_up(b, TILE);
    launch_ingest((const char*)q_src + (size_t)q0 * rs * esz, q_dtype, b, g->d, rs, cs, q_norm, ws.q_f32, ws.q_img,
                  g->img_f16, ws.q_stat, g->dp, qpad, s);
    launch_dense_score64(g->gal_f32, ws.q_f32, g->dp, g->n, b, dense, g->n, s);
    launch_dense_topk64(dense, g->n, g->n, b, k, g->row_offset, out_idx_dev + q0 * k,
                        out_score_dev ? out_score_dev + q0 * k : nullptr,
                        out_score64_dev ? out_score64_dev + q0 * k : nullptr, s);
    HIPC(hipGetLastError());
  }
  HIPC(hipStreamSynchronize(s));      // `dense` is freed on return
  return MI_OK;
}

extern "C" {

int mi_knn_dense_search(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                        int32_t k, int64_t* out_idx, float* out_score, double* out_seconds) {
  REQUIRE(g && q && out_idx, "null pointer");
  REQUIRE(nq >= 1, "no queries");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  const auto t0 = std::chrono::steady_clock::now();
  int64_t elems;
  int rc = strided_extent(nq, g->d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  TmpAlloc tmp;
  char* qd = tmp.get<char>((size_t)elems * esz);
  int64_t* idx_d = tmp.get<int64_t>((size_t)nq * k);
  float* sc_d = tmp.get<float>((size_t)nq * k);
  if (!qd || !idx_d || !sc_d) return fail(MI_ERR_NOMEM, "dense search buffers");
  HIPC(hipMemcpy(qd, q, (size_t)elems * esz, hipMemcpyHostToDevice));
  if ((rc = dense_search_device(g, qd, dtype, row_stride, col_stride, g->norm_mode, nq, k, idx_d, sc_d, g->stream)) != MI_OK)
    return rc;
  HIPC(hipMemcpy(out_idx, idx_d, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
  if (out_score) HIPC(hipMemcpy(out_score, sc_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
  if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return MI_OK;
}

extern "C" int mi_knn_dense64_search(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                                     int32_t k, int64_t* out_idx, float* out_score, double* out_score64, double* out_seconds) {
  REQUIRE(g && q && out_idx, "null pointer");
  REQUIRE(nq >= 1, "no queries");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  const auto t0 = std::chrono::steady_clock::now();
  int64_t elems;
  int rc = strided_extent(nq, g->d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  TmpAlloc tmp;
  char* qd = tmp.get<char>((size_t)elems * esz);
  int64_t* idx_d = tmp.get<int64_t>((size_t)nq * k);
  float* sc_d = tmp.get<float>((size_t)nq * k);
  double* sc64_d = tmp.get<double>((size_t)nq * k);
  if (!qd || !idx_d || !sc_d || !sc64_d) return fail(MI_ERR_NOMEM, "dense search buffers");
  HIPC(hipMemcpy(qd, q, (size_t)elems * esz, hipMemcpyHostToDevice));
  if ((rc = dense64_search_device(g, qd, dtype, row_stride, col_stride, g->norm_mode, nq, k, idx_d, sc_d, sc64_d, g->stream)) !=
      MI_OK)
    return rc;
  HIPC(hipMemcpy(out_idx, idx_d, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
  if (out_score) HIPC(hipMemcpy(out_score, sc_d, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
  if (out_score64) HIPC(hipMemcpy(out_score64, sc64_d, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
  if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return MI_OK;
}

// full-length ranking of every query, of which the first `keep` columns are copied out (keep = N: the whole ranking)
static int rank_all_impl(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                         int query_norm, int64_t keep, int64_t* out_idx, float* out_score, double* out_seconds) {
  REQUIRE(g && q && out_idx, "null pointer");
  REQUIRE(nq >= 1, "no queries");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  REQUIRE(query_norm >= -1 && query_norm <= 2, "query_norm: -1 (as the gallery) or an mi_norm value");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  REQUIRE(keep >= 1 && keep <= g->n, "keep must be in [1, N]");
  const auto t0 = std::chrono::steady_clock::now();
  const int qn = query_norm < 0 ? g->norm_mode : query_norm;
  int64_t elems;
  int rc = strided_extent(nq, g->d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  if ((rc = ws_ensure(g, 1)) != MI_OK) return rc;
  if ((rc = join_tails(g, g->stream)) != MI_OK) return rc;     // (an asynchronous tail still reads the query buffers)
  Workspace& ws = g->ws;
  hipStream_t s = g->stream;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  const int64_t n = g->n;
  const int64_t qb = std::max<int64_t>(1, std::min<int64_t>(128, ((int64_t)1 << 28) / n));
  TmpAlloc tmp;
  char* qd = tmp.get<char>((size_t)elems * esz);
  float* dense = tmp.get<float>((size_t)round_up(qb, 64) * n);
  uint32_t* ka = tmp.get<uint32_t>((size_t)qb * n);
  uint32_t* ia = tmp.get<uint32_t>((size_t)qb * n);
  uint32_t* kb = tmp.get<uint32_t>((size_t)qb * n);
  uint32_t* ib = tmp.get<uint32_t>((size_t)qb * n);
  int64_t* oi = tmp.get<int64_t>((size_t)qb * n);
  float* os = out_score ? tmp.get<float>((size_t)qb * n) : nullptr;
  if (!qd || !dense || !ka || !ia || !kb || !ib || !oi || (out_score && !os)) return fail(MI_ERR_NOMEM, "rank_all buffers");
  HIPC(hipMemcpy(qd, q, (size_t)elems * esz, hipMemcpyHostToDevice));
  for (int64_t q0 = 0; q0 < nq; q0 += qb) {
    const int32_t b = (int32_t)std::min<int64_t>(qb, nq - q0);
    const int32_t qpad = (int32_t)round_up(b, TILE);
    launch_ingest(qd + (size_t)q0 * row_stride * esz, dtype, b, g->d, row_stride, col_stride, qn, ws.q_f32, ws.q_img,
                  g->img_f16, ws.q_stat, g->dp, qpad, s);
    ExactArgs a;
    a.gal_f32 = g->gal_f32;
    a.qry_f32 = ws.q_f32;
    a.dp = g->dp;
    a.row0 = 0;
    a.row1 = n;
    a.n = n;
    a.nq = b;
    a.st = make_state(ws);
    a.dense_out = dense;
    a.dense_ld = n;
    launch_exact_select(a, false, s);
    launch_rank_all(dense, n, n, b, ka, ia, kb, ib, g->row_offset, oi, os, s);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(s));
    HIPC(hipMemcpy2D(out_idx + q0 * keep, (size_t)keep * 8, oi, (size_t)n * 8, (size_t)keep * 8, (size_t)b,
                     hipMemcpyDeviceToHost));
    if (out_score)
      HIPC(hipMemcpy2D(out_score + q0 * keep, (size_t)keep * 4, os, (size_t)n * 4, (size_t)keep * 4, (size_t)b,
                       hipMemcpyDeviceToHost));
  }
  if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return MI_OK;
}

int mi_rank_all(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                int query_norm, int64_t* out_idx, float* out_score, double* out_seconds) {
  REQUIRE(g, "null handle");
  return rank_all_impl(g, q, nq, dtype, row_stride, col_stride, query_norm, g->n, out_idx, out_score, out_seconds);
}

int mi_rank_prefix(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                   int query_norm, int64_t keep, int64_t* out_idx, float* out_score, double* out_seconds) {
  REQUIRE(g, "null handle");
  return rank_all_impl(g, q, nq, dtype, row_stride, col_stride, query_norm, keep, out_idx, out_score, out_seconds);
}

int mi_rank_positions(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                      int query_norm, const int64_t* row_ids, int32_t m, int64_t* out_pos) {
  REQUIRE(g && q && row_ids && out_pos, "null pointer");
  REQUIRE(nq >= 1 && m >= 1, "bad sizes");
  REQUIRE(m <= rank_positions_max_listed(), "too many listed rows per query (max 2048)");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  REQUIRE(query_norm >= -1 && query_norm <= 2, "query_norm: -1 (as the gallery) or an mi_norm value");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  const int qn = query_norm < 0 ? g->norm_mode : query_norm;
  int64_t elems;
  int rc = strided_extent(nq, g->d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  if ((rc = ws_ensure(g, 1)) != MI_OK) return rc;
  if ((rc = join_tails(g, g->stream)) != MI_OK) return rc;     // (an asynchronous tail still reads the query buffers)
  Workspace& ws = g->ws;
  hipStream_t s = g->stream;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  const int64_t n = g->n;
  const int64_t qb = std::max<int64_t>(1, std::min<int64_t>(QB, ((int64_t)1 << 28) / n));
  TmpAlloc tmp;
  char* qd = tmp.get<char>((size_t)elems * esz);
  float* dense = tmp.get<float>((size_t)round_up(qb, 64) * n);
  int64_t* ids_d = tmp.get<int64_t>((size_t)nq * m);
  unsigned long long* pos_d = tmp.get<unsigned long long>((size_t)nq * m);
  if (!qd || !dense || !ids_d || !pos_d) return fail(MI_ERR_NOMEM, "rank_positions buffers");
  HIPC(hipMemcpy(qd, q, (size_t)elems * esz, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(ids_d, row_ids, (size_t)nq * m * 8, hipMemcpyHostToDevice));
  HIPC(hipMemsetAsync(pos_d, 0, (size_t)nq * m * 8, s));
  for (int64_t q0 = 0; q0 < nq; q0 += qb) {
    const int32_t b = (int32_t)std::min<int64_t>(qb, nq - q0);
    const int32_t qpad = (int32_t)round_up(b, TILE);
    launch_ingest(qd + (size_t)q0 * row_stride * esz, dtype, b, g->d, row_stride, col_stride, qn, ws.q_f32, ws.q_img,
                  g->img_f16, ws.q_stat, g->dp, qpad, s);
    ExactArgs a;
    a.gal_f32 = g->gal_f32;
    a.qry_f32 = ws.q_f32;
    a.dp = g->dp;
    a.row0 = 0;
    a.row1 = n;
    a.n = n;
    a.nq = b;
    a.st = make_state(ws);
    a.dense_out = dense;
    a.dense_ld = n;
    launch_exact_select(a, false, s);
    launch_rank_positions(dense, n, n, b, ids_d + q0 * m, m, g->row_offset, pos_d + q0 * m, s);
    HIPC(hipGetLastError());
  }
  HIPC(hipStreamSynchronize(s));
  HIPC(hipMemcpy(out_pos, pos_d, (size_t)nq * m * 8, hipMemcpyDeviceToHost));
  // ids outside this shard (padding, -1) get position -1
  for (int64_t i = 0; i < nq * m; ++i) {
    const int64_t id = row_ids[i] - g->row_offset;
    if (id < 0 || id >= n) out_pos[i] = -1;
  }
  return MI_OK;
}

int mi_gather_weighted(mi_gallery* g, const int64_t* ranks, int64_t rank_stride_j, int64_t rank_stride_q, int64_t nq,
                       int32_t k, const double* weights, double* out_sum) {
  REQUIRE(g && ranks && weights && out_sum, "null pointer");
  REQUIRE(nq >= 1 && k >= 1, "bad sizes");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  int64_t elems;
  int rc = strided_extent(k, nq, rank_stride_j, rank_stride_q, &elems);
  if (rc != MI_OK) return rc;
  for (int64_t j = 0; j < k; ++j)
    for (int64_t q = 0; q < nq; ++q) {
      const int64_t v = ranks[j * rank_stride_j + q * rank_stride_q] - g->row_offset;
      if (v < 0 || v >= g->n) return fail(MI_ERR_INVALID, "rank id outside the gallery");
    }
  TmpAlloc tmp;
  int64_t* rd = tmp.get<int64_t>((size_t)elems);
  double* wd = tmp.get<double>((size_t)k);
  double* sd = tmp.get<double>((size_t)nq * g->d);
  if (!rd || !wd || !sd) return fail(MI_ERR_NOMEM, "gather buffers");
  HIPC(hipMemcpy(rd, ranks, (size_t)elems * 8, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(wd, weights, (size_t)k * 8, hipMemcpyHostToDevice));
  launch_aqe_partial(g->gal_f32, g->dp, g->d, g->n, g->row_offset, rd, rank_stride_j, rank_stride_q, nq, k, 0.0, wd, sd,
                     g->stream);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(g->stream));
  HIPC(hipMemcpy(out_sum, sd, (size_t)nq * g->d * 8, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_column_sum(const void* X, int64_t n, int32_t d, int dtype, int64_t row_stride, int64_t col_stride, int device,
                  double* out) {
  REQUIRE(X && out, "null pointer");
  REQUIRE(n >= 1 && d >= 1, "bad sizes");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  HIPC(hipSetDevice(device));
  int64_t elems;
  int rc = strided_extent(n, d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  TmpAlloc tmp;
  char* xd = tmp.get<char>((size_t)elems * esz);
  double* od = tmp.get<double>((size_t)d);
  if (!xd || !od) return fail(MI_ERR_NOMEM, "column sum buffers");
  HIPC(hipMemcpy(xd, X, (size_t)elems * esz, hipMemcpyHostToDevice));
  launch_column_sum(xd, dtype, n, d, row_stride, col_stride, od, nullptr);
  HIPC(hipGetLastError());
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpy(out, od, (size_t)d * 8, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_whiten_apply(const void* X, int64_t n, int32_t d, int dtype, int64_t row_stride, int64_t col_stride,
                    const double* m, const double* P, int32_t dims, double eps, int device, double* out) {
  REQUIRE(X && m && P && out, "null pointer");
  REQUIRE(n >= 1 && d >= 1 && dims >= 1 && dims <= d, "bad sizes (dims must be in [1, d])");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  HIPC(hipSetDevice(device));
  int64_t elems;
  int rc = strided_extent(n, d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  TmpAlloc tmp;
  char* xd = tmp.get<char>((size_t)elems * esz);
  double* md = tmp.get<double>((size_t)d);
  double* pd = tmp.get<double>((size_t)dims * d);
  double* yd = tmp.get<double>((size_t)n * dims);
  if (!xd || !md || !pd || !yd) return fail(MI_ERR_NOMEM, "whiten buffers");
  HIPC(hipMemcpy(xd, X, (size_t)elems * esz, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(md, m, (size_t)d * 8, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(pd, P, (size_t)dims * d * 8, hipMemcpyHostToDevice));
  launch_whiten(xd, dtype, n, d, row_stride, col_stride, md, pd, dims, eps, yd, nullptr);
  HIPC(hipGetLastError());
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpy(out, yd, (size_t)n * dims * 8, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_whiten_apply_device(const void* X_dev, int64_t n, int32_t d, int dtype, int64_t row_stride, int64_t col_stride,
                           const double* m_dev, const double* P_dev, int32_t dims, double eps, double* out_dev, void* stream) {
  REQUIRE(X_dev && m_dev && P_dev && out_dev, "null pointer");
  REQUIRE(n >= 1 && d >= 1 && dims >= 1 && dims <= d, "bad sizes (dims must be in [1, d])");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  REQUIRE(row_stride >= 0 && col_stride >= 0, "negative strides are not supported");
  REQUIRE((n + 127) / 128 * ((dims + 127) / 128) < ((int64_t)1 << 31), "too many tiles for one launch");
  launch_whiten(X_dev, dtype, n, d, row_stride, col_stride, m_dev, P_dev, dims, eps, out_dev, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

int mi_gallery_append_whitened_device(mi_gallery* g, const void* X_dev, int64_t m, int32_t d, int dtype, int64_t row_stride,
                                      int64_t col_stride, const double* mean_dev, const double* P_dev, void* stream) {
  REQUIRE(g && X_dev && mean_dev && P_dev, "null pointer");
  REQUIRE(m >= 1 && d >= 1 && g->d <= d, "bad sizes (the gallery's dimension is the number of rows of P that are applied)");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  REQUIRE(row_stride >= 0 && col_stride >= 0, "negative strides are not supported");
  REQUIRE(g->n + m <= g->cap, "gallery capacity exceeded");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  hipStream_t s = (hipStream_t)stream;
  const int32_t dims = g->d;
  const int64_t chunk = std::min<int64_t>(m, 32768);       // 512 MiB of float64 rows at dims = 2048: the only scratch there is
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  TmpAlloc tmp;
  double* y = tmp.get<double>((size_t)chunk * dims);
  if (!y) return fail(MI_ERR_NOMEM, "whitening scratch block");
  for (int64_t r0 = 0; r0 < m; r0 += chunk) {
    const int64_t rows = std::min<int64_t>(chunk, m - r0);
    // P (x - m) of the chunk in float64, un-normalised; the ingest normalises while it reads (MI_NORM_L2_EPS = whitenapply's
    // `X / (norm + 1e-6)`, src/utils/whiten.py:10) and writes f32 rows, 16-bit image and rounding norms at their final place
    launch_whiten((const char*)X_dev + (size_t)r0 * row_stride * esz, dtype, rows, d, row_stride, col_stride, mean_dev, P_dev,
                  dims, -1.0, y, s);
    launch_ingest(y, MI_F64, rows, dims, dims, 1, g->norm_mode, g->gal_f32, g->gal_img, g->img_f16, g->rowstat, g->dp, rows, s,
                  g->n + r0);
  }
  launch_rowstat_max(g->rowstat + g->n, m, g->gstat3, s, /*reset=*/false);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(s));           // the scratch block is freed on return
  g->n += m;
  g->npad = round_up(g->n, TILE);
  return MI_OK;
}

int mi_kr_rerank(const void* qvecs, int64_t nq, int64_t q_row_stride, int64_t q_col_stride, const void* vecs, int64_t n,
                 int64_t v_row_stride, int64_t v_col_stride, int32_t d, int dtype, int32_t k1, int32_t k2,
                 double lambda_value, int device, int64_t* out_idx, float* out_dist) {
  REQUIRE(qvecs && vecs && out_idx, "null pointer");
  REQUIRE(nq >= 1 && n >= 1 && d >= 1, "bad sizes");
  REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  const int64_t all = nq + n;
  REQUIRE(all <= 32768, "k-reciprocal re-ranking holds all x all arrays: at most 32768 images (queries + gallery)");
  REQUIRE(k1 >= 1 && k1 + 1 <= 64 && k1 + 1 <= all, "k1 must be in [1, min(63, all - 1)]");
  REQUIRE(k2 >= 1 && k2 <= k1 + 1, "k2 must be in [1, k1 + 1]");
  const int khalf = (int)nearbyint(k1 / 2.0);
  REQUIRE((k1 + 1) * (khalf + 2) <= kr_rmax(), "k1 too large for the reciprocal-set buffers");
  HIPC(hipSetDevice(device));
  int64_t qe, ve;
  int rc = strided_extent(nq, d, q_row_stride, q_col_stride, &qe);
  if (rc == MI_OK) rc = strided_extent(n, d, v_row_stride, v_col_stride, &ve);
  if (rc != MI_OK) return rc;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  const int32_t dp = (int32_t)round_up(d, 16);
  const int64_t all_pad = round_up(all, 64);
  const int ld = k1 + 1;
  TmpAlloc tmp;
  char* qd = tmp.get<char>((size_t)qe * esz);
  char* vd = tmp.get<char>((size_t)ve * esz);
  float* feat = tmp.get<float>((size_t)all_pad * dp);
  float* S = tmp.get<float>((size_t)all * all);
  int64_t* rank = tmp.get<int64_t>((size_t)all * ld);
  int32_t* R = tmp.get<int32_t>((size_t)all * kr_rmax());
  int32_t* Rcnt = tmp.get<int32_t>((size_t)all);
  float* V = tmp.get<float>((size_t)all * kr_rmax());
  float* dmax = tmp.get<float>((size_t)all);
  uint16_t* Vqe = tmp.get<uint16_t>((size_t)all * all);
  uint16_t* VqeT = tmp.get<uint16_t>((size_t)all * all);
  uint32_t* flags = tmp.get<uint32_t>(4);
  float* negf = tmp.get<float>((size_t)nq * n);
  uint32_t* ka = tmp.get<uint32_t>((size_t)nq * n);
  uint32_t* ia = tmp.get<uint32_t>((size_t)nq * n);
  uint32_t* kb = tmp.get<uint32_t>((size_t)nq * n);
  uint32_t* ib = tmp.get<uint32_t>((size_t)nq * n);
  int64_t* oi = tmp.get<int64_t>((size_t)nq * n);
  float* os = tmp.get<float>((size_t)nq * n);
  if (!qd || !vd || !feat || !S || !rank || !R || !Rcnt || !V || !dmax || !Vqe || !VqeT || !flags || !negf || !ka || !ia ||
      !kb || !ib || !oi || !os)
    return fail(MI_ERR_NOMEM, "k-reciprocal buffers");
  hipStream_t s = nullptr;
  HIPC(hipMemcpy(qd, qvecs, (size_t)qe * esz, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(vd, vecs, (size_t)ve * esz, hipMemcpyHostToDevice));
  HIPC(hipMemsetAsync(feat, 0, (size_t)all_pad * dp * 4, s));
  HIPC(hipMemsetAsync(flags, 0, 16, s));
  // feat = [queries; gallery] (torch.cat([probFea, galFea]), :555)
  launch_kr_pack(qd, dtype, nq, d, q_row_stride, q_col_stride, feat, dp, s);
  launch_kr_pack(vd, dtype, n, d, v_row_stride, v_col_stride, feat + (size_t)nq * dp, dp, s);
  // all x all inner products, k-ordered f32 fmaf chains (both S[i, j] and S[j, i] are the same chain)
  ExactArgs a;
  a.gal_f32 = feat;
  a.qry_f32 = feat;
  a.dp = dp;
  a.row0 = 0;
  a.row1 = all;
  a.n = all;
  a.nq = (int32_t)all;
  a.st = QueryState{};
  a.dense_out = S;
  a.dense_ld = all;
  launch_exact_select(a, false, s);
  // initial_rank: the k1 + 1 nearest of every image among all images, itself included (:555)
  launch_dense_topk(S, all, all, (int32_t)all, ld, 0, rank, nullptr, s);
  launch_kr_sets(rank, ld, (int)all, k1, R, Rcnt, flags, s);
  launch_kr_weights(S, (int)all, R, Rcnt, V, dmax, s);
  launch_kr_expand(rank, ld, k2, (int)all, R, Rcnt, V, Vqe, VqeT, s);
  launch_kr_final(Vqe, VqeT, S, dmax, (int)all, (int)nq, (float)(1.0 - lambda_value), (float)lambda_value, negf, flags, s);
  // np.argsort(final_dist, axis=1) (:618): ascending distance = descending -distance, ties to the lower index
  launch_rank_all(negf, n, n, (int32_t)nq, ka, ia, kb, ib, 0, oi, os, s);
  HIPC(hipGetLastError());
  HIPC(hipDeviceSynchronize());
  uint32_t fl = 0;
  HIPC(hipMemcpy(&fl, flags, 4, hipMemcpyDeviceToHost));
  if (fl) return fail(MI_ERR_OVERFLOW, "k-reciprocal set buffers overflowed");
  HIPC(hipMemcpy(out_idx, oi, (size_t)nq * n * 8, hipMemcpyDeviceToHost));
  if (out_dist) {
    HIPC(hipMemcpy(out_dist, os, (size_t)nq * n * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nq * n; ++i) out_dist[i] = -out_dist[i];
  }
  return MI_OK;
}

int mi_diffusion_offline(mi_gallery* g, int32_t n_trunc, int32_t kd, double alpha, int32_t gamma, int32_t maxiter,
                         double tol, int64_t* out_ids, float* out_vals, float* out_knn_sims) {
  REQUIRE(g, "null handle");
  return mi_diffusion_offline_nodes(g, n_trunc, kd, alpha, gamma, maxiter, tol, 0, g->n, out_ids, out_vals, out_knn_sims);
}

int mi_diffusion_offline_nodes(mi_gallery* g, int32_t n_trunc, int32_t kd, double alpha, int32_t gamma, int32_t maxiter,
                               double tol, int64_t node0, int64_t node1, int64_t* out_ids, float* out_vals,
                               float* out_knn_sims) {
  REQUIRE(g, "null handle");
  REQUIRE(node0 >= 0 && node0 <= node1 && node1 <= g->n, "node range must lie inside [0, N]");
  REQUIRE(n_trunc >= 2 && (int64_t)n_trunc <= g->n && n_trunc <= 4096, "n_trunc must be in [2, min(N, 4096)]");
  REQUIRE(kd >= 1 && kd <= n_trunc, "kd must be in [1, n_trunc]");
  REQUIRE(g->n < ((int64_t)1 << 31), "too many rows");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  hipStream_t s = g->stream;
  const int64_t n = g->n;
  const int32_t T = n_trunc;
  TmpAlloc tmp;
  int64_t* ids = tmp.get<int64_t>((size_t)n * T);
  float* sims = tmp.get<float>((size_t)n * T);
  float* lap = tmp.get<float>((size_t)n * kd);
  float* dinv = tmp.get<float>((size_t)n);
  float* diag = tmp.get<float>((size_t)n);
  const unsigned grid = 512;
  int32_t* map_all = tmp.get<int32_t>((size_t)grid * n);
  if (!ids || !sims || !lap || !dinv || !diag || !map_all) return fail(MI_ERR_NOMEM, "diffusion buffers");
  // 1) kNN graph: the stored rows against themselves, exact inner product, top n_trunc (knn.search(features, n_trunc))
  int rc = dense_search_device(g, g->gal_f32, MI_F32, g->dp, 1, MI_NORM_NONE, n, T, ids, sims, s);
  if (rc != MI_OK) return rc;
  // 2) mutual-kNN affinity on the first kd columns, normalised Laplacian
  launch_affinity(ids, sims, T, n, kd, gamma, (float)alpha, lap, dinv, diag, s);
  // 3) truncated CG per node
  HIPC(hipMemsetAsync(map_all, 0xFF, (size_t)grid * n * 4, s));
  (void)hipFree(g->dif_ids);
  (void)hipFree(g->dif_vals);
  g->dif_ids = nullptr;
  g->dif_vals = nullptr;
  HIPC(hipMalloc((void**)&g->dif_ids, (size_t)n * T * 4));
  HIPC(hipMalloc((void**)&g->dif_vals, (size_t)n * T * 4));
  g->dif_T = T;
  // rows outside [node0, node1) stay zero until mi_diffusion_set_offline installs the gathered result
  HIPC(hipMemsetAsync(g->dif_vals, 0, (size_t)n * T * 4, s));
  HIPC(hipMemsetAsync(g->dif_ids, 0, (size_t)n * T * 4, s));
  if (node1 > node0)
    launch_diffusion_cg(ids, T, n, T, kd, lap, diag, maxiter, tol, map_all, grid, g->dif_ids, g->dif_vals, s, node0, node1);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(s));
  if (out_ids) HIPC(hipMemcpy(out_ids, ids, (size_t)n * T * 8, hipMemcpyDeviceToHost));
  if (out_vals && node1 > node0)
    HIPC(hipMemcpy(out_vals, g->dif_vals + (size_t)node0 * T, (size_t)(node1 - node0) * T * 4, hipMemcpyDeviceToHost));
  if (out_knn_sims) HIPC(hipMemcpy(out_knn_sims, sims, (size_t)n * T * 4, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_diffusion_set_offline(mi_gallery* g, const int64_t* ids, const float* vals, int32_t n_trunc) {
  REQUIRE(g && ids && vals && n_trunc >= 1, "bad arguments");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  const size_t cnt = (size_t)g->n * n_trunc;
  std::vector<int32_t> ids32(cnt);
  for (size_t i = 0; i < cnt; ++i) {
    if (ids[i] < 0 || ids[i] >= g->n) return fail(MI_ERR_INVALID, "offline id outside the gallery");
    ids32[i] = (int32_t)ids[i];
  }
  (void)hipFree(g->dif_ids);
  (void)hipFree(g->dif_vals);
  g->dif_ids = nullptr;
  g->dif_vals = nullptr;
  HIPC(hipMalloc((void**)&g->dif_ids, cnt * 4));
  HIPC(hipMalloc((void**)&g->dif_vals, cnt * 4));
  HIPC(hipMemcpy(g->dif_ids, ids32.data(), cnt * 4, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(g->dif_vals, vals, cnt * 4, hipMemcpyHostToDevice));
  g->dif_T = n_trunc;
  return MI_OK;
}

int mi_diffusion_online(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                        int32_t k_query, int32_t gamma, int32_t trunc, int64_t* out_ranks, float* out_scores) {
  REQUIRE(g && q && out_ranks, "null pointer");
  REQUIRE(g->dif_ids && g->dif_vals, "no offline diffusion result on this handle");
  REQUIRE(nq >= 1 && k_query >= 1 && (int64_t)k_query <= g->n, "bad sizes");
  REQUIRE(trunc >= 1 && (int64_t)trunc < g->n && trunc <= 4096, "trunc must be in [1, min(N-1, 4096)] (np.argpartition needs kth < N)");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  hipStream_t s = g->stream;
  int64_t elems;
  int rc = strided_extent(nq, g->d, row_stride, col_stride, &elems);
  if (rc != MI_OK) return rc;
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  TmpAlloc tmp;
  char* qd = tmp.get<char>((size_t)elems * esz);
  int64_t* nn_idx = tmp.get<int64_t>((size_t)nq * k_query);
  float* nn_sims = tmp.get<float>((size_t)nq * k_query);
  float* dense = tmp.get<float>((size_t)nq * g->n);
  int64_t* ranks_d = tmp.get<int64_t>((size_t)nq * trunc);
  float* sc_d = tmp.get<float>((size_t)nq * trunc);
  if (!qd || !nn_idx || !nn_sims || !dense || !ranks_d || !sc_d) return fail(MI_ERR_NOMEM, "diffusion online buffers");
  HIPC(hipMemcpy(qd, q, (size_t)elems * esz, hipMemcpyHostToDevice));
  // knn.search(q, k_query): exact top-k_query by inner product (the queries are used as given)
  if ((rc = search_sync(g, qd, dtype, row_stride, col_stride, MI_NORM_NONE, nq, k_query, nn_idx, nn_sims, nullptr)) != MI_OK)
    return rc;
  launch_diffusion_combine(nn_idx, nn_sims, k_query, gamma, g->dif_ids, g->dif_vals, g->dif_T, g->n, (int32_t)nq, dense, s);
  launch_dense_topk(dense, g->n, g->n, (int32_t)nq, trunc, 0, ranks_d, sc_d, s);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(s));
  HIPC(hipMemcpy(out_ranks, ranks_d, (size_t)nq * trunc * 8, hipMemcpyDeviceToHost));
  if (out_scores) HIPC(hipMemcpy(out_scores, sc_d, (size_t)nq * trunc * 4, hipMemcpyDeviceToHost));
  return MI_OK;
}

}  // extern "C"

extern "C" {

// ---- status / options ------------------------------------------------------------------------------
int mi_profile_enable(mi_gallery* g, int on) {
  REQUIRE(g, "null handle");
  g->profile = on != 0;
  return MI_OK;
}

int mi_search_status(mi_gallery* g, mi_search_stats* out, int reset) {
  REQUIRE(g, "null handle");
  HIPC(hipSetDevice(g->device));
  if (g->ev_stream) HIPC(hipStreamSynchronize(g->ev_stream));
  HIPC(hipStreamSynchronize(g->stream));
  HIPC(hipDeviceSynchronize());
  prof_collect(g);
  Workspace& sw = g->ws.qcap ? g->ws : g->ws_alt;       // flags / statistics / clocks are ONE set for both slots
  if (sw.qcap) {
    // per-query accumulators (the kernels add to the words of their own query: no atomics on one address from 1024
    // workgroups, which cost the final maintain launch 15 us per batch); summed here
    std::vector<uint64_t> s2(3 * (size_t)QB);
    HIPC(hipMemcpy(s2.data(), sw.stats2, s2.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)QB; ++i) {
      g->stats.survivors += (int64_t)s2[2 * i];
      g->stats.candidates += (int64_t)s2[2 * i + 1];
      g->stats.inkernel_repairs += (int64_t)s2[2 * (size_t)QB + i];
    }
    HIPC(hipMemset(sw.stats2, 0, s2.size() * 8));
    uint32_t flags = 0;
    HIPC(hipMemcpy(&flags, sw.flags, 4, hipMemcpyDeviceToHost));
    if (flags) {
      // sticky device flag: a device-API batch overflowed since the last status call.  Counted once and cleared, so
      // that polling without reset does not count the same event again.
      count_flagged_batch(g, flags);
      HIPC(hipMemset(sw.flags, 0, 4));
    }
    // in-kernel clock of the last tile-kernel launch: median over its waves of cycles / (10 ns ticks) x 100 MHz
    std::vector<unsigned long long> c((size_t)sw.nseg * 8);
    HIPC(hipMemcpy(c.data(), sw.dbg, c.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (size_t w = 0; w < sw.nseg; ++w)
      if (c[w * 8 + 7] > 0 && c[w * 8 + 6] > 0) mhz.push_back((double)c[w * 8 + 6] / (double)c[w * 8 + 7] * 100.0);
    if (!mhz.empty()) {
      std::nth_element(mhz.begin(), mhz.begin() + mhz.size() / 2, mhz.end());
      g->stats.kernel_clock_mhz = mhz[mhz.size() / 2];
    }
  }
  if (out) *out = g->stats;
  if (reset) {
    g->stats = mi_search_stats{};
    g->launch_ms_log.clear();
  }
  return MI_OK;
}

int mi_profile_launch_ms(mi_gallery* g, float* out_host, int64_t cap, int64_t* out_count) {
  REQUIRE(g && out_count, "null");
  HIPC(hipSetDevice(g->device));
  if (g->ev_stream) HIPC(hipStreamSynchronize(g->ev_stream));
  prof_collect(g);
  const int64_t n = (int64_t)g->launch_ms_log.size();
  *out_count = n;
  if (out_host)
    for (int64_t i = 0; i < std::min(n, cap); ++i) out_host[i] = g->launch_ms_log[(size_t)i];
  return MI_OK;
}

int mi_gallery_norm_bounds(mi_gallery* g, float* bounds3, int raise) {
  REQUIRE(g && bounds3, "null");
  std::lock_guard<std::mutex> lock(g->mu);
  HIPC(hipSetDevice(g->device));
  HIPC(hipStreamSynchronize(g->stream));
  float own[3] = {0, 0, 0};
  HIPC(hipMemcpy(own, g->gstat3, 12, hipMemcpyDeviceToHost));
  if (raise) {
    for (int i = 0; i < 3; ++i) {
      REQUIRE(bounds3[i] == bounds3[i], "NaN bound");
      own[i] = std::max(own[i], bounds3[i]);
    }
    HIPC(hipMemcpy(g->gstat3, own, 12, hipMemcpyHostToDevice));
  }
  for (int i = 0; i < 3; ++i) bounds3[i] = own[i];
  return MI_OK;
}

int mi_gallery_set_image_dtype(mi_gallery* g, int f16) {
  REQUIRE(g, "null handle");
  f16 = f16 != 0;
  std::lock_guard<std::mutex> lock(g->mu);
  if (g->img_f16 == f16 || g->n == 0) {
    g->img_f16 = f16;
    return MI_OK;
  }
  HIPC(hipSetDevice(g->device));
  HIPC(hipStreamSynchronize(g->stream));
  // the stored f32 rows are already normalised: re-round them into the other 16-bit type (rows, image and rounding
  // norms are rewritten; the f32 rows come out bit-identical)
  launch_ingest(g->gal_f32, MI_F32, g->n, g->d, g->dp, 1, MI_NORM_NONE, g->gal_f32, g->gal_img, f16, g->rowstat, g->dp,
                g->npad, g->stream);
  launch_rowstat_max(g->rowstat, g->n, g->gstat3, g->stream);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(g->stream));
  g->img_f16 = f16;
  g->samp_for_n = -1;          // the bootstrap sample image is rebuilt from the new image
  ws_free(g->ws);              // the query image buffers follow the element type
  ws_free(g->ws_alt);
  return MI_OK;
}

int mi_search_flags(mi_gallery* g, uint32_t* out_flags) {
  REQUIRE(g && out_flags, "null");
  HIPC(hipSetDevice(g->device));
  HIPC(hipStreamSynchronize(g->stream));
  HIPC(hipDeviceSynchronize());
  *out_flags = 0;
  Workspace& sw = g->ws.qcap ? g->ws : g->ws_alt;          // one set of flags for both workspace slots
  if (sw.qcap) {
    HIPC(hipMemcpy(out_flags, sw.flags, 4, hipMemcpyDeviceToHost));
    if (*out_flags) {
      HIPC(hipMemset(sw.flags, 0, 4));
      count_flagged_batch(g, *out_flags);
    }
  }
  return MI_OK;
}

int mi_get_option(const mi_gallery* g, const char* name, double* out_value) {
  REQUIRE(g && name && out_value, "null");
  const std::string n(name);
  if (n == "chunk0_tiles") *out_value = g->chunk0_tiles;
  else if (n == "sample_rows") *out_value = (double)(bootstrap_tiles(g) * TILE);
  else if (n == "rescore_grid_x") *out_value = g->rescore_grid_x;
  else if (n == "workspace_slot") *out_value = g->ws_slot;
  else if (n == "spec_max_ratio") *out_value = g->spec_max_ratio;
  else if (n == "chunk_growth") *out_value = g->chunk_growth;
  else if (n == "survivor_cap") *out_value = g->surv_cap;
  else if (n == "rescore_cap") *out_value = g->rescore_cap;
  else if (n == "exact_fallback") *out_value = g->exact_fallback;
  else if (n == "force_exact") *out_value = g->force_exact;
  else if (n == "speculative") *out_value = g->speculative;
  else if (n == "device_repair") *out_value = g->device_repair;
  else if (n == "small_batch_kernel") *out_value = g->small_batch_kernel;
  else if (n == "xcc_balance") *out_value = g->xcc_balance;
  else if (n == "ladder") *out_value = g->ladder;
  else if (n == "boot_ksplit") *out_value = g->boot_ksplit;
  else if (n == "stream_tail") *out_value = g->stream_tail;
  else if (n == "async_tail") *out_value = g->async_tail;
  else if (n == "query_norm_override") *out_value = g->qnorm_override;
  else if (n == "image_dtype") *out_value = g->img_f16;
  else return fail(MI_ERR_INVALID, "unknown option: " + n);
  return MI_OK;
}

int mi_set_option(mi_gallery* g, const char* name, double value) {
  REQUIRE(g && name, "null");
  const std::string n(name);
  if (g->pending.valid) {
    // a deferred tail (async_tail 3) is enqueued before ANY option changes: it must run with the buffers, caps and workspace
    // its batch was scored with
    HIPC(hipSetDevice(g->device));
    const int rc = flush_pending_tail(g, nullptr, false);
    if (rc != MI_OK) return rc;
  }
  if (n == "chunk0_tiles") { REQUIRE(value >= 0, "chunk0_tiles >= 0 (0 = default)"); g->chunk0_tiles = (int)value; }
  else if (n == "spec_max_ratio") { REQUIRE(value >= 1 && value <= 4096, "spec_max_ratio in [1, 4096]"); g->spec_max_ratio = (int)value; }
  else if (n == "workspace_slot") {
    REQUIRE(value == 0 || value == 1, "workspace_slot: 0 or 1");
    if ((int)value != g->ws_slot) {
      std::swap(g->ws, g->ws_alt);
      g->ws_slot = (int)value;
    }
  }
  else if (n == "rescore_grid_x") { REQUIRE(value >= 0 && value <= 4096, "rescore_grid_x in [0, 4096]"); g->rescore_grid_x = (int)value; }
  else if (n == "chunk_growth") { REQUIRE(value >= 1, "chunk_growth >= 1"); g->chunk_growth = (int)value; }
  else if (n == "survivor_cap") {
    const uint32_t v = (uint32_t)value;
    REQUIRE(v >= 1024 && v <= 16384 && (v % 256) == 0, "survivor_cap: multiple of 256 in [1024, 16384]");
    g->surv_cap = v;
  } else if (n == "rescore_cap") {
    const uint32_t v = (uint32_t)value;
    REQUIRE(v >= 64 && v <= 8192 && (v & (v - 1)) == 0, "rescore_cap: power of two in [64, 8192]");
    g->rescore_cap = v;
  } else if (n == "exact_fallback") g->exact_fallback = value != 0;
  else if (n == "force_exact") g->force_exact = value != 0;
  else if (n == "speculative") g->speculative = value != 0;
  else if (n == "device_repair") g->device_repair = value < 0 ? -1 : (value != 0);
  else if (n == "small_batch_kernel") g->small_batch_kernel = value != 0;
  else if (n == "xcc_balance") g->xcc_balance = value != 0;
  else if (n == "ladder") {
    REQUIRE(value == 0 || value == 1, "ladder: 0 (off) or 1 (on)");
    g->ladder = (int)value;
  }
  else if (n == "boot_ksplit") g->boot_ksplit = value != 0;
  else if (n == "stream_tail") g->stream_tail = value != 0;
  else if (n == "async_tail") {
    REQUIRE(value == 0 || value == 1 || value == 2 || value == 3, "async_tail: 0, 1, 2 or 3");
    g->async_tail = (int)value;
  }
  else if (n == "query_norm_override") {
    REQUIRE(value >= -1 && value <= 2, "query_norm_override: -1 or an mi_norm value");
    g->qnorm_override = (int)value;
  }
  else return fail(MI_ERR_INVALID, "unknown option: " + n);
  return MI_OK;
}

int64_t mi_debug_sample_source_row(int64_t i, int64_t n, int64_t n_s) {
  if (i < 0 || n_s <= 0 || n < n_s || i >= n_s) return -1;
  return sample_source_row_host(i, n, n_s);
}

int mi_debug_xcc_shares(mi_gallery* g, float* out_w8, int32_t* out_launches) {
  REQUIRE(g && out_w8, "null");
  HIPC(hipSetDevice(g->device));
  HIPC(hipDeviceSynchronize());
  const Workspace& sw = g->ws.bal ? g->ws : g->ws_alt;
  if (sw.bal) {
    XccBalance hb;
    HIPC(hipMemcpy(&hb, sw.bal, sizeof hb, hipMemcpyDeviceToHost));
    memcpy(out_w8, hb.w, sizeof hb.w);
    if (out_launches) *out_launches = (int32_t)hb.launches;
  } else {
    // no workspace yet: what the first one will start from (the file's shares, else an even split: -1 launches)
    for (int x = 0; x < 8; ++x) out_w8[x] = g->file_w_valid ? g->file_w[x] : 0.125f;
    if (out_launches) *out_launches = -1;
  }
  return MI_OK;
}

int mi_debug_read_cycles(mi_gallery* g, uint64_t* out_host, int64_t count) {
  REQUIRE(g && out_host, "null");
  REQUIRE(g->ws.dbg && count <= (int64_t)g->ws.nseg * 8, "no diagnostics buffer");
  HIPC(hipSetDevice(g->device));
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpy(out_host, g->ws.dbg, (size_t)count * 8, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_set_global_option(const char* name, double value) {
  REQUIRE(name, "null");
  const std::string n(name);
  if (n == "image_dtype") g_default_img_f16 = value != 0;
  else if (n == "host_ingest") {
    REQUIRE(value == 0 || value == 1, "host_ingest: 0 (one copy of the whole array) or 1 (row blocks)");
    g_host_ingest = (int)value;
  }
  else if (n == "keep_buffers") {
    REQUIRE(value == 0 || value == 1, "keep_buffers: 0 or 1");
    g_keep_buffers = (int)value;
    if (value == 0) {
      std::lock_guard<std::mutex> lock(g_spare_mu);
      spare_release_locked();
      spare_ws_release_locked();
    }
  }
  else return fail(MI_ERR_INVALID, "unknown global option: " + n);
  return MI_OK;
}

int mi_synth_fill_device(float* dst_dev, uint64_t seed, int64_t row0, int64_t nrows, int32_t d, void* stream) {
  REQUIRE(dst_dev, "null pointer");
  launch_synth_fill(dst_dev, seed, row0, nrows, d, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

}  // extern "C"
