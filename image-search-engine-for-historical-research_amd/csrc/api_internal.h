// Internals shared by the translation units of the C ABI (api_*.hip): the handle, its workspace, the process-wide state and
// the helpers more than one of them calls.  Nothing here is part of the public interface (include/mi355_retrieval.h).
//   api_state.hip     process-wide state (spare-buffer slots, XCD-share cache, global options), workspace allocation
//   api_schedule.hip  the search itself: phase 1 (plan / pre / main: sample schedule, chunk schedule, repair), phase 2, the
//                     asynchronous tails, the verified host loop with its fallbacks, the dense paths
//   api_gallery.hip   gallery life cycle: create / append / destroy, ingest of any layout, norm bounds, image type
//   api_file.hip      the prepared-gallery file (MI355GAL v2): save / load, parallel copies, checksums
//   api_entry.hip     search entry points: kNN (host / device / phases / merge), alpha-QE, dense, full-length ranking
//   api_aux.hip       descriptor tail, whitening, k-reciprocal re-ranking, diffusion, column sums, synthetic rows
//   api_options.hip   per-handle options, statistics, profiling, flags, diagnostics
#pragma once
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

#define MI_INTERNAL __attribute__((visibility("hidden")))

// ---- process-wide state (defined in api_state.hip; the comments on what each is for are there)
extern MI_INTERNAL std::atomic<int> g_default_img_f16;
extern MI_INTERNAL std::atomic<int> g_host_ingest;
extern MI_INTERNAL std::mutex g_bal_mu;
extern MI_INTERNAL std::map<int, std::vector<float>> g_bal_cache;
struct SpareBuffers {
  int device = -1;
  size_t f32_bytes = 0, img_bytes = 0, stat_bytes = 0;
  float* gal_f32 = nullptr;
  void* gal_img = nullptr;
  RowStat* rowstat = nullptr;
  float* gstat3 = nullptr;
};
struct SpareArena {           // the search workspace of the handle destroyed last (one carved allocation, ~200 MB)
  int device = -1;
  size_t bytes = 0;
  void* p = nullptr;
};
extern MI_INTERNAL std::mutex g_spare_mu;
extern MI_INTERNAL SpareBuffers g_spare;
extern MI_INTERNAL SpareArena g_spare_ws;
extern MI_INTERNAL std::atomic<int> g_keep_buffers;
constexpr size_t SPARE_MAX_BYTES = (size_t)16 << 30;
MI_INTERNAL void spare_release_locked();
MI_INTERNAL void spare_ws_release_locked();
MI_INTERNAL hipError_t device_malloc(void** p, size_t bytes);      // hipMalloc; out of memory: spares released, one more try
MI_INTERNAL int fail(int code, const std::string& msg);            // records the calling thread's error message, returns code
MI_INTERNAL const char* last_error_message();
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
    if (device_malloc(&p, count * sizeof(T) + 256) != hipSuccess) return nullptr;
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
  bool samp_ext = false;           // the sample's scores go to the handle's own buffer (mi_gallery::samp_scores), not into survivor rows
};

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
  float* samp_f32 = nullptr;       // the 8192-row sample as stored f32 rows (built when the f32 scorer first needs its thresholds)
  int64_t samp_f32_for_n = -1, samp_f32_tiles = 0;
  float* samp_scores = nullptr;    // [QB][samp_tiles * 256] scores of a sample too large for a survivor row (shards beyond 3.9 M rows)
  int64_t samp_scores_tiles = 0;
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
  std::atomic<int> online_users{0};   // mi_online handles built on this gallery: it cannot be destroyed under them
};

// ---- api_state.hip
MI_INTERNAL int ws_free(Workspace& ws);
MI_INTERNAL int ws_ensure(mi_gallery* g, int32_t k);
MI_INTERNAL bool snapshot_balance(const mi_gallery* g, float* out_w8);
// ---- api_schedule.hip
MI_INTERNAL QueryState make_state(const Workspace& ws);
MI_INTERNAL int64_t bootstrap_tiles(const mi_gallery* g);      // rows / 256 of the bootstrap chunk / the default threshold sample
MI_INTERNAL int64_t sample_rows_in_effect(const mi_gallery* g);
MI_INTERNAL void prof_collect(mi_gallery* g);
MI_INTERNAL int flush_pending_tail(mi_gallery* g, hipStream_t s, bool beside_scoring);
MI_INTERNAL int phase1_batch(mi_gallery* g, const void* q_src, int q_dtype, int64_t q_rs, int64_t q_cs, int q_norm, int32_t nq,
                             int32_t k, bool exact, hipStream_t s, bool fuse_cand = false, bool caller_checks_flags = false);
MI_INTERNAL int phase2_batch(mi_gallery* g, int32_t nq, int32_t k, const float* L_dev, int64_t* out_idx, float* out_score,
                             double* out_score64, hipStream_t s, bool have_cand = false, bool resident = false,
                             Workspace* wsp = nullptr);
MI_INTERNAL void count_flagged_batch(mi_gallery* g, uint32_t flags);
MI_INTERNAL int check_k(const mi_gallery* g, int32_t k);
MI_INTERNAL int search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t q_rs, int64_t q_cs, int q_norm, int64_t nq,
                              int32_t k, int64_t* out_idx, float* out_score, double* out_score64, bool exact, hipStream_t s,
                              bool allow_async = false, bool caller_checks_flags = false);
MI_INTERNAL int join_tails(mi_gallery* g, hipStream_t s);
MI_INTERNAL int strided_extent(int64_t n, int64_t d, int64_t rs, int64_t cs, int64_t* elems);
MI_INTERNAL int read_and_clear_flags(mi_gallery* g, uint32_t* flags);
MI_INTERNAL int search_sync(mi_gallery* g, const void* q_dev, int q_dtype, int64_t rs, int64_t cs, int q_norm, int64_t nq,
                            int32_t k, int64_t* idx_dev, float* score_dev, double* score64_dev);
MI_INTERNAL int dense_search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t rs, int64_t cs, int q_norm, int64_t nq,
                                    int32_t k, int64_t* out_idx_dev, float* out_score_dev, hipStream_t s);
MI_INTERNAL int dense64_search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t rs, int64_t cs, int q_norm,
                                      int64_t nq, int32_t k, int64_t* out_idx_dev, float* out_score_dev,
                                      double* out_score64_dev, hipStream_t s);
// ---- api_gallery.hip
MI_INTERNAL int gallery_alloc(mi_gallery* g);
