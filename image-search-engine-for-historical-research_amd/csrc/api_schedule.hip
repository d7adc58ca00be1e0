// The search: phase 1 (plan, pre, main: sample schedule, chunk schedule, repair pass), phase 2 (exact re-score, emit), the
// asynchronous tails, the verified host loop with its fallbacks, the dense paths.  No torch, no numpy: plain pointers.
#include "api_internal.h"

QueryState make_state(const Workspace& ws) {
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
void prof_collect(mi_gallery* g) {
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
int64_t bootstrap_tiles(const mi_gallery* g) { return g->chunk0_tiles > 0 ? g->chunk0_tiles : 32; }

constexpr int64_t BIG_SAMPLE_TILES = 96;      // threshold sample of shards too large for the 32-tile one (plan_phase1)
constexpr int64_t HUGE_SAMPLE_TILES = 256;    // ... and for the 96-tile one: 65 536 rows, scores in a buffer of the handle's own

// the sample as f32 rows (f32 scorer on the sample schedule)
static int ensure_sample_f32(mi_gallery* g, int64_t tiles, hipStream_t s) {
  if (g->samp_f32 && g->samp_f32_tiles == tiles && g->samp_f32_for_n == g->n) return MI_OK;
  if (!g->samp_f32 || g->samp_f32_tiles != tiles) {
    (void)hipFree(g->samp_f32);
    g->samp_f32 = nullptr;
    HIPC(device_malloc((void**)&g->samp_f32, (size_t)tiles * TILE * g->dp * sizeof(float)));
    g->samp_f32_tiles = tiles;
  }
  launch_build_sample_f32(g->gal_f32, g->samp_f32, g->n, tiles * TILE, g->dp, s);
  HIPC(hipGetLastError());
  g->samp_f32_for_n = g->n;
  return MI_OK;
}

// (re)build the bootstrap sample image for the current number of rows
static int ensure_sample(mi_gallery* g, int64_t tiles, hipStream_t s) {
  if (g->samp_img && g->samp_tiles == tiles && g->samp_for_n == g->n && (tiles <= BIG_SAMPLE_TILES || g->samp_scores_tiles == tiles))
    return MI_OK;
  if (!g->samp_img || g->samp_tiles != tiles) {
    (void)hipFree(g->samp_img);
    g->samp_img = nullptr;
    HIPC(device_malloc(&g->samp_img, (size_t)tiles * TILE * g->dp * 2 + 256));
    g->samp_tiles = tiles;
  }
  if (tiles > BIG_SAMPLE_TILES && g->samp_scores_tiles != tiles) {      // the sample's scores do not fit the survivor rows
    (void)hipFree(g->samp_scores);
    g->samp_scores = nullptr;
    g->samp_scores_tiles = 0;
    HIPC(device_malloc((void**)&g->samp_scores, (size_t)QB * tiles * TILE * sizeof(float)));
    g->samp_scores_tiles = tiles;
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
  // The f32 scorer (exact: `force_exact`, or the re-run of a batch whose buffers overflowed) takes its thresholds from the
  // hashed sample too since round 6: its chunk schedule bootstrapped from the FIRST rows of the shard, which on a gallery
  // stored cluster by cluster let the survivor buffers overflow.  Its sample is the 8192-row one as stored f32
  // rows, scored by the f32 kernel itself -- no 16-bit margin enters (on raw galleries whose row norms differ 60 x the
  // 16-bit margin is what sent the batch here) --, admitted up to 3 x spec_max_ratio rows per sample row (3.93 M rows).
  // Single-launch schedule?  The speculative threshold is an order statistic of the scores of a SAMPLE: t0 * 256
  // rows drawn evenly (one hashed draw per stratum) into their own small image, so that the order in which the shard
  // was ingested cannot bias it.  With n_s sampled rows the shard's K-th largest score sits near sample rank
  // lambda = K * n_s / N; the r-th largest sample score with r = spec_rank(lambda) lies below it except with
  // probability 1e-7 per query (Poisson tail) and keeps the expected survivors at r * N / n_s.  The sample entries
  // are dropped once the threshold is taken (the scoring launch visits every tile, sample rows included).
  if (g->speculative && pl.ntiles >= 2 * t0 && g->n / (t0 * TILE) <= (exact ? 3 : 1) * (int64_t)g->spec_max_ratio) {
    const double lambda = (double)k * (double)(t0 * TILE) / (double)g->n;
    const int32_t r = spec_rank(lambda);
    if (r < k) pl.samp_r = r;
  } else if (g->speculative && !exact && g->chunk0_tiles <= 0 && g->small_batch_kernel &&
             (int64_t)2 * ws.cap >= BIG_SAMPLE_TILES * TILE && pl.ntiles >= 2 * BIG_SAMPLE_TILES &&
             g->n / (BIG_SAMPLE_TILES * TILE) <= g->spec_max_ratio) {
    // Round 6: shards beyond spec_max_ratio x 8192 = 1.31 M rows used to take the chunk schedule, whose speculative last launch
    // reads "the rows seen so far" -- the FIRST rows of the shard -- as its sample: on a gallery stored cluster by cluster that
    // sample is a handful of clusters (bench.py `hard_data`).  Up to spec_max_ratio x 24 576 = 3.93 M rows
    // they get a hashed sample of their own now: 96 tiles, whose scores fit a survivor row as 4-byte floats (2 x
    // survivor_cap) -- the same rows per sample row, so the same survivors per query, the single filtered launch, the ladder
    // and the repair pass of every smaller shard.  (At 2.6 x that ratio -- the 10 M-row gallery on ONE GPU with its bf16
    // margin -- some queries kept more than survivor_cap rows: measured, profiles/r06l_big_sample_10m.txt.)  Larger shards
    // still take the chunk schedule: the 8-way split of BASELINE configs[3] has 1.25 M rows per GPU and is on the sample path.
    const double lambda = (double)k * (double)(BIG_SAMPLE_TILES * TILE) / (double)g->n;
    const int32_t r = spec_rank(lambda);
    if (r < k && sample_threshold_applies((uint32_t)(BIG_SAMPLE_TILES * TILE), k, r)) {
      pl.samp_r = r;
      t0 = BIG_SAMPLE_TILES;
      pl.t0 = t0;
    }
  }
  else if (g->speculative && !exact && g->chunk0_tiles <= 0 && g->small_batch_kernel && pl.ntiles >= 2 * HUGE_SAMPLE_TILES &&
           g->n / (HUGE_SAMPLE_TILES * TILE) <= g->spec_max_ratio) {
    // ... and up to spec_max_ratio x 65 536 = 10.5 M rows (BASELINE configs[3] on ONE GPU: 10 M rows) a 256-tile sample whose
    // scores live in a buffer of the handle's own (256 MB, allocated with the sample image) and are ranked by
    // sample_threshold_big_kernel.  Beyond that: the chunk schedule.
    const double lambda = (double)k * (double)(HUGE_SAMPLE_TILES * TILE) / (double)g->n;
    const int32_t r = spec_rank(lambda);
    if (r < k && 4 * r <= 256) {
      pl.samp_r = r;
      pl.samp_ext = true;
      t0 = HUGE_SAMPLE_TILES;
      pl.t0 = t0;
    }
  }
  pl.first_cnt = (uint32_t)(pl.samp_r > 0 ? t0 * TILE : std::min<int64_t>(g->n, t0 * TILE));
  pl.gamma = 2.0f * (float)g->dp * 5.9604645e-08f;  // 2 * dp * 2^-24 (f32 accumulation, doubled)
  pl.thr_kernel = pl.samp_r > 0 && !exact && (pl.samp_ext || sample_threshold_applies(pl.first_cnt, k, pl.samp_r));
  // small batches on the sample schedule: the bootstrap launch splits K over several workgroups per (sample tile, query
  // group) and adds its partial scores onto zeros that the query ingest writes (ScoreArgs::ksplit)
  if (pl.samp_r > 0 && g->boot_ksplit && g->small_batch_kernel && pl.thr_kernel && g->dp <= 4096) {
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
  if (g->ladder && !exact && pl.samp_r > 1 && pl.thr_kernel) {
    const double lambda = (double)k * (double)(t0 * TILE) / (double)g->n;
    int32_t lr = (int32_t)std::lround(std::sqrt(lambda * pl.samp_r));
    pl.lad_r = std::max<int32_t>(1, std::min<int32_t>(lr, pl.samp_r - 1));
  }
  return pl;
}

// rows of the threshold sample a search of this handle would score (K = 100, the handle's options as they are)
int64_t sample_rows_in_effect(const mi_gallery* g) {
  Workspace w;
  w.cap = g->surv_cap;
  return plan_phase1(g, w, 1024, (int32_t)std::min<int64_t>(100, std::max<int64_t>(1, g->n)), false).t0 * TILE;
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
    const int rc = pl.exact ? ensure_sample_f32(g, pl.t0, s) : ensure_sample(g, pl.t0, s);
    if (rc != MI_OK) return rc;
  }
  int32_t boot_ksplit = pl.boot_ksplit;
  const int use_img = pl.exact ? 0 : 1;
  // query ingest (normalise, f32 rows, 16-bit image, rounding norms) and the per-query search state in one launch
  if (!launch_ingest_queries(q_src, q_dtype, pl.nq, g->d, q_rs, q_cs, q_norm, ws.q_f32, ws.q_img, g->img_f16, ws.q_stat, g->dp,
                             pl.qpad, g->gstat3, pl.gamma, use_img, pl.first_cnt, st, s,
                             boot_ksplit > 1 ? pl.first_cnt : 0u)) {
    boot_ksplit = 1;
    launch_ingest(q_src, q_dtype, pl.nq, g->d, q_rs, q_cs, q_norm, ws.q_f32, ws.q_img, g->img_f16, ws.q_stat, g->dp, pl.qpad, s);
    launch_init_query_state(ws.q_stat, g->gstat3, pl.nq, pl.qpad, pl.gamma, use_img, pl.first_cnt, st, s);
  }
  if (pl.samp_r > 0) {
    P1Plan p2 = pl;
    p2.boot_ksplit = boot_ksplit;
    if (pl.exact) {                                    // f32 scorer: the sample's f32 rows, every exact score stored
      ExactArgs a;
      a.gal_f32 = g->samp_f32;
      a.qry_f32 = ws.q_f32;
      a.dp = g->dp;
      a.row0 = 0;
      a.row1 = pl.t0 * TILE;
      a.n = pl.t0 * TILE;
      a.nq = pl.nq;
      a.st = st;
      launch_exact_select(a, true, s);
    } else
    p1_score_launch(g, ws, st, p2, 0, pl.t0, true, nullptr, false, true, false, s);     // bootstrap on the sample image
    if (pl.thr_kernel && pl.samp_ext)
      launch_sample_threshold_big(st, g->samp_scores, (uint32_t)(pl.t0 * TILE), pl.first_cnt, pl.nq, pl.k, pl.samp_r, pl.lad_r,
                                  0.f, s);
    else if (pl.thr_kernel)
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
  if (a.scores_only && pl.samp_ext) {
    a.samp_out = g->samp_scores;
    a.samp_ld = (uint32_t)(pl.t0 * TILE);
  }
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
    // (the f32 scorer has no conditional launch: a failed threshold is final there, FLAG_SPEC_FAIL, and the caller goes on to
    // the dense path)
    const bool repair_pass = !exact && (g->device_repair < 0 ? !small : g->device_repair != 0);
    const int rep_mode = repair_pass ? 0 : ((small && !caller_checks_flags && g->device_repair < 0 && !exact) ? 3 : 2);
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

int phase1_batch(mi_gallery* g, const void* q_src, int q_dtype, int64_t q_rs, int64_t q_cs, int q_norm,
                        int32_t nq, int32_t k, bool exact, hipStream_t s, bool fuse_cand,
                        bool caller_checks_flags) {
  Workspace& ws = g->ws;
  const P1Plan pl = plan_phase1(g, ws, nq, k, exact);
  const int rc = p1_pre(g, ws, pl, q_src, q_dtype, q_rs, q_cs, q_norm, s);
  if (rc != MI_OK) return rc;
  return p1_main(g, ws, pl, s, fuse_cand, caller_checks_flags);
}

// ---- phase 2 for one batch: candidates within the margin of L, exact f64 re-score, sorted emit --------
int phase2_batch(mi_gallery* g, int32_t nq, int32_t k, const float* L_dev, int64_t* out_idx, float* out_score,
                        double* out_score64, hipStream_t s, bool have_cand, bool resident,
                        Workspace* wsp) {
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
int flush_pending_tail(mi_gallery* g, hipStream_t s, bool beside_scoring) {
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
void count_flagged_batch(mi_gallery* g, uint32_t flags) {
  if (flags & ~(uint32_t)FLAG_SPEC_FAIL) g->stats.overflow_batches += 1;
  else g->stats.spec_retries += 1;
}

int check_k(const mi_gallery* g, int32_t k) {
  REQUIRE(k >= 1, "k must be >= 1");
  if ((int64_t)k > g->n)
    return fail(MI_ERR_INVALID, "k > number of gallery rows (the reference fails in numpy broadcasting here, "
                                "src/utils/nnsearch.py:703)");
  if ((uint32_t)k * 4 > g->surv_cap || (uint32_t)k > g->rescore_cap)
    return fail(MI_ERR_UNSUPPORTED, "k too large for the top-K path (needs k <= survivor_cap/4 and k <= rescore_cap)");
  return MI_OK;
}

// full search of up to any nq on device inputs (strided, any dtype), outputs on device
int search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t q_rs, int64_t q_cs, int q_norm,
                         int64_t nq, int32_t k, int64_t* out_idx, float* out_score, double* out_score64, bool exact,
                         hipStream_t s, bool allow_async, bool caller_checks_flags) {
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
int join_tails(mi_gallery* g, hipStream_t s) {
  if (g->pending.valid) {
    const int rc = flush_pending_tail(g, s, false);
    if (rc != MI_OK) return rc;
  }
  for (int i = 0; i < 2; ++i)
    if (g->ev_tail_valid[i]) HIPC(hipStreamWaitEvent(s, g->ev_tail[i], 0));
  return MI_OK;
}

int strided_extent(int64_t n, int64_t d, int64_t rs, int64_t cs, int64_t* elems) {
  REQUIRE(rs >= 0 && cs >= 0, "negative strides are not supported");
  *elems = (n > 0 && d > 0) ? (n - 1) * rs + (d - 1) * cs + 1 : 0;
  return MI_OK;
}

// ---- search ----------------------------------------------------------------------------------------
int read_and_clear_flags(mi_gallery* g, uint32_t* flags) {
  HIPC(hipMemcpy(flags, g->ws.flags, 4, hipMemcpyDeviceToHost));
  if (*flags) HIPC(hipMemset(g->ws.flags, 0, 4));
  return MI_OK;
}


// synchronous search with overflow handling: 16-bit MFMA pass, then (if buffers overflowed) the f32-scored filter pass,
// then (massive ties: more rows within the margin of the K-th score than the buffers hold) the dense f64 path: every
// score at full precision, exact top-k of the dense matrix -- the same contract as the filtered paths, at any data
int search_sync(mi_gallery* g, const void* q_dev, int q_dtype, int64_t rs, int64_t cs, int q_norm, int64_t nq,
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

// ---- dense exact kNN (k a large fraction of N) and truncated graph diffusion ---------------------------------------
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

// the same at full precision (search_sync's last resort): dense f64 scores of a sub-batch of queries (<= 2 GiB of scores
// at a time), exact top-k by (f64 score desc, idx asc)
int dense64_search_device(mi_gallery* g, const void* q_src, int q_dtype, int64_t rs, int64_t cs, int q_norm,
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
