// Search entry points of the C ABI: exhaustive kNN (host / device / phases / merge), alpha-QE, dense kNN, full-length ranking.
#include "api_internal.h"

extern "C" {

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
      if (device_malloc(&g->io_buf[slot], want) != hipSuccess) return nullptr;
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
    if (device_malloc(&p, bytes + 256) != hipSuccess) return nullptr;
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

}  // extern "C"
