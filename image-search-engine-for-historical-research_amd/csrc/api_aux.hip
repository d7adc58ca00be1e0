// Operators beside the search: descriptor tail of the extractor, whitening, k-reciprocal re-ranking, graph diffusion,
// column sums, synthetic rows.
#include "api_internal.h"

extern "C" {

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
  HIPC(device_malloc((void**)&g->dif_ids, (size_t)n * T * 4));
  HIPC(device_malloc((void**)&g->dif_vals, (size_t)n * T * 4));
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
  HIPC(device_malloc((void**)&g->dif_ids, cnt * 4));
  HIPC(device_malloc((void**)&g->dif_vals, cnt * 4));
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

int mi_synth_fill_device(float* dst_dev, uint64_t seed, int64_t row0, int64_t nrows, int32_t d, void* stream) {
  REQUIRE(dst_dev, "null pointer");
  launch_synth_fill(dst_dev, seed, row0, nrows, d, (hipStream_t)stream);
  HIPC(hipGetLastError());
  return MI_OK;
}

}  // extern "C"
