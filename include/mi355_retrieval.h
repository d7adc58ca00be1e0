/*
 * mi355_retrieval.h -- C ABI of libmi355_retrieval.so, the MI355X (gfx950) retrieval hot path.
 *
 * Plain pointers and sizes only (no torch / numpy types).  Each entry point names the reference
 * interface it replaces (paths relative to the reference tree).  The reference has no FFI: its
 * "plugin surface" is the --matching_method if/elif chain calling free functions of
 * src/utils/nnsearch.py (src/offline.py:107-118, src/online.py:132-143), so the binding a
 * maintainer adds is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - status codes (MI_OK == 0); mi_last_error() returns the message of the calling thread's last
 *     failure.  Nothing throws across the boundary.
 *   - a gallery handle is ONE row shard on ONE device (one process per GPU; row_offset globalises
 *     the indices it returns).  Inputs are never modified; outputs are caller-owned.
 *   - "host" entry points are synchronous and take strided f32/f64 host arrays (callers of the
 *     reference pass `.T` views of [D,N] arrays, so strides are part of the contract).
 *   - "_device" entry points take device pointers, enqueue on `stream` (a hipStream_t) and return
 *     without synchronising; sticky error flags are read with mi_search_status().
 *   - a handle is not re-entrant: serialise calls on one handle (online.py's Flask threads must
 *     hold a lock; the Python wrapper does).
 */
#ifndef MI355_RETRIEVAL_H
#define MI355_RETRIEVAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_gallery mi_gallery; /* opaque */

enum mi_status {
  MI_OK = 0,
  MI_ERR_INVALID = 1,     /* bad argument (K > N like the reference's broadcast error, null ptr, ...) */
  MI_ERR_HIP = 2,         /* HIP runtime failure; message carries hipGetErrorString */
  MI_ERR_NOMEM = 3,
  MI_ERR_IO = 4,
  MI_ERR_OVERFLOW = 5,    /* candidate buffers overflowed and the exact fallback was disabled */
  MI_ERR_UNSUPPORTED = 6
};
enum mi_dtype { MI_F32 = 0, MI_F64 = 1 };
enum mi_memspace { MI_HOST = 0, MI_DEVICE = 1 };
enum mi_norm {
  MI_NORM_NONE = 0,  /* rows used as given: inner-product ranker, src/main_retrieve.py:175-176; faiss
                        IndexFlatIP, src/utils/knn.py:33-40; QGE re-score, src/utils/Reranking.py:206 */
  MI_NORM_L2 = 1,    /* x / ||x||, no eps: matching_L2, src/utils/nnsearch.py:693-698 */
  MI_NORM_L2_EPS = 2 /* x / (||x|| + 1e-6): l2n, src/layers/functional.py:129-130; whitenapply tail,
                        src/utils/whiten.py:10 */
};

const char* mi_last_error(void);
int mi_device_count(int* count);

/* ---- gallery ingest: replaces the per-call normalisation of matching_L2
 * (src/utils/nnsearch.py:693-698) and faiss index.add (src/utils/knn.py:17-23).
 * data: element (i,j) at data[i*row_stride + j*col_stride] (strides in elements), n rows, d cols.
 * Builds, resident in HBM: normalised f32 rows [n][d64], the tile-blocked bf16 copy the MFMA
 * kernel streams, and per-row rounding-error norms used for the exactness certificate.
 * Synchronous.  MI_DEVICE data is read on a stream of the handle's own (non-blocking, NOT ordered against any stream of the
 * caller): whatever produced `data` must have completed -- synchronise the producing stream first -- and the pointer may be
 * freed as soon as the call returns.  mi_gallery_append_device is the stream-ordered way in. */
int mi_gallery_create(const void* data, int64_t n, int32_t d, int dtype, int64_t row_stride,
                      int64_t col_stride, int memspace, int norm_mode, int device,
                      int64_t row_offset, mi_gallery** out);
/* Appendable gallery (offline pipeline): capacity rows are allocated, rows arrive from the device
 * (descriptors straight out of the extractor tail) and are normalised / imaged / measured in place.  Appends and
 * searches on one handle must be serialised by the caller; `stream` orders the append against the producer. */
int mi_gallery_create_empty(int64_t capacity, int32_t d, int norm_mode, int device, int64_t row_offset,
                            mi_gallery** out);
int mi_gallery_append_device(mi_gallery* g, const float* rows_dev /*[m][d] f32*/, int64_t m, void* stream);
/* Strided append of m rows (f32 | f64, host or device memory; synchronous, device rows must be complete as for create).  A host column block of a [D, N] array --
 * row_stride 1, col_stride N, the layout of the reference's feature pickles and of its 1M-distractor tensor
 * (src/utils/general.py:67-92, src/extract_1m.py:97-98, src/test_rOP1m.py:136-139) -- is packed by one 2-D copy and read
 * with strides on the device: no host transpose, no concatenated host copy, no float64 promotion. */
int mi_gallery_append(mi_gallery* g, const void* data, int64_t m, int dtype, int64_t row_stride, int64_t col_stride,
                      int memspace);
int mi_gallery_destroy(mi_gallery* g);
int mi_gallery_info(const mi_gallery* g, int64_t* n, int32_t* d, int32_t* norm_mode, int32_t* device,
                    int64_t* row_offset, int64_t* hbm_bytes);
/* Prepared-gallery persistence: the counterpart of the ANN methods' `outputs/<dataset>/` index files
 * guarded by `ifgenerate` (src/utils/nnsearch.py:503-525, 1033-1044). */
int mi_gallery_save(const mi_gallery* g, const char* path);
int mi_gallery_load(const char* path, int device, mi_gallery** out);
/* Copies normalised f32 rows [row0,row0+nrows) x d to a host buffer (tests / diffusion host logic). */
int mi_gallery_get_rows(const mi_gallery* g, int64_t row0, int64_t nrows, float* out_host);

/* ---- exhaustive kNN: replaces matching_L2(K, train, test) -> (idx, time_per_query)
 * (src/utils/nnsearch.py:687-706) and KNN.search (src/utils/knn.py:25-31).
 * q: nq x d strided host array.  out_idx [nq][k] int64 (row_offset + local row), out_score [nq][k]
 * f32 (may be NULL): exact inner product of the stored rows with the (normalised) query, descending,
 * ties to the lower index.  out_seconds (may be NULL): device-synchronised wall time of the call
 * including query upload/normalisation, excluding nothing. */
int mi_knn_search(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride,
                  int64_t col_stride, int32_t k, int64_t* out_idx, float* out_score,
                  double* out_seconds);

/* Device-resident variant: q_dev [nq][d] row-major f32 (C order), outputs are device buffers.
 * out_score64_dev (may be NULL) receives the float64 exact scores. */
int mi_knn_search_device(mi_gallery* g, const float* q_dev, int64_t nq, int32_t k,
                         int64_t* out_idx_dev, float* out_score_dev, double* out_score64_dev,
                         void* stream);

/* Asynchronous tail (mi_set_option "async_tail" = 1): mi_knn_search_device enqueues the scoring / filtering part of a batch
 * on `stream` and the exact re-score + sort on a stream of the handle, so that the tail of batch i runs beside the scoring
 * launch of batch i + 1 (the MFMA kernel leaves exactly the registers one re-score wave per SIMD needs and no LDS).  The
 * outputs of every call made so far are complete, in the order of `stream`, after mi_search_join(g, stream).  Default off:
 * outputs are then complete in stream order when mi_knn_search_device returns, as before.
 * "async_tail" = 3 (deferred): the tail of a batch of > 128 queries is not enqueued by its own call but by the NEXT
 * mi_knn_search_device call on the handle, after that batch's query ingest / bootstrap / threshold launches and right before
 * its scoring launch (or by mi_search_join, or by any call that cannot carry it on): the re-score gather then shares the
 * device with the power-bound scoring launch only.  The output buffers of a call must stay valid until the join. */
int mi_search_join(mi_gallery* g, void* stream);

/* Sharded search = phase 1 on every shard, all-gather of approx top-k values, phase 2, all-gather of
 * exact (score64, idx), merge.  New functionality (the reference is single-process).
 * phase 1: bf16 MFMA scoring + survivor filtering; writes the shard's k largest approximate scores
 *          (unsorted) to out_approx_dev [nq][k] (-inf padded when the shard has < k rows). */
int mi_knn_phase1_device(mi_gallery* g, const float* q_dev, int64_t nq, int32_t k,
                         float* out_approx_dev, void* stream);
/* K-th largest of the gathered [g][nq][k] approximate values -> lower bound L [nq]. */
int mi_kth_of_gathered_device(const float* gathered_dev, int32_t nshards, int64_t nq, int32_t k,
                              float* out_L_dev, void* stream);
/* phase 2: exact f64 re-score of every local row whose approximate score is within the rigorous error
 *          margin of L; emits the shard's exact top-k (score64 desc, idx asc), -inf/-1 padded. */
int mi_knn_phase2_device(mi_gallery* g, int64_t nq, int32_t k, const float* L_dev,
                         int64_t* out_idx_dev, float* out_score_dev, double* out_score64_dev,
                         void* stream);
/* merge of [nshards][nq][k] exact lists -> [nq][k] by (score64 desc, idx asc).  Every list must be what phase 2 emits:
 * sorted in that order, padded with -1 / -inf at the end; row ids of different shards are disjoint (row shards). */
int mi_topk_merge_device(const double* score64_dev, const int64_t* idx_dev, int32_t nshards,
                         int64_t nq, int32_t k, int64_t* out_idx_dev, float* out_score_dev,
                         void* stream);
/* Same merge for lists that arrive interleaved per shard: shard g's scores start at score64_dev + g * shard_stride and its
 * indices at idx_dev + g * shard_stride (elements of 8 bytes), e.g. one all-gather of a packed [2][nq][k] buffer per rank
 * (scores, then indices) instead of two collectives. */
int mi_topk_merge_strided_device(const double* score64_dev, const int64_t* idx_dev, int64_t shard_stride,
                                 int32_t nshards, int64_t nq, int32_t k, int64_t* out_idx_dev,
                                 float* out_score_dev, void* stream);

/* ---- alpha query expansion: replaces feature_enhancement (src/utils/Reranking.py:195-208, copy at
 * :288-301): q' = sum_j ((k-j)/k)^w * G[ranks[j,q]], q' /= (||q'|| + eps), then a full re-search.
 * ranks: element (j,q) at ranks[j*rank_stride_j + q*rank_stride_q], global row ids (int64).
 * partial: this shard's contribution sum (rows it owns) as f64 [nq][d]; finish: normalise the
 * (all-reduced) sum into f32 queries [nq][d]. */
int mi_aqe_partial_device(mi_gallery* g, const int64_t* ranks_dev, int64_t rank_stride_j,
                          int64_t rank_stride_q, int64_t nq, int32_t k_qe, double w,
                          double* out_sum_dev, void* stream);
/* Across shards (round 4; replaces the all-gather of f64 partial sums): `rows` writes, for every requested (j, q), the f32 row
 * this shard owns -- zeros when another shard owns it -- into out_rows_dev [k_qe][nq][d]; the blocks of all shards are SUMMED
 * (all-reduce: every element has exactly one non-zero contributor, so the sum is exact whatever order the collective adds
 * in; k_qe * nq * d * 4 bytes, 24 MiB at k_qe = 3, nq = 1024, d = 2048); `combine` adds the rows in j order with the
 * single-shard kernel's own weight and fused multiply-add -> the f64 sum [nq][d] of ONE gallery, bit for bit, independent of
 * the shard boundaries. */
int mi_aqe_rows_device(mi_gallery* g, const int64_t* ranks_dev, int64_t rank_stride_j, int64_t rank_stride_q, int64_t nq,
                       int32_t k_qe, float* out_rows_dev, void* stream);
int mi_aqe_combine_device(const float* rows_dev, int64_t nq, int32_t d, int32_t k_qe, double w, double* out_sum_dev,
                          void* stream);
int mi_aqe_finish_device(const double* sum_dev, int64_t nq, int32_t d, double eps, float* out_q_dev,
                         double* out_q64_dev, void* stream);
/* Host convenience (single shard): ranks host int64; out_qexp (may be NULL) f64 [nq][d]. */
int mi_aqe_search(mi_gallery* g, const int64_t* ranks, int64_t rank_stride_j, int64_t rank_stride_q,
                  int64_t nq, int32_t k_qe, double w, double eps, int32_t k, int64_t* out_idx,
                  float* out_score, double* out_qexp, double* out_seconds);

/* ---- dense exact kNN: faiss IndexFlatIP.search (src/utils/knn.py:25-31) when k is a large fraction of N (the kNN
 * graph of the diffusion, src/utils/diffusion.py:66).  Exact f32 inner products (k-ordered fmaf chain) of all
 * stored rows, top-k by (score desc, idx asc), k <= 4096. */
int mi_knn_dense_search(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride,
                        int64_t col_stride, int32_t k, int64_t* out_idx, float* out_score,
                        double* out_seconds);
/* The same at full precision and with no threshold logic at all: EVERY score of the gallery in float64 (f32 stored rows,
 * exact f64 products, f64 accumulation -- the arithmetic of the certificate's re-score) into a dense [queries, N] matrix,
 * then the exact top-k of it by (score desc, idx asc), k <= 4096.  The last resort of mi_knn_search on massively tied
 * data, and the INDEPENDENT checker of the filtered path at full size: whatever the filter, thresholds or candidate
 * buffers did, the two answers must agree (tests/test_gpu_full_size.py, bench.py `score_check`).  Replaces the full
 * argsort of src/utils/nnsearch.py:701-703 as the definition of "nothing is missing".  out_score / out_score64 may be NULL. */
int mi_knn_dense64_search(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride,
                          int64_t col_stride, int32_t k, int64_t* out_idx, float* out_score,
                          double* out_score64, double* out_seconds);

/* ---- full-length ranking: `np.argsort(-scores, axis=0)` over ALL rows (src/main_retrieve.py:176,
 * src/utils/Reranking.py:207; --mode mAP of src/test_rOP1m.py:144-149).  Exact f32 inner products, stable radix sort:
 * out_idx [nq][n] (score desc, idx asc, NaN last), out_score [nq][n] (may be NULL).  query_norm: -1 = normalise the
 * queries like the gallery, otherwise an mi_norm value (MI_NORM_NONE for an already expanded query). */
int mi_rank_all(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                int query_norm, int64_t* out_idx, float* out_score, double* out_seconds);
/* The first `keep` columns of that ranking only ([nq, keep] outputs): K beyond the top-K path's limit without [nq, N] host
 * arrays (matching_HIP with 2048 < K < N). */
int mi_rank_prefix(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                   int query_norm, int64_t keep, int64_t* out_idx, float* out_score, double* out_seconds);
/* Zero-based positions, in that same full ranking, of m listed rows per query (row_ids [nq, m], global ids, entries
 * outside the shard -> -1), computed by counting on the device: what compute_map2 needs of `ranks_aqe` [N, Q]
 * (src/utils/Reranking.py:280-283, src/utils/evaluate2.py:73-86) without the [N, Q] array.  m <= 2048. */
int mi_rank_positions(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride, int64_t col_stride,
                      int query_norm, const int64_t* row_ids, int32_t m, int64_t* out_pos);

/* ---- k-reciprocal re-ranking: kr_reranking(qvecs, vecs) of src/utils/Reranking.py:447-624 (k1 = 20, k2 = 6,
 * lambda = 0.3 there).  qvecs [nq, d], vecs [n, d] host arrays (element strides; the reference takes the [d, .] arrays and
 * transposes), rows assumed L2-normalised like the reference assumes.  out_idx [nq, n]: gallery indices by ascending final
 * distance (= the reference's returned `indices`); out_dist (may be NULL): those distances.  nq + n <= 32768. */
int mi_kr_rerank(const void* qvecs, int64_t nq, int64_t q_row_stride, int64_t q_col_stride, const void* vecs, int64_t n,
                 int64_t v_row_stride, int64_t v_col_stride, int32_t d, int dtype, int32_t k1, int32_t k2,
                 double lambda_value, int device, int64_t* out_idx, float* out_dist);

/* ---- truncated graph diffusion: Diffusion.get_offline_results (src/utils/diffusion.py:52-84 with :15-19, :87-116)
 * on a MI_NORM_NONE gallery of the features.  out_ids [n][n_trunc] (the kNN lists = columns of the sparse
 * `offline` matrix), out_vals [n][n_trunc] f32 (its values), out_knn_sims (may be NULL).  The result also stays
 * on the device for mi_diffusion_online. */
int mi_diffusion_offline(mi_gallery* g, int32_t n_trunc, int32_t kd, double alpha, int32_t gamma,
                         int32_t maxiter, double tol, int64_t* out_ids, float* out_vals, float* out_knn_sims);
/* The same for the nodes [node0, node1) only (the N truncated CG solves of src/utils/diffusion.py:15-19 are
 * independent, so the ranks of a multi-GPU run each take a node range of the replicated feature set; the k-NN graph and
 * the Laplacian are computed in full on every rank).  out_ids [n][n_trunc] as above; out_vals [node1 - node0][n_trunc] =
 * those rows of the offline matrix.  Gather the parts and install them with mi_diffusion_set_offline. */
int mi_diffusion_offline_nodes(mi_gallery* g, int32_t n_trunc, int32_t kd, double alpha, int32_t gamma,
                               int32_t maxiter, double tol, int64_t node0, int64_t node1, int64_t* out_ids,
                               float* out_vals, float* out_knn_sims);
/* Re-installs a cached offline result (the reference caches it as offline.jbl, src/utils/diffusion.py:21-40). */
int mi_diffusion_set_offline(mi_gallery* g, const int64_t* ids, const float* vals, int32_t n_trunc);
/* Online stage (src/utils/Reranking.py:238-253): top-k_query neighbours of each query, sims**gamma, weighted sum of
 * their offline rows, top-`trunc` ranks [nq][trunc] (score desc, idx asc) and scores. */
int mi_diffusion_online(mi_gallery* g, const void* q, int64_t nq, int dtype, int64_t row_stride,
                        int64_t col_stride, int32_t k_query, int32_t gamma, int32_t trunc,
                        int64_t* out_ranks, float* out_scores);

/* ---- descriptor tail of the extractor (src/networks/imageretrievalnet.py:183-187, 464-479; src/layers/functional.py:
 * 20-22, 129-130): GeM pooling of the last feature map feat[b][c][hw] (p, eps), L2N (eps 1e-6), optional whitening
 * Linear(c -> c_out, bias) + L2N (scratch_dev: [b][c] floats), into out_dev [b][c_out or c].  Multi-scale: accumulate
 * desc^msp per scale (first != 0 overwrites), then finish = (acc / nscales)^(1/msp) / ||.||. */
int mi_desc_tail_device(const float* feat_dev, int32_t b, int32_t c, int32_t hw, float p, float eps,
                        const float* whiten_w_dev, const float* whiten_b_dev, int32_t c_out, float* scratch_dev,
                        float* out_dev, void* stream);
int mi_desc_ms_accumulate_device(float* acc_dev, const float* desc_dev, int64_t count, float msp, int first,
                                 void* stream);
int mi_desc_ms_finish_device(float* acc_dev, int32_t b, int32_t d, int32_t nscales, float msp, void* stream);

/* ---- building blocks of average_query_expansion / database_augmentation (src/utils/Reranking.py:314-432):
 * out_sum[q][:] = sum_j weights[j] * row(ranks[j,q]) in float64 (means / logspace-weighted sums of neighbours), and
 * column sums of a strided matrix (the centring step). */
int mi_gather_weighted(mi_gallery* g, const int64_t* ranks, int64_t rank_stride_j, int64_t rank_stride_q, int64_t nq,
                       int32_t k, const double* weights, double* out_sum);
int mi_column_sum(const void* X, int64_t n, int32_t d, int dtype, int64_t row_stride, int64_t col_stride, int device,
                  double* out);

/* ---- whitenapply (src/utils/whiten.py:4-12): out[n][dims] = P[:dims] (x_n - m), rows divided by (||.|| + eps)
 * (eps < 0: no normalisation).  X: n images x d, strided (the reference's [D,N] array is passed with
 * row_stride 1, col_stride N); m f64 [d]; P f64 row-major [dims][d] (the first dims rows of the reference's P).
 * float64 arithmetic like the reference. */
int mi_whiten_apply(const void* X, int64_t n, int32_t d, int dtype, int64_t row_stride, int64_t col_stride,
                    const double* m, const double* P, int32_t dims, double eps, int device, double* out);

/* The same on device-resident operands, enqueued on `stream` without synchronising: X_dev strided f32 | f64, m_dev f64 [d],
 * P_dev f64 row-major [>= dims][d], out_dev f64 [n][dims] (un-normalised when eps < 0).  An f64 GEMM of 2 * dims * d flop per
 * image on v_mfma_f64_16x16x4_f64 (csrc/whiten.hip); the centring is applied while X is loaded. */
int mi_whiten_apply_device(const void* X_dev, int64_t n, int32_t d, int dtype, int64_t row_stride, int64_t col_stride,
                           const double* m_dev, const double* P_dev, int32_t dims, double eps, double* out_dev, void* stream);
/* Learned whitening straight into an appendable gallery (mi_gallery_create_empty with d = dims and MI_NORM_L2_EPS, whose
 * normalisation IS whitenapply's tail `X / (norm + 1e-6)`, src/utils/whiten.py:10): m device rows are whitened in chunks of
 * 32 768 rows into one float64 scratch block and ingested from there -- the [N, dims] float64 matrix that
 * src/main_train.py:711-712 holds never exists.  Rows of P applied = the gallery's dimension.  Synchronises `stream` before
 * it returns (the scratch block is freed); appends and searches on one handle must be serialised by the caller. */
int mi_gallery_append_whitened_device(mi_gallery* g, const void* X_dev, int64_t m, int32_t d, int dtype, int64_t row_stride,
                                      int64_t col_stride, const double* mean_dev, const double* P_dev, void* stream);

/* ---- agreement between the shards of one gallery (multi-GPU).  The exactness certificate keeps every row
 * whose approximate score is within 2*eps of the K-th largest approximate score L of the WHOLE gallery; eps is derived
 * from the norm maxima measured at ingest {max ||g||, max ||g_hat||, max ||g_hat - g||} and from the image element type.
 * A row on shard B is only guaranteed approx >= L - eps_A - eps_B when the rows at L sit on shard A, so every shard must
 * use the maxima over ALL shards and the same image type: all-reduce(MAX) the bounds, all-reduce(MIN) the type, write
 * them back (sharded.ShardedGallery does this on construction). */
int mi_gallery_norm_bounds(mi_gallery* g, float* bounds3 /* in-out */, int raise);  /* raise=0: read; 1: bounds = max(own, given) */
int mi_gallery_set_image_dtype(mi_gallery* g, int f16);  /* re-images the stored f32 rows (1 = fp16, 0 = bf16); no-op if equal */

/* The eight XCDs of one MI355X hold different clocks under the same load, so the tile kernel splits the gallery tiles over
 * them by their MEASURED speed (option "xcc_balance"); the shares start equal and converge over the first ~4 large launches
 * of a handle (finish times spread by 2-3 % until then).  mi_gallery_calibrate runs `launches` (<= 64; 8 is plenty) scoring
 * launches of up to 1024 of the gallery's own rows against the whole gallery on `stream`, asynchronously, and discards the
 * answers: the shares are converged before the first real search instead of during it.  No-op for galleries of < 512 tiles
 * (131 072 rows).  One-off cost: `launches` x one batch (27 ms for 8 launches at 1 M x 2048 rows).  Clears the sticky flags
 * (read them first if asynchronous searches are outstanding). */
int mi_gallery_calibrate(mi_gallery* g, int32_t launches, void* stream);

/* ---- status / instrumentation */
typedef struct mi_search_stats {
  int64_t searches;           /* query batches processed */
  int64_t queries;
  int64_t overflow_batches;   /* batches whose candidate / survivor / record buffers overflowed (or whose queries left fp16's range):
                               * answered again through the f32 scorer, then the dense f64 path */
  int64_t survivors;          /* sum over queries of entries kept by the filter */
  int64_t candidates;         /* sum over queries of rows re-scored exactly */
  double gemm_ms;             /* HIP-event time of the MFMA scoring launches (profiling on) */
  int64_t gemm_launches;
  double gemm_flops;          /* algorithmic 2*Q*N*D of those launches */
  double gemm_bytes;          /* algorithmic gallery + query bytes of those launches */
  double kernel_clock_mhz;    /* shader clock inside the most recent tile-kernel launch (s_memtime / s_memrealtime around its
                               * main loop, median over waves); 0 if that kernel has not run.  The chip lowers its clock under
                               * MFMA load, so this is what the dense peak scales with */
  int64_t spec_retries;       /* batches whose speculative threshold failed its verification (and, where a device repair pass ran,
                               * that too): answered again by the rigorous chunk schedule.  Not an overflow */
  int64_t inkernel_repairs;   /* queries of batches of <= 128 queries on the asynchronous (device / phase) entry points whose
                               * speculative threshold failed and that were repaired INSIDE the maintain launch: the query's workgroup
                               * rescans the whole shard (~0.1 s per 1 M rows x 2048, over 1 s on a 10 M-row shard; expected once per
                               * ~10^7 queries).  The answer is complete; this counter is the only trace of why that call was slow */
} mi_search_stats;
int mi_profile_enable(mi_gallery* g, int on);      /* times the scoring launches with HIP events (dispatch timestamps) */
int mi_search_status(mi_gallery* g, mi_search_stats* out, int reset); /* synchronises the handle's work */
/* The durations (ms, launch order) of the timed scoring launches since the last mi_search_status(reset = 1): what gemm_ms is
 * the sum of.  Waits for the launches enqueued so far; writes min(count, cap) values, *out_count = launches logged. */
int mi_profile_launch_ms(mi_gallery* g, float* out_host, int64_t cap, int64_t* out_count);
/* Tunables (19 names; everything that was an A/B switch of a measured-and-rejected variant -- "debug", "kernel_variant",
 * "small_tail", "stream_lookahead", "inkernel_repair_max", "ladder" = 2 -- left the product in round 5: MI_ERR_INVALID):
 * "chunk0_tiles" (rows / 256 of the bootstrap chunk and of the threshold sample; 0 = default 32), "chunk_growth",
 * "workspace_slot" (0 | 1: which of the handle's two per-batch workspaces the phase API uses -- phase 1 of batch i + 1 may
 * be enqueued before phase 2 of batch i; sticky flags and statistics are one set for both),
 * "spec_max_ratio" (largest shard rows / sample rows for which the single-launch sample schedule is taken; default 160),
 * "survivor_cap", "rescore_cap", "exact_fallback" (0 = report MI_ERR_OVERFLOW instead of falling back to the f32 scorer and
 * then the dense f64 path), "ladder" (in-launch threshold ladder of the tile kernel: 0 = off, 1 = on (default)),
 * "boot_ksplit" (batches of <= 512 queries: the bootstrap launch
 * on the sample splits K over several workgroups that add their partial scores with float atomics; default 1.  The order of
 * those adds is not fixed, so the sample scores -- and with them the survivor / candidate statistics and which queries need a
 * repair -- may differ by an ulp from run to run; the answers do not: the threshold is speculative and verified), "xcc_balance" (XCD shares by measured
 * speed), "stream_tail" (default 1: a HOST entry point called with more than 1024 queries runs its internal batches with the
 * deferred tail of "async_tail" 3 and reads the sticky flags once at the end; 0 = one verified batch after the other),
 * "async_tail" (1 | 2 | 3: re-score + sort on the handle's own stream beside the next batch's scoring launch | beside its
 * query ingest and bootstrap only | deferred: enqueued by the next call right before its scoring launch; see mi_search_join), "rescore_grid_x" (workgroups of 2
 * candidates per query in the re-score launch; 0 = 64; a shard of a G-way gallery sets ~96 / G),
 * "force_exact" (score with the f32 kernel instead of the 16-bit MFMA), "speculative" (0 = rigorous chunk schedule only),
 * "device_repair" (-1 = default: the scoring launch of a batch of > 128 queries is followed by a device-conditional repair
 * pass for queries whose speculative threshold failed verification; smaller batches launch none: through a HOST entry point
 * the sticky flag makes the call answer the batch again, through the asynchronous device / phase entry points the workgroup
 * of the failed query repairs it inside the maintain launch (a scan of the shard's rows, ~0.1 s per million -- over a second
 * on a 10 M-row shard --, once per 10^7 queries; counted in mi_search_stats.inkernel_repairs), so their answers are complete
 * without anybody reading flags; 0 / 1 = never / always launch the repair pass),
 * "small_batch_kernel" (0 = batches of <= 128 queries use the 256 x 256-tile kernel too),
 * "query_norm_override" (-1 | mi_norm: how the _device entry points normalise their queries; MI_NORM_NONE for the
 * already normalised expanded queries of alpha-QE).
 * mi_get_option also answers "image_dtype" (1 = fp16, 0 = bf16; read-only, see mi_gallery_set_image_dtype) and
 * "sample_rows" (rows of the threshold sample in effect). */
int mi_set_option(mi_gallery* g, const char* name, double value);
int mi_get_option(const mi_gallery* g, const char* name, double* out_value);   /* same names as mi_set_option */
/* Synchronises the handle's work, returns the sticky device flags raised by the asynchronous _device entry points since
 * the last call (0 = none; any bit = that batch must be answered again: buffer overflow or failed speculative threshold)
 * and clears them.  Unlike mi_search_status it leaves the statistics accumulators alone. */
int mi_search_flags(mi_gallery* g, uint32_t* out_flags);

/* Process-wide defaults for galleries created afterwards.  "image_dtype": element type of the 16-bit tile-blocked image the
 * MFMA kernel streams, 1 = fp16 (default: 2^-11 rounding, 8x tighter certificate than bf16 at the same MFMA rate;
 * un-normalised galleries whose rows exceed its comfortable range are stored as bf16 automatically), 0 = bf16.
 * "host_ingest": how mi_gallery_create moves a HOST array in one of the reference's two layouts to the device: 1 (default) = row
 * blocks of ~32 MiB copied by the runtime straight from the caller's pageable array into two alternating device blocks, the copy
 * of block i + 1 under the ingest of block i, no staging allocation the size of the gallery; 0 = one copy of the whole array into
 * a same-size staging buffer, then one ingest (rounds 1-4).  Both reach 0.96 of the pinned H2D rate (profiles/r05f_*).
 * "keep_buffers": 1 (default) = mi_gallery_destroy keeps the buffers of a gallery of up to 16 GiB and its search workspace (one
 * carved allocation of ~200 MB) in one spare slot each per process, and the next gallery of exactly the same sizes on the same
 * device takes them instead of allocating: a caller that prepares a gallery per call (create, search, destroy: a stateless
 * matching_<method>, src/utils/nnsearch.py:687-706) stops paying 1-6 ms of hipMalloc / hipFree per 12 GB and ~4 ms for the
 * workspace; 0 = free the spares now and keep nothing. */
int mi_set_global_option(const char* name, double value);
/* "release_spares" (any value) gives the spare slots back now and leaves "keep_buffers" as it is.  An allocation of the library
 * that fails with out-of-memory releases them by itself and is tried once more; a gallery of other sizes than the spare releases
 * it when the device could not hold both.  mi_get_global_option reads "image_dtype", "host_ingest", "keep_buffers" and
 * "spare_bytes": the device memory the process holds in the spare slots right now -- what a co-tenant of the GPU (the
 * extractor's PyTorch allocator) cannot see otherwise. */
int mi_get_global_option(const char* name, double* out_value);

/* ---- the online chain behind one coalescing front: replaces the body of the Flask route of src/online.py:108-163 (threaded
 * server, every request thread searches on module-level globals) minus the CNN: search (K nearest by cosine on g_search,
 * src/online.py:124-131) -> qge1 expansion from the first k_qe rows as stored in g_rows (src/utils/Reranking.py:195-208; the
 * route uses k_qe = 3, w = 4, src/online.py:148) -> re-search of g_rows with the expanded query as it is.  g_rows = NULL: the
 * plain search.  A launch of <= 128 queries costs what a launch of one does, so concurrent callers of mi_online_query are
 * answered TOGETHER: they block inside the call (a Python host's request threads: outside the interpreter lock) while one
 * worker thread of the handle drains the waiting descriptors -- up to max_batch rows -- into one chain and hands every caller
 * its rows.  The worker waits -- at most until max_wait_us after it handed out the previous chain -- for as many requests as
 * callers were around then (answered + queued): the callers just answered are on their way back, and two half crowds taking
 * turns are half the throughput of one; a lone sequential caller never waits, nor does anyone after an idle period.  Every chain runs the verified
 * loop of the host entry points (sticky flags read, f32 scorer / dense float64 fallbacks), so the answers are those of
 * sequential mi_knn_search + mi_aqe_search calls bit for bit.  The galleries must outlive the handle (mi_gallery_destroy
 * refuses a gallery an online handle is built on) and share device, dimension and rows; other threads may keep using them
 * (the chain takes their locks). */
typedef struct mi_online mi_online;
int mi_online_create(mi_gallery* g_search, mi_gallery* g_rows, int32_t k, int32_t k_qe, double w, double eps,
                     int32_t max_batch /*1..1024, 128: the streaming kernel's limit*/, int32_t max_wait_us, mi_online** out);
/* desc: nq <= max_batch descriptors [nq][d] f32, host or device memory.  pending != 0 (device descriptors only): they are still
 * being produced on producer_stream (NULL = the null stream) -- an event is recorded there and the chain waits for it on the
 * device; pending = 0: they are complete.  Blocks until the chain that carried the request is done; out_idx [nq][k] int64 (global row ids) and out_score [nq][k] f32 (may be NULL) are HOST arrays.
 * A failed chain fails every request it carried, with the chain's message on each caller's thread (mi_last_error). */
int mi_online_query(mi_online* o, const float* desc, int32_t nq, int memspace, int pending, void* producer_stream,
                    int64_t* out_idx, float* out_score);
int mi_online_stats(mi_online* o, int64_t* out_chains, int64_t* out_requests);
/* Answers what is still queued, stops the worker, frees the handle (not the galleries). */
int mi_online_destroy(mi_online* o);

/* The XCD shares of the tile kernel (relative speeds of the eight XCD labels, summing to 1) as the handle's launches have left
 * them, and how many launches have updated them since the workspace was created (-1: no workspace yet; the values are then what
 * the first one will start from).  The shares are saved with the prepared-gallery file (optional trailer) and remembered per
 * device inside the process, so `load -> first search` and a second gallery of a process start calibrated. */
int mi_debug_xcc_shares(mi_gallery* g, float* out_w8, int32_t* out_launches);
/* Diagnostics / bench only: `threads` request threads of the library itself in a closed loop of per_thread requests each (request
 * i of thread t = descriptor (t + i) mod n_desc of desc_dev [n_desc][d], complete) -- what the coalescing front sustains when
 * the host's request threads are not serialised by an interpreter lock.  out_last_idx [threads][k] host: every thread's last answer. */
int mi_debug_online_clients(mi_online* o, const float* desc_dev, int32_t n_desc, int32_t threads, int32_t per_thread,
                            int64_t* out_last_idx, double* out_seconds);
/* Diagnostics only: the per-wave words the tile kernel leaves behind, layout [workgroups * 8][8]: word 5 = K-slices done, word 6 =
 * shader cycles and word 7 = 10-ns ticks around the main loop (what kernel_clock_mhz and the XCD shares are computed from); words
 * 0-4 are written by the stamped build of scripts/kbench.hip only. */
int mi_debug_read_cycles(mi_gallery* g, uint64_t* out_host, int64_t count);

/* Diagnostics only: the gallery row that sample row i of the bootstrap sample image is drawn from (shard of n rows, sample
 * of n_s rows: one hashed draw per stratum of n / n_s consecutive rows).  Tests use it to plant rows inside the sample. */
int64_t mi_debug_sample_source_row(int64_t i, int64_t n, int64_t n_s);

/* ---- synthetic data (bench / tests): device twin of synth.synth_rows. */
int mi_synth_fill_device(float* dst_dev, uint64_t seed, int64_t row0, int64_t nrows, int32_t d,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_RETRIEVAL_H */
