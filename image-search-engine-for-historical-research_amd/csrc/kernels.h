// Host-callable launchers of the gfx950 kernels (internal).
#pragma once
#include "common.h"

namespace mi {

// ingest.hip
// strided source -> row-major [m][d] copy of the same element type (scratch for layouts launch_ingest does not take in one pass)
void launch_transpose_rows(const void* src, int dtype, int64_t m, int32_t d, int64_t rs, int64_t cs, void* dst, hipStream_t stream);
bool ingest_takes_layout(int32_t d, int64_t rs, int64_t cs);
void launch_ingest(const void* src, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, int norm_mode,
                   float* out_f32, void* out_img, int img_f16, RowStat* rowstat, int32_t dp, int64_t npad, hipStream_t stream,
                   int64_t row_base = 0);
// query batch of the search path: ingest + the per-query search state (launch_init_query_state) in ONE launch; returns false
// (nothing launched) when the rows are too wide for it
bool launch_ingest_queries(const void* src, int dtype, int32_t nq, int32_t d, int64_t rs, int64_t cs, int norm_mode,
                           float* out_f32, void* out_img, int img_f16, RowStat* rowstat, int32_t dp, int32_t qpad,
                           const float* gstat3, float gamma, int use_img_terms, uint32_t first_cnt, const struct QueryState& st,
                           hipStream_t stream, uint32_t zero_scores = 0);
// bootstrap sample image (n_s = multiple of TILE rows, one hashed draw per stratum of the shard)
void launch_build_sample(const void* gal_img, void* samp_img, int64_t n, int64_t n_s, int32_t dp, hipStream_t stream);
void launch_build_sample_f32(const float* gal_f32, float* samp_f32, int64_t n, int64_t n_s, int32_t dp, hipStream_t stream);
int64_t sample_source_row_host(int64_t i, int64_t n, int64_t n_s);   // row_base: output rows start here (gallery append); src row 0 <-> row_base
void launch_checksum(const void* data, size_t bytes, unsigned long long* out_dev, hipStream_t stream);   // bytes % 8 == 0
void launch_rowstat_max(const RowStat* rowstat, int64_t n, float* out3, hipStream_t stream, bool reset = true);

// gemm_select.hip -- fp16/bf16 MFMA scoring of gallery tiles [tile0, tile0+ntiles) against nqt query tiles with the
// fused survivor filter.  first != 0: store every score of the chunk (rows tile0*256.. at position row).
struct ScoreArgs {
  const void* gal_img;   // blocked image of the shard
  const void* qry_img;   // blocked image of the query batch
  int32_t img_f16;        // 16-bit image element type: 1 = fp16, 0 = bf16
  int32_t nslices;        // dp / 32
  int32_t tile0, ntiles;  // gallery tiles of this launch
  int32_t nqt;            // query tiles (qpad / 256)
  int64_t n;              // valid gallery rows in the shard
  int32_t nq;             // valid queries
  int32_t debug = 0;      // scripts/kbench.hip (-DMI_KBENCH builds) only: diagnostic instantiation; the product passes 0
  int32_t small_batch_kernel;   // 1: launches with <= STREAM_MAX_QUERIES queries go to stream_select.hip
  int32_t variant = 0;          // scripts/kbench.hip (-DMI_KBENCH builds) only: A/B instantiation; the product passes 0
  int32_t walk = 0;             // tile kernel: 0 = every XCD label walks all query tiles of its gallery range; 1 = labels 2y, 2y + 1
                                // share a range and take half of the query tiles each (A/B, gemm_select.hip)
  SurvRec* rec;           // [grid * 8 waves][rec_cap] wave-private survivor records of this launch
  uint32_t* rec_cnt;      // [grid * 8]
  uint32_t rec_cap;
  const uint32_t* cond;   // non-null: the whole launch is skipped when *cond == 0 (repair pass)
  const XccBalance* bal = nullptr;   // non-null: weighted split of the gallery tiles over the XCD labels (tile kernel)
  int32_t lad_k = 0;                 // > 0: in-launch threshold ladder on (K of the search); tile kernel, filtered launch only
  int32_t scores_only = 0;           // bootstrap launch on the sample image (stream_select MODE 2): store the scores as 4-byte
                                     // floats at ((float*)(surv + q * cap))[sample row] instead of 8-byte (score, row) entries --
                                     // sample_threshold_kernel reads nothing but the scores, and the entries are dropped afterwards
  float* samp_out = nullptr;         // scores_only: non-null -> the scores go to samp_out[q * samp_ld + sample row] instead of the
  uint32_t samp_ld = 0;              // survivor rows (the 65 536-row sample of shards beyond 3.9 M rows does not fit one)
  int32_t ksplit = 1;                // bootstrap launch with scores_only: workgroups per (sample tile, query group), each taking
                                     // nslices / ksplit K-slices and ADDING its partial scores (atomic f32 adds onto zeros
                                     // written by the query ingest): small batches have 32 .. 64 bootstrap workgroups of 64
                                     // serial slices each otherwise.  The sum order is then not fixed -- the sample scores
                                     // feed the speculative (verified) threshold and the ladder level only
  unsigned long long* dbg; // diagnostics (DBG & 8): per-wave cycle sums, [grid * 8][8]
  QueryState st;
};
// stream_select.hip: the scoring + filter launch for small query batches (HBM-bound; same records and thresholds)
constexpr int STREAM_MAX_QUERIES = 128;
bool stream_select_applies(const ScoreArgs& a);
bool stream_bootstrap_applies(const ScoreArgs& a);   // bootstrap launches of any batch size
void launch_stream_select(const ScoreArgs& a, bool first, hipStream_t stream);
void launch_gemm_select(const ScoreArgs& a, bool first, hipStream_t stream);
unsigned gemm_select_grid();   // persistent grid size (workgroups); record segments = grid * 8
// A wave whose private record segment is full (rec_cap records in one launch: ten times what a batch of descriptors leaves
// per wave, but a batch whose queries are ordered like the gallery -- five queries per landmark, landmarks stored together:
// bench.py `hard_data` -- or a batch of near-identical queries concentrates its survivors on the few waves that own the
// matching (gallery tile, query block) pairs) appends the record straight to its query's survivor bucket: one RETURNING atomic
// per record on this cold path (it drains the wave's DMA ring, which is why the hot path avoids it), the same counter
// scatter_records_kernel adds to afterwards.  Round 5 raised FLAG_REC_OVERFLOW instead and the whole batch was answered again.
__device__ __forceinline__ void spill_record(const QueryState& st, float score, uint32_t row, uint32_t q) {
  const uint32_t pos = atomicAdd(&st.cnt[q * CNT_STRIDE], 1u);
  if (pos < st.cap) st.surv[(uint64_t)q * st.cap + pos] = pack_entry(score, row);
  else atomicOr(st.flags, FLAG_SURV_OVERFLOW);
}

// Timing of ONE scoring launch without extra packets on the stream: the next launch_gemm_select / launch_stream_select of
// this thread goes out through hipExtLaunchKernelGGL with these events, which receive the begin / end timestamps of the
// dispatch itself (events recorded around a launch with hipEventRecord are barrier packets of their own: ~5 us of gap
// before and after a 3 ms kernel).  Consumed by that launch.
void set_launch_events(hipEvent_t start, hipEvent_t stop);
void take_launch_events(hipEvent_t* start, hipEvent_t* stop);
// buckets the wave-private records of the last scoring launch into the per-query survivor buffers
// bal / dbg / ntiles non-null / non-zero: block 0 also updates the XCD shares from the loop times of the launch that wrote dbg
void launch_scatter_records(const SurvRec* rec, const uint32_t* rec_cnt, uint32_t rec_cap, uint32_t nseg,
                            QueryState st, const uint32_t* cond, hipStream_t stream, XccBalance* bal = nullptr,
                            const unsigned long long* dbg = nullptr, uint32_t ntiles = 0, int32_t nq = 0);   // nq: queries of the batch
void init_xcc_balance_host(XccBalance* host);
void init_xcc_balance_from(XccBalance* host, const float* w8);   // from remembered shares (file / process cache)

// exact_score.hip -- f32 FMA scoring with the same filter (fallback / force_exact)
struct ExactArgs {
  const float* gal_f32;   // [n][dp]
  const float* qry_f32;   // [qpad][dp]
  int32_t dp;
  int64_t row0, row1;     // gallery rows of this launch
  int64_t n;
  int32_t nq;
  QueryState st;
  float* dense_out = nullptr;   // non-null: dense mode, scores to dense_out[q * dense_ld + row]
  int64_t dense_ld = 0;
};
void launch_exact_select(const ExactArgs& a, bool first, hipStream_t stream);

// select.hip
void launch_init_query_state(const RowStat* qstat, const float* gstat3, int32_t nq, int32_t qpad, float gamma,
                             int use_img_terms, uint32_t first_cnt, QueryState st, hipStream_t stream);
// mode 0: maintain (threshold <- K-th largest - margin, compact survivors)
// mode 1: maintain + write the K largest approximate values to topvals[q][K] and L_local[q]
// thresholds from the 2048 / 4096 / 8192-score bootstrap sample (single-launch schedule), cheaper than launch_select_maintain(mode 0)
void set_tail_debug_phase(int phase);   // diagnostics only (scripts/tailbench.hip): selection kernels return after phase N; 0 = product
bool sample_threshold_applies(uint32_t first_cnt, int32_t k, int32_t spec_r);
// the r-th / (4 r)-th / lad_r-th largest of n sample scores per query read from scores[q * ld + i] (samples too large for a survivor row)
void launch_sample_threshold_big(QueryState st, const float* scores, uint32_t ld, uint32_t n, int32_t nq, int32_t k,
                                 int32_t spec_r, int32_t lad_r, float order_slack, hipStream_t stream);
void launch_sample_threshold(QueryState st, int32_t nq, int32_t k, int32_t spec_r, uint32_t first_cnt, hipStream_t stream,
                             int32_t lad_r = 0, int32_t f32_scores = 0, float order_slack = 0.f);   // f32_scores: see ScoreArgs::scores_only
// what the in-kernel repair of a failed query scans (repair == 3): the shard's stored f32 rows and the batch's f32 queries
struct RepairScan {
  const float* gal_f32 = nullptr;
  const float* qry_f32 = nullptr;
  int32_t dp = 0;
  int64_t n = 0;
  uint64_t* repairs = nullptr;   // per query: in-kernel repairs so far (one writer per word; summed by mi_search_status)
};
void launch_select_maintain(QueryState st, int32_t nq, int32_t k, int mode, float* topvals, float* l_local,
                            uint64_t* stats2, int32_t spec_r, int32_t spec, int32_t repair, const uint32_t* cond,
                            hipStream_t stream, uint32_t* cand_rows = nullptr, uint32_t* cand_cnt = nullptr,
                            uint32_t rcap = 0,   // cand_*: mode 1 also writes the candidate rows (single-shard search)
                            const RepairScan* scan = nullptr);   // repair == 3: the workgroup of a failed query re-scans the shard
void launch_select_candidates(QueryState st, int32_t nq, const float* L, uint32_t* cand_rows, uint32_t* cand_cnt,
                              uint32_t rcap, uint64_t* stats2, hipStream_t stream);
void launch_rescore(const float* gal_f32, const float* qry_f32, int32_t dp, int32_t nq, const uint32_t* cand_rows,
                    const uint32_t* cand_cnt, uint32_t rcap, double* cand_score, hipStream_t stream,
                    uint32_t grid_x, uint32_t last_row);   // last_row: n - 1 of the shard (row ids are clamped to it)
void launch_rescore_resident(const float* gal_f32, const float* qry_f32, int32_t dp, int32_t nq, const uint32_t* cand_rows,
                             const uint32_t* cand_cnt, uint32_t rcap, double* cand_score, hipStream_t stream,
                             uint32_t last_row);
void launch_emit(const uint32_t* cand_rows, const uint32_t* cand_cnt, const double* cand_score, uint32_t rcap,
                 int32_t nq, int32_t k, int64_t row_offset, int64_t* out_idx, float* out_score,
                 double* out_score64, hipStream_t stream);
void launch_kth_of_gathered(const float* gathered, int32_t nshards, int64_t nq, int32_t k, float* out_L,
                            hipStream_t stream);
void launch_merge(const double* score64, const int64_t* idx, int32_t nshards, int64_t nq, int32_t k, int64_t shard_stride,
                  int64_t* out_idx, float* out_score, hipStream_t stream);

// dense.hip -- exact top-k of dense score rows (global-memory radix select + LDS bitonic sort), k <= 4096
void launch_dense_topk(const float* scores, int64_t ld, int64_t n, int32_t nq, int32_t k, int64_t row_offset,
                       int64_t* out_idx, float* out_score, hipStream_t stream);

// last resort of the exhaustive search (massive ties): dense f64 scores of every row + exact top-k of them
void launch_dense_score64(const float* gal_f32, const float* qry_f32, int32_t dp, int64_t n, int32_t nq, double* out,
                          int64_t ld, hipStream_t stream);
void launch_dense_topk64(const double* scores, int64_t ld, int64_t n, int32_t nq, int32_t k, int64_t row_offset,
                         int64_t* out_idx, float* out_score, double* out_score64, hipStream_t stream);
void launch_rank_all(const float* scores, int64_t ld, int64_t n, int32_t nq, uint32_t* keys_a, uint32_t* idx_a,
                     uint32_t* keys_b, uint32_t* idx_b, int64_t row_offset, int64_t* out_idx, float* out_score,
                     hipStream_t stream);

void launch_rank_positions(const float* scores, int64_t ld, int64_t n, int32_t nq, const int64_t* ids, int32_t m,
                           int64_t row_offset, unsigned long long* out_pos, hipStream_t stream);
int rank_positions_max_listed();

// diffusion.hip
void launch_affinity(const int64_t* ids, const float* sims, int64_t ld, int64_t n, int32_t kd, int32_t gamma,
                     float alpha, float* lap, float* dinv, float* diag, hipStream_t stream);
void launch_diffusion_cg(const int64_t* ids, int64_t ld, int64_t n, int32_t T, int32_t kd, const float* lap,
                         const float* diag, int32_t maxiter, double tol, int32_t* map_all, unsigned grid,
                         int32_t* out_ids, float* out_vals, hipStream_t stream, int64_t node0 = 0,
                         int64_t node1 = -1);
void launch_diffusion_combine(const int64_t* nn_idx, const float* nn_sims, int32_t kq, int32_t gamma,
                              const int32_t* off_ids, const float* off_vals, int32_t T, int64_t n, int32_t nq,
                              float* dense, hipStream_t stream);

// kr_rerank.hip -- k-reciprocal re-ranking (src/utils/Reranking.py:447-624)
void launch_kr_pack(const void* src, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, float* out, int32_t dp,
                    hipStream_t stream);
void launch_kr_sets(const int64_t* rank, int ld, int all, int k1, int32_t* R, int32_t* Rcnt, uint32_t* flags,
                    hipStream_t stream);
void launch_kr_weights(const float* S, int all, const int32_t* R, const int32_t* Rcnt, float* V, float* dmax,
                       hipStream_t stream);
void launch_kr_expand(const int64_t* rank, int ld, int k2, int all, const int32_t* R, const int32_t* Rcnt, const float* V,
                      void* Vqe, void* VqeT, hipStream_t stream);
void launch_kr_final(const void* Vqe, const void* VqeT, const float* S, const float* dmax, int all, int nq, float w_jac,
                     float w_org, float* neg_final, uint32_t* flags, hipStream_t stream);
int kr_rmax();

// whiten.hip
void launch_whiten(const void* X, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, const double* m,
                   const double* P, int32_t dims, double eps, double* Y, hipStream_t stream);

// desc_tail.hip
void launch_desc_tail(const float* feat, int32_t b, int32_t c, int32_t hw, float p, float eps, const float* W,
                      const float* bias, int32_t c_out, float* pooled, float* out, hipStream_t stream);
void launch_ms_accumulate(float* acc, const float* desc, int64_t count, float msp, int first, hipStream_t stream);
void launch_ms_finish(float* acc, int32_t b, int32_t d, int32_t nscales, float msp, hipStream_t stream);

// aqe.hip
void launch_aqe_partial(const float* gal_f32, int32_t dp, int32_t d, int64_t n, int64_t row_offset,
                        const int64_t* ranks, int64_t sj, int64_t sq, int64_t nq, int32_t k_qe, double w,
                        const double* weights, double* out_sum, hipStream_t stream);
void launch_aqe_rows(const float* gal_f32, int32_t dp, int32_t d, int64_t n, int64_t row_offset, const int64_t* ranks,
                     int64_t sj, int64_t sq, int64_t nq, int32_t k_qe, float* out_rows, hipStream_t stream);
void launch_aqe_combine(const float* rows, int64_t nq, int32_t d, int32_t k_qe, double w, const double* weights,
                        double* out_sum, hipStream_t stream);
void launch_column_sum(const void* X, int dtype, int64_t n, int32_t d, int64_t rs, int64_t cs, double* out,
                       hipStream_t stream);
void launch_aqe_finish(const double* sum, int64_t nq, int32_t d, double eps, float* out_q, double* out_q64,
                       hipStream_t stream);

// synth.hip
void launch_synth_fill(float* dst, uint64_t seed, int64_t row0, int64_t nrows, int32_t d, hipStream_t stream);

}  // namespace mi
