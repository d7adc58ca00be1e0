// Gallery life cycle: create (device / host arrays of any layout), append, destroy (spare-buffer slots), norm bounds, image type.
#include "api_internal.h"

extern "C" {

int mi_gallery_destroy(mi_gallery* g) {
  if (!g) return MI_OK;
  // the worker thread of an online handle would search a freed gallery with the next request
  REQUIRE(g->online_users.load() == 0, "the gallery is in use by an online handle: mi_online_destroy comes first");
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
  (void)hipFree(g->samp_scores);
  (void)hipFree(g->samp_f32);
  for (void* b : g->io_buf) (void)hipFree(b);
  (void)hipFree(g->dif_ids);
  (void)hipFree(g->dif_vals);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return MI_OK;
}

}  // extern "C"

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

int gallery_alloc(mi_gallery* g) {
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
    // a gallery of other sizes: the spare is given back when it is big (> 8 GiB requested) or when the device could not hold
    // both (free memory below the request plus a quarter) -- and device_malloc releases it on any out-of-memory anyway
    size_t free_b = 0, total_b = 0;
    const size_t want = f32_bytes + bf_bytes + stat_bytes;
    if (g_spare.device >= 0 &&
        (want > SPARE_MAX_BYTES / 2 || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < want + want / 4)))
      spare_release_locked();
  }
  HIPC(device_malloc((void**)&g->gal_f32, f32_bytes + 256));
  HIPC(device_malloc(&g->gal_img, bf_bytes + 256));
  HIPC(device_malloc((void**)&g->rowstat, stat_bytes));
  HIPC(device_malloc((void**)&g->gstat3, 16));
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

extern "C" {

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
    hipError_t e = device_malloc(&staged, (size_t)elems * esz + 256);
    if (e != hipSuccess) return cleanup(fail(MI_ERR_NOMEM, std::string("staging allocation: ") + hipGetErrorString(e)));
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

}  // extern "C"
