// The prepared-gallery file (MI355GAL v2): save / load through rings of pinned buffers, per-section device checksums.
#include "api_internal.h"

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
  std::atomic<int> io_err{0}, hip_err{0} /* the failing hipError_t of a writer thread, 0 = none */, stop{0};
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
          const hipError_t we = hipEventSynchronize(ev[i % ring]);
          if (we != hipSuccess) hip_err = (int)we;
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
  if (he != hipSuccess || hip_err.load())
    return fail(MI_ERR_HIP, std::string("device -> gallery file: ") +
                                hipGetErrorString(he != hipSuccess ? he : (hipError_t)hip_err.load()));
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
// Section checksums on a stream of the CALL (mi_gallery_save runs on a writer thread beside searches of the same handle,
// nnsearch._save_behind: the handle's own stream belongs to the host entry points, under g->mu)
int section_sums(const mi_gallery* g, uint64_t out[3]) {
  unsigned long long* d = nullptr;
  hipStream_t s = nullptr;
  HIPC(device_malloc((void**)&d, 24));
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e == hipSuccess) {
    launch_checksum(g->gal_f32, (size_t)g->n * g->dp * 4, d + 0, s);
    launch_checksum(g->gal_img, (size_t)g->npad * g->dp * 2, d + 1, s);
    launch_checksum(g->rowstat, (size_t)g->npad * sizeof(RowStat), d + 2, s);
    e = hipMemcpyAsync(out, d, 24, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  if (s) (void)hipStreamDestroy(s);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MI_ERR_HIP, std::string("checksum: ") + hipGetErrorString(e));
  return MI_OK;
}
}  // namespace


extern "C" {

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
  int rc = section_sums(g, h.section_sum);      // (the gallery buffers are immutable once ingested; appends are the caller's to serialise)
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
    bool have;
    {
      // the shares live in the search workspace, which a host entry point of another thread may be re-building (ws_ensure)
      std::lock_guard<std::mutex> lock(const_cast<mi_gallery*>(g)->mu);
      have = snapshot_balance(g, tr.w);
    }
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

}  // extern "C"
