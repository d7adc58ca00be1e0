// Process-wide state of the library and the allocation of a handle's search workspace.
#include "api_internal.h"

static thread_local std::string g_err;        // message of the calling thread's last failure
std::atomic<int> g_default_img_f16{1};   // mi_set_global_option("image_dtype", 0 = bf16 | 1 = fp16); read at gallery creation
// mi_set_global_option("host_ingest", ...): how mi_gallery_create moves a HOST array to the device.  1 (default) = row blocks of
// ~32 MiB copied straight from the caller's (pageable) array by the runtime into two alternating device blocks, the copy of
// block i + 1 under the ingest of block i, no staging the size of the gallery; 0 = one copy of the whole array into a same-size
// staging allocation, then one ingest (rounds 1-4).  Measured at 1 005 994 x 2048 float32 (profiles/r05f_host_ingest_modes.json):
// 53.1 and 53.6 GB/s = 0.96 of the box's pinned H2D rate -- the runtime's pageable path is as fast as a pinned copy here.  A
// third mode, blocks through two pinned buffers filled by 2-8 host threads (what VERDICT r04 prescribed), reached 30-35 GB/s
// and was removed again.
std::atomic<int> g_host_ingest{1};
// XCD shares of the tile kernel as the last handle on a device left them (common.h XccBalance): a handle created later -- or
// loaded from a file written without shares -- starts from these instead of from an even split
std::mutex g_bal_mu;
std::map<int, std::vector<float>> g_bal_cache;
// mi_set_global_option("keep_buffers", 0 | 1): a caller that prepares a gallery per call -- create, search, destroy: what a
// stateless matching_<method>(K, train, test) is (src/utils/nnsearch.py:687-706) -- pays hipMalloc + hipFree of the gallery's
// buffers every time (12.4 GB at 1 005 994 x 2048: 1-6 ms, more than half an ingest, and an idle GPU meanwhile).  With 1
// (default) mi_gallery_destroy hands the four buffers of a gallery of up to 16 GiB to ONE spare slot per process instead of
// freeing them, and the next gallery of exactly the same sizes on the same device takes them (every byte a search reads is
// written by the ingest or by an explicit memset; nothing depends on fresh memory).  0 frees the spare and stops keeping.
std::mutex g_spare_mu;
SpareBuffers g_spare;
SpareArena g_spare_ws;
std::atomic<int> g_keep_buffers{1};
void spare_release_locked() {
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
void spare_ws_release_locked() {
  if (!g_spare_ws.p) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(g_spare_ws.device);
  (void)hipFree(g_spare_ws.p);
  (void)hipSetDevice(cur);
  g_spare_ws = SpareArena();
}
// hipMalloc for everything the library allocates: when the device is out of memory the spare slots (up to 16 GiB of a destroyed
// gallery + its ~200 MB workspace, invisible to any other allocator of the process -- the extractor's PyTorch caching
// allocator, a second gallery) are given back and the allocation is tried once more.
hipError_t device_malloc(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipErrorOutOfMemory) return e;
  (void)hipGetLastError();
  {
    std::lock_guard<std::mutex> lock(g_spare_mu);
    if (g_spare.device < 0 && !g_spare_ws.p) return e;
    spare_release_locked();
    spare_ws_release_locked();
  }
  return hipMalloc(p, bytes);
}
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
const char* last_error_message() { return g_err.c_str(); }

int ws_free(Workspace& ws) {
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


int ws_ensure(mi_gallery* g, int32_t k) {
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
      if (!v) HIPC(device_malloc(&v, total));
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
bool snapshot_balance(const mi_gallery* g, float* out_w8) {
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

extern "C" {

const char* mi_last_error(void) { return last_error_message(); }

int mi_device_count(int* count) {
  REQUIRE(count, "null");
  HIPC(hipGetDeviceCount(count));
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
  else if (n == "release_spares") {
    // gives the spare slots back NOW and leaves the mode alone (a caller that is done with its galleries for a while --
    // nnsearch.drop_cached_galleries -- or a co-tenant that needs the memory)
    std::lock_guard<std::mutex> lock(g_spare_mu);
    spare_release_locked();
    spare_ws_release_locked();
  }
  else return fail(MI_ERR_INVALID, "unknown global option: " + n);
  return MI_OK;
}

int mi_get_global_option(const char* name, double* out_value) {
  REQUIRE(name && out_value, "null");
  const std::string n(name);
  if (n == "image_dtype") *out_value = g_default_img_f16.load();
  else if (n == "host_ingest") *out_value = g_host_ingest.load();
  else if (n == "keep_buffers") *out_value = g_keep_buffers.load();
  else if (n == "spare_bytes") {
    // device memory this process holds in the spare slots right now (gallery buffers + search workspace of destroyed handles)
    std::lock_guard<std::mutex> lock(g_spare_mu);
    *out_value = (double)((g_spare.device >= 0 ? g_spare.f32_bytes + g_spare.img_bytes + g_spare.stat_bytes : 0) +
                          (g_spare_ws.p ? g_spare_ws.bytes : 0));
  }
  else return fail(MI_ERR_INVALID, "unknown global option: " + n);
  return MI_OK;
}

}  // extern "C"
