// Per-handle options, statistics, launch timing, sticky flags, diagnostics.
#include "api_internal.h"

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
  else if (n == "sample_rows") *out_value = (double)sample_rows_in_effect(g);
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

}  // extern "C"
