// The online chain behind ONE native front (src/online.py:108-163: Flask's threaded server, every request thread runs the route
// on module-level globals): search (K nearest by cosine, src/online.py:124-131) -> qge1 expansion from the first k_qe rows AS
// STORED (src/utils/Reranking.py:195-208 with k = 3, w = 4, src/online.py:148) -> re-search with the expanded query.
//
// A launch of <= 128 queries costs what a launch of one does (bench.py q1 / q70: the gallery streams once), so concurrent
// request threads are answered TOGETHER: they block in mi_online_query -- outside the interpreter lock of a Python host --
// while one worker thread of the handle drains the waiting descriptors into one chain and hands every caller its rows.
// The answers are those of sequential calls bit for bit: the exact float64 re-score defines them, not the batch they rode in.
// (Round 6 first had this worker in Python, entry/online.py: 64 client threads spend more interpreter time handing requests
// over than the GPU spends answering them -- 10-14 k queries/s against 0.75 k uncoalesced)
#include <condition_variable>
#include <cstring>
#include <deque>
#include <new>
#include <thread>

#include "api_internal.h"

namespace {

struct OnlineReq {
  const float* desc;
  int32_t nq;
  int memspace;
  hipEvent_t ev;           // behind the producer's work on the caller's stream, or null: the descriptor is complete
  int64_t* out_idx;
  float* out_score;
  int rc = MI_OK;
  std::string err;
  bool done = false;
  std::mutex m;
  std::condition_variable cv;
};

// rows from anywhere the device can read (device memory, the handle's pinned host slots) into the batch
__global__ __launch_bounds__(256) void online_gather_kernel(const float* const* __restrict__ src, int32_t d,
                                                            float* __restrict__ dst) {
  const float* s = src[blockIdx.x];
  float* o = dst + (size_t)blockIdx.x * d;
  for (int c = threadIdx.x; c < d; c += 256) o[c] = s[c];
}

}  // namespace

struct mi_online {
  mi_gallery* g1 = nullptr;   // the gallery searched first (its norm_mode normalises the descriptors)
  mi_gallery* g2 = nullptr;   // the rows as stored: expansion + re-search; null = plain search
  int device = 0;             // of both galleries; the handle's own copy: destroy must not touch galleries that may be gone
  int32_t k = 0, k_qe = 0, max_batch = 0, max_wait_us = 0;
  double w = 0, eps = 0;
  std::mutex mu;
  std::condition_variable cv_work;
  std::deque<OnlineReq*> queue;
  std::vector<hipEvent_t> events;   // free list (under mu): a request whose descriptor is still being produced borrows one
  bool stop = false;
  std::thread worker;
  int64_t batches = 0, requests = 0;
  // device
  float *batch_d = nullptr, *sc_d = nullptr, *qx_d = nullptr;
  int64_t* idx_d = nullptr;
  double* sum_d = nullptr;
  // pinned host, readable by the device
  float *batch_h = nullptr, *sc_h = nullptr;
  int64_t* idx_h = nullptr;
  const float** ptr_h = nullptr;
};

namespace {

void online_free(mi_online* o) {
  (void)hipSetDevice(o->device);
  for (hipEvent_t e : o->events) (void)hipEventDestroy(e);
  for (void* p : {(void*)o->batch_d, (void*)o->sc_d, (void*)o->qx_d, (void*)o->idx_d, (void*)o->sum_d})
    if (p) (void)hipFree(p);
  for (void* p : {(void*)o->batch_h, (void*)o->sc_h, (void*)o->idx_h, (void*)o->ptr_h})
    if (p) (void)hipHostFree(p);
}

// one chain for the requests in `take` (rows in total); every error is the whole batch's
int online_chain(mi_online* o, const std::vector<OnlineReq*>& take, int32_t rows) {
  mi_gallery* g1 = o->g1;
  const int32_t d = g1->d, k = o->k;
  {
    std::lock_guard<std::mutex> lock(g1->mu);
    hipStream_t s = g1->stream;
    int32_t r0 = 0;
    for (OnlineReq* r : take) {
      if (r->ev) HIPC(hipStreamWaitEvent(s, r->ev, 0));
      for (int32_t i = 0; i < r->nq; ++i) {
        if (r->memspace == MI_HOST) {
          std::memcpy(o->batch_h + (size_t)(r0 + i) * d, r->desc + (size_t)i * d, (size_t)d * 4);
          o->ptr_h[r0 + i] = o->batch_h + (size_t)(r0 + i) * d;
        } else {
          o->ptr_h[r0 + i] = r->desc + (size_t)i * d;
        }
      }
      r0 += r->nq;
    }
    hipLaunchKernelGGL(online_gather_kernel, dim3(rows), dim3(256), 0, s, o->ptr_h, d, o->batch_d);
    HIPC(hipGetLastError());
    const int qn = g1->qnorm_override >= 0 ? g1->qnorm_override : g1->norm_mode;
    int rc = search_sync(g1, o->batch_d, MI_F32, d, 1, qn, rows, k, o->idx_d, o->sc_d, nullptr);
    if (rc != MI_OK) return rc;
  }
  if (o->g2) {
    mi_gallery* g2 = o->g2;
    std::lock_guard<std::mutex> lock(g2->mu);
    hipStream_t s = g2->stream;
    // ranks[j][q] = idx[q][j]: the first k_qe columns of the result just written (global row ids)
    launch_aqe_partial(g2->gal_f32, g2->dp, g2->d, g2->n, g2->row_offset, o->idx_d, 1, k, rows, o->k_qe, o->w, nullptr,
                       o->sum_d, s);
    launch_aqe_finish(o->sum_d, rows, d, o->eps, o->qx_d, nullptr, s);
    HIPC(hipGetLastError());
    // the expanded query is used as it is (no second normalisation), like `np.dot(vecs.T, qvecs_qe)`
    int rc = search_sync(g2, o->qx_d, MI_F32, d, 1, MI_NORM_NONE, rows, k, o->idx_d, o->sc_d, nullptr);
    if (rc != MI_OK) return rc;
  }
  HIPC(hipMemcpy(o->idx_h, o->idx_d, (size_t)rows * k * 8, hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(o->sc_h, o->sc_d, (size_t)rows * k * 4, hipMemcpyDeviceToHost));
  return MI_OK;
}

void online_serve(mi_online* o) {
  (void)hipSetDevice(o->device);
  std::vector<OnlineReq*> take;
  // the callers known to be around when the last chain was handed out: the ones it answered (on their way back) + the ones
  // already queued behind it
  int expect = 1;
  auto t_done = std::chrono::steady_clock::now();
  for (;;) {
    take.clear();
    int32_t rows = 0;
    {
      std::unique_lock<std::mutex> lk(o->mu);
      o->cv_work.wait(lk, [&] { return o->stop || !o->queue.empty(); });
      if (o->stop && o->queue.empty()) return;
      if (expect > 1 && o->max_wait_us > 0) {
        // A chain of 64 descriptors costs what a chain of one does, so the chain worth launching holds EVERY caller that is
        // around: wait -- until max_wait_us after the last hand-out at most -- for as many requests as that.  (Launching
        // whatever is queued when a chain ends settles into two half crowds taking turns: half the throughput.)  A lone
        // sequential caller (expect 1) never waits, and neither does anyone after an idle period.
        const int want = std::min(expect, (int)o->max_batch);
        o->cv_work.wait_until(lk, t_done + std::chrono::microseconds(o->max_wait_us),
                              [&] { return o->stop || (int)o->queue.size() >= want; });
      }
      while (!o->queue.empty() && (take.empty() || rows + o->queue.front()->nq <= o->max_batch)) {
        take.push_back(o->queue.front());
        rows += o->queue.front()->nq;
        o->queue.pop_front();
      }
      o->batches += 1;
      o->requests += (int64_t)take.size();
    }
    const int rc = online_chain(o, take, rows);
    const std::string err = rc == MI_OK ? std::string() : std::string(last_error_message());
    {
      std::lock_guard<std::mutex> lk(o->mu);
      expect = (int)take.size() + (int)o->queue.size();
    }
    int32_t r0 = 0;
    for (OnlineReq* r : take) {
      if (rc == MI_OK) {
        std::memcpy(r->out_idx, o->idx_h + (size_t)r0 * o->k, (size_t)r->nq * o->k * 8);
        if (r->out_score) std::memcpy(r->out_score, o->sc_h + (size_t)r0 * o->k, (size_t)r->nq * o->k * 4);
      }
      r0 += r->nq;
      std::lock_guard<std::mutex> lk(r->m);      // notified under the lock: the request lives on its caller's stack
      r->rc = rc;
      r->err = err;
      r->done = true;
      r->cv.notify_one();
    }
    t_done = std::chrono::steady_clock::now();
  }
}

}  // namespace

extern "C" {

int mi_online_create(mi_gallery* g_search, mi_gallery* g_rows, int32_t k, int32_t k_qe, double w, double eps,
                     int32_t max_batch, int32_t max_wait_us, mi_online** out) {
  REQUIRE(g_search && out, "null pointer");
  REQUIRE(max_batch >= 1 && max_batch <= QB, "max_batch must be in [1, 1024]");
  REQUIRE(max_wait_us >= 0, "negative wait");
  int rc = check_k(g_search, k);
  if (rc != MI_OK) return rc;
  if (g_rows) {
    REQUIRE(g_rows->d == g_search->d && g_rows->device == g_search->device, "the two galleries differ in dimension or device");
    REQUIRE(g_rows->n == g_search->n && g_rows->row_offset == g_search->row_offset, "the two galleries hold different rows");
    REQUIRE(k_qe >= 1 && k_qe <= k, "k_qe must be in [1, k]");
    if ((rc = check_k(g_rows, k)) != MI_OK) return rc;
  }
  HIPC(hipSetDevice(g_search->device));
  mi_online* o = new (std::nothrow) mi_online();
  if (!o) return fail(MI_ERR_NOMEM, "online handle");
  o->g1 = g_search;
  o->device = g_search->device;
  o->g2 = g_rows;
  o->k = k;
  o->k_qe = k_qe;
  o->w = w;
  o->eps = eps;
  o->max_batch = max_batch;
  o->max_wait_us = max_wait_us;
  const size_t mb = (size_t)max_batch, d = (size_t)g_search->d;
  bool ok = device_malloc((void**)&o->batch_d, mb * d * 4) == hipSuccess &&
            device_malloc((void**)&o->qx_d, mb * d * 4) == hipSuccess &&
            device_malloc((void**)&o->sum_d, mb * d * 8) == hipSuccess &&
            device_malloc((void**)&o->idx_d, mb * k * 8) == hipSuccess &&
            device_malloc((void**)&o->sc_d, mb * k * 4) == hipSuccess &&
            hipHostMalloc((void**)&o->batch_h, mb * d * 4, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&o->idx_h, mb * k * 8, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&o->sc_h, mb * k * 4, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&o->ptr_h, mb * sizeof(void*), hipHostMallocDefault) == hipSuccess;
  if (!ok) {
    online_free(o);
    delete o;
    return fail(MI_ERR_NOMEM, "online chain buffers");
  }
  try {
    o->worker = std::thread(online_serve, o);
  } catch (const std::exception& e) {                     // no exception crosses the C boundary
    online_free(o);
    delete o;
    return fail(MI_ERR_NOMEM, std::string("could not start the worker thread: ") + e.what());
  }
  g_search->online_users.fetch_add(1);
  if (g_rows) g_rows->online_users.fetch_add(1);
  *out = o;
  return MI_OK;
}

int mi_online_query(mi_online* o, const float* desc, int32_t nq, int memspace, int pending, void* producer_stream,
                    int64_t* out_idx, float* out_score) {
  REQUIRE(o && desc && out_idx, "null pointer");
  REQUIRE(nq >= 1 && nq <= o->max_batch, "a request holds 1 .. max_batch descriptors");
  REQUIRE(memspace == MI_HOST || memspace == MI_DEVICE, "bad memspace");
  OnlineReq r;
  r.desc = desc;
  r.nq = nq;
  r.memspace = memspace;
  r.ev = nullptr;
  r.out_idx = out_idx;
  r.out_score = out_score;
  auto give_back = [&] {
    if (!r.ev) return;
    std::lock_guard<std::mutex> lk(o->mu);
    o->events.push_back(r.ev);
  };
  if (memspace == MI_DEVICE && pending) {
    // the event belongs to the handle's device; the calling thread keeps the device it had
    int prev = o->device;
    (void)hipGetDevice(&prev);
    struct Restore {
      int dev, want;
      ~Restore() { if (dev != want) (void)hipSetDevice(dev); }
    } restore{prev, o->device};
    if (prev != o->device) HIPC(hipSetDevice(o->device));
    {
      std::lock_guard<std::mutex> lk(o->mu);
      if (!o->events.empty()) {
        r.ev = o->events.back();
        o->events.pop_back();
      }
    }
    if (!r.ev) HIPC(hipEventCreateWithFlags(&r.ev, hipEventDisableTiming));
    if (hipEventRecord(r.ev, (hipStream_t)producer_stream) != hipSuccess) {
      give_back();
      return fail(MI_ERR_HIP, "hipEventRecord on the producer's stream failed");
    }
  }
  {
    std::unique_lock<std::mutex> lk(o->mu);
    if (o->stop) {
      lk.unlock();
      give_back();
      return fail(MI_ERR_INVALID, "the online handle is closing");
    }
    o->queue.push_back(&r);
  }
  o->cv_work.notify_one();
  {
    std::unique_lock<std::mutex> lk(r.m);
    r.cv.wait(lk, [&] { return r.done; });
  }
  give_back();
  if (r.rc != MI_OK) return fail(r.rc, r.err);       // the worker's message, on the caller's thread
  return MI_OK;
}

int mi_online_stats(mi_online* o, int64_t* batches, int64_t* requests) {
  REQUIRE(o, "null pointer");
  std::lock_guard<std::mutex> lk(o->mu);
  if (batches) *batches = o->batches;
  if (requests) *requests = o->requests;
  return MI_OK;
}

// Diagnostics / bench only: `threads` request threads of the LIBRARY in a closed loop, one descriptor per request -- what the
// front sustains when the host's request threads are not serialised by an interpreter lock.
int mi_debug_online_clients(mi_online* o, const float* desc_dev, int32_t n_desc, int32_t threads, int32_t per_thread,
                            int64_t* out_last_idx, double* out_seconds) {
  REQUIRE(o && desc_dev && out_last_idx && out_seconds, "null pointer");
  REQUIRE(n_desc >= 1 && threads >= 1 && threads <= 1024 && per_thread >= 1, "bad sizes");
  const int32_t d = o->g1->d, k = o->k;
  std::vector<int> rcs((size_t)threads, MI_OK);
  std::vector<std::string> errs((size_t)threads);
  std::vector<std::thread> th;
  const auto t0 = std::chrono::steady_clock::now();
  bool started = true;
  for (int32_t t = 0; t < threads && started; ++t) try {
    th.emplace_back([&, t] {
      for (int32_t i = 0; i < per_thread; ++i) {
        const int rc = mi_online_query(o, desc_dev + (size_t)((t + i) % n_desc) * d, 1, MI_DEVICE, 0, nullptr,
                                       out_last_idx + (size_t)t * k, nullptr);
        if (rc != MI_OK) {
          rcs[t] = rc;
          errs[t] = last_error_message();
          return;
        }
      }
    });
  } catch (const std::exception&) {
    started = false;
  }
  for (auto& x : th) x.join();
  if (!started) return fail(MI_ERR_NOMEM, "could not start the client threads");
  *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int32_t t = 0; t < threads; ++t)
    if (rcs[t] != MI_OK) return fail(rcs[t], errs[t]);
  return MI_OK;
}

int mi_online_destroy(mi_online* o) {
  if (!o) return MI_OK;
  {
    std::lock_guard<std::mutex> lk(o->mu);
    o->stop = true;
  }
  o->cv_work.notify_all();
  if (o->worker.joinable()) o->worker.join();          // answers what is still queued, then returns
  o->g1->online_users.fetch_sub(1);
  if (o->g2) o->g2->online_users.fetch_sub(1);
  online_free(o);
  delete o;
  return MI_OK;
}

}  // extern "C"
