// Standalone A/B driver for the scoring kernels of libmi355_retrieval.so (no Python, no torch): builds random
// 16-bit operand images with the score distribution of unit vectors (sigma = 1 / sqrt(D)), launches the kernel
// variants named on the command line in interleaved rounds inside ONE process (boxes of the pool differ by several
// per cent, and so do separate processes), times every launch with HIP events and checks that all variants emit the
// same multiset of survivor records.
//
//   kbench [--rows N] [--dim D] [--queries Q] [--rounds R] [--reps K] [--thr T] [--bf16] variant...
//   variant = <label>:<debug>[:<variant id>]       e.g.  base:0  nofilter:4  stamps:8  v2:0:2
//
// Build: see scripts/kbench_build.sh.  Diagnostics only; nothing here is part of the product path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../image-search-engine-for-historical-research_amd/csrc/kernels.h"

using namespace mi;

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// element e of the image: approximately N(0, scale^2) (sum of four uniforms), stored as fp16 or bf16
__global__ void fill_image(uint16_t* img, size_t count, uint64_t seed, float scale, int f16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix(seed ^ (i * 0x2545F4914F6CDD1Dull));
    float u = 0.f;
    for (int j = 0; j < 4; ++j) u += (float)((h >> (16 * j)) & 0xFFFF) * (1.0f / 65536.0f);
    const float v = (u - 2.0f) * 1.7320508f * scale;   // var of sum of 4 U(0,1) = 1/3
    uint16_t bits;
    if (f16) {
      const _Float16 hv = (_Float16)v;
      bits = *reinterpret_cast<const uint16_t*>(&hv);
    } else {
      const uint32_t ub = __float_as_uint(v);
      bits = (uint16_t)((ub + 0x7FFFu + ((ub >> 16) & 1u)) >> 16);
    }
    img[i] = bits;
  }
}

__global__ void fill_f32(float* p, int n, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// order-independent checksum of the survivor records of one launch
__global__ void rec_checksum(const SurvRec* rec, const uint32_t* rec_cnt, uint32_t rec_cap, uint32_t nseg,
                             unsigned long long* out2) {
  const uint32_t seg = blockIdx.x;
  if (seg >= nseg) return;
  const uint32_t n = min(rec_cnt[seg], rec_cap);
  unsigned long long s = 0, c = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const SurvRec r = rec[(size_t)seg * rec_cap + i];
    s += mix(((uint64_t)r.q << 40) ^ ((uint64_t)r.row << 8) ^ ((uint64_t)__float_as_uint(r.score) * 0x9E3779B1ull));
    c += 1;
  }
  atomicAdd(&out2[0], s);
  atomicAdd(&out2[1], c);
}

struct Variant {
  std::string label;
  int debug = 0, variant = 0;
  std::vector<float> ms;
  unsigned long long sum = 0, cnt = 0;
};

int main(int argc, char** argv) {
  int64_t rows = 1005994;
  int d = 2048, nq = 1024, rounds = 3, reps = 5, f16 = 1, rotate = 0;
  float thr = 0.0663f;
  std::vector<Variant> vs;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : "0"); };
    if (a == "--rows") rows = atoll(next().c_str());
    else if (a == "--dim") d = atoi(next().c_str());
    else if (a == "--queries") nq = atoi(next().c_str());
    else if (a == "--rounds") rounds = atoi(next().c_str());
    else if (a == "--reps") reps = atoi(next().c_str());
    else if (a == "--thr") thr = (float)atof(next().c_str());
    else if (a == "--bf16") f16 = 0;
    else if (a == "--rotate") rotate = 1;     // round r starts at variant r: no variant is always the one behind the same neighbour
    else {
      Variant v;
      const size_t c1 = a.find(':');
      v.label = a.substr(0, c1);
      if (c1 != std::string::npos) {
        const size_t c2 = a.find(':', c1 + 1);
        v.debug = atoi(a.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1).c_str());
        if (c2 != std::string::npos) v.variant = atoi(a.substr(c2 + 1).c_str());
      }
      vs.push_back(v);
    }
  }
  if (vs.empty()) vs.push_back(Variant{"base", 0, 0});
  const int dp = (int)round_up(d, BK);
  const int64_t npad = round_up(rows, TILE), ntiles = npad / TILE;
  const int qpad = (int)round_up(nq, TILE);
  const int nsl = dp / SLICE_K;
  const size_t gal_elems = (size_t)npad * dp, q_elems = (size_t)qpad * dp;
  uint16_t *gal = nullptr, *qry = nullptr;
  CK(hipMalloc((void**)&gal, gal_elems * 2 + 256));
  CK(hipMalloc((void**)&qry, q_elems * 2 + 256));
  const float scale = 1.0f / std::sqrt((float)d);
  hipLaunchKernelGGL(fill_image, dim3(4096), dim3(256), 0, 0, gal, gal_elems, 0x1234ull, scale, f16);
  hipLaunchKernelGGL(fill_image, dim3(512), dim3(256), 0, 0, qry, q_elems, 0x9876ull, scale, f16);
  QueryState st{};
  CK(hipMalloc((void**)&st.thr, qpad * 4));
  CK(hipMalloc((void**)&st.margin, qpad * 4));
  CK(hipMalloc((void**)&st.thr2, qpad * 4));
  CK(hipMalloc((void**)&st.qflag, qpad * 4));
  CK(hipMalloc((void**)&st.cnt, (size_t)qpad * CNT_STRIDE * 4));
  CK(hipMalloc((void**)&st.flags, 16));
  CK(hipMemset(st.flags, 0, 16));
  st.repair = st.flags + 1;
  CK(hipMemset(st.cnt, 0, (size_t)qpad * CNT_STRIDE * 4));
  st.cap = 12288;
  st.surv = nullptr;                                     // only the FIRST (bootstrap) variant writes survivors directly
  hipLaunchKernelGGL(fill_f32, dim3((qpad + 255) / 256), dim3(256), 0, 0, st.thr, qpad, INFINITY);
  hipLaunchKernelGGL(fill_f32, dim3((nq + 255) / 256), dim3(256), 0, 0, st.thr, nq, thr);
  const uint32_t nseg = gemm_select_grid() * 8, rec_cap = 4096;
  SurvRec* rec = nullptr;
  uint32_t* rec_cnt = nullptr;
  unsigned long long *dbg = nullptr, *chk = nullptr;
  CK(hipMalloc((void**)&rec, (size_t)nseg * rec_cap * sizeof(SurvRec)));
  CK(hipMalloc((void**)&rec_cnt, nseg * 4));
  CK(hipMalloc((void**)&dbg, (size_t)nseg * 8 * 8));
  CK(hipMalloc((void**)&chk, 16));
  CK(hipMemset(dbg, 0, (size_t)nseg * 8 * 8));
  CK(hipDeviceSynchronize());

  ScoreArgs a{};
  a.gal_img = gal;
  a.qry_img = qry;
  a.img_f16 = f16;
  a.nslices = nsl;
  a.tile0 = 0;
  a.ntiles = (int32_t)ntiles;
  a.nqt = qpad / TILE;
  a.n = rows;
  a.nq = nq;
  a.small_batch_kernel = 1;
  a.rec = rec;
  a.rec_cnt = rec_cnt;
  a.rec_cap = rec_cap;
  a.cond = nullptr;
  a.dbg = dbg;
  a.st = st;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flops = 2.0 * nq * (double)rows * d;
  printf("# rows=%lld d=%d q=%d tiles=%lld slices=%d thr=%.4f %s grid=%u\n", (long long)rows, d, nq, (long long)ntiles,
         nsl, thr, f16 ? "f16" : "bf16", nseg / 8);
  for (int r = 0; r < rounds; ++r)
    for (size_t vi = 0; vi < vs.size(); ++vi) {
      auto& v = vs[(vi + (rotate ? (size_t)r : 0)) % vs.size()];
      a.debug = v.debug;
      a.variant = v.variant;
      CK(hipMemset(rec_cnt, 0, nseg * 4));
      launch_gemm_select(a, false, nullptr);   // warm-up of this variant (code + clocks)
      CK(hipDeviceSynchronize());
      for (int k = 0; k < reps; ++k) {
        CK(hipEventRecord(e0, nullptr));
        launch_gemm_select(a, false, nullptr);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        v.ms.push_back(ms);
      }
      CK(hipGetLastError());
      CK(hipMemset(chk, 0, 16));
      hipLaunchKernelGGL(rec_checksum, dim3(nseg), dim3(256), 0, 0, rec, rec_cnt, rec_cap, nseg, chk);
      unsigned long long h[2];
      CK(hipMemcpy(h, chk, 16, hipMemcpyDeviceToHost));
      v.sum = h[0];
      v.cnt = h[1];
      if ((v.debug & 8) && r == rounds - 1) {
        std::vector<unsigned long long> c((size_t)nseg * 8);
        CK(hipMemcpy(c.data(), dbg, c.size() * 8, hipMemcpyDeviceToHost));
        for (int grp = 0; grp < 2; ++grp) {
          double s[5] = {0, 0, 0, 0, 0}, sl = 0;
          for (uint32_t w = 0; w < nseg; ++w)
            if ((int)((w % 8) / 4) == grp && c[(size_t)w * 8 + 5] > 0) {
              for (int j = 0; j < 5; ++j) s[j] += (double)c[(size_t)w * 8 + j];
              sl += (double)c[(size_t)w * 8 + 5];
            }
          printf("#   %s group%d cycles/slice: load=%.0f bar1=%.0f mfma+wait=%.0f bar2=%.0f epi=%.0f total=%.0f\n",
                 v.label.c_str(), grp, s[0] / sl, s[1] / sl, s[2] / sl, s[3] / sl, s[4] / sl,
                 (s[0] + s[1] + s[2] + s[3] + s[4]) / sl);
        }
      }
      if (((v.debug & 16) || (v.debug != 0 && !(v.debug & 8))) && r == rounds - 1) {
        std::vector<unsigned long long> c((size_t)nseg * 8);
        CK(hipMemcpy(c.data(), dbg, c.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> mhz, cps;
        for (uint32_t w = 0; w < nseg; ++w)
          if (c[(size_t)w * 8 + 7] > 0) {
            mhz.push_back((double)c[(size_t)w * 8 + 6] / (double)c[(size_t)w * 8 + 7] * 100.0);
            if (!(v.debug & 8) && c[(size_t)w * 8 + 5] > 0) cps.push_back((double)c[(size_t)w * 8 + 6] / (double)c[(size_t)w * 8 + 5]);
          }
        if (v.debug & 2048) {
          for (int grp = 0; grp < 2; ++grp) {
            double f[4] = {0, 0, 0, 0}, hits = 0, rounds = 0, tiles = 0;
            for (uint32_t w = 0; w < nseg; ++w)
              if ((int)((w % 8) / 4) == grp && c[(size_t)w * 8 + 7] > 0) {
                for (int j = 0; j < 4; ++j) f[j] += (double)c[(size_t)w * 8 + j];
                const unsigned long long pk = c[(size_t)w * 8 + 4];
                hits += (double)(pk >> 40);
                rounds += (double)((pk >> 20) & 0xFFFFF);
                tiles += (double)(pk & 0xFFFFF);
              }
            printf("#   %s group%d filter per tile: decide %.0f  write burst %.0f  read wait %.0f  emit %.0f cycles; hits %.1f, rounds %.2f\n",
                   v.label.c_str(), grp, f[0] / tiles, f[1] / tiles, f[2] / tiles, f[3] / tiles, hits / tiles, rounds / tiles);
          }
        }
        if (v.variant >= 1 && !(v.debug & 8) && !(v.debug & 2048)) {
          // per XCC: loop duration (us) of its workgroups and their clock
          for (int x = 0; x < 8; ++x) {
            std::vector<double> dur, mh;
            int nb = 0;
            for (uint32_t w = 0; w < nseg; w += 8)
              if (c[(size_t)w * 8 + 7] > 0 && (int)c[(size_t)w * 8 + 4] == x) {
                dur.push_back((double)c[(size_t)w * 8 + 7] * 0.01);
                mh.push_back((double)c[(size_t)w * 8 + 6] / (double)c[(size_t)w * 8 + 7] * 100.0);
                nb += ((w / 8) & 7) == (uint32_t)x ? 1 : 0;
              }
            if (dur.empty()) continue;
            std::sort(dur.begin(), dur.end());
            std::sort(mh.begin(), mh.end());
            printf("#     xcc %d: %zu workgroups (%d with blockIdx %% 8 == xcc)  loop us min %.0f med %.0f max %.0f   clock med %.0f MHz\n",
                   x, dur.size(), nb, dur.front(), dur[dur.size() / 2], dur.back(), mh[mh.size() / 2]);
          }
        }
        if (!mhz.empty()) {
          std::sort(mhz.begin(), mhz.end());
          std::sort(cps.begin(), cps.end());
          printf("#   %s in-kernel clock: median %.0f MHz (min %.0f, max %.0f); loop cycles per slice: median %.0f\n",
                 v.label.c_str(), mhz[mhz.size() / 2], mhz.front(), mhz.back(), cps.empty() ? 0.0 : cps[cps.size() / 2]);
        }
      }
    }
  for (auto& v : vs) {
    std::vector<float> m = v.ms;
    std::sort(m.begin(), m.end());
    const double med = m[m.size() / 2], mn = m.front();
    printf("%-14s debug=%-3d var=%-2d  median %.4f ms = %7.1f TF   min %.4f ms = %7.1f TF   records=%llu sum=%016llx%s\n",
           v.label.c_str(), v.debug, v.variant, med, flops / med / 1e9, mn, flops / mn / 1e9, v.cnt, v.sum,
           (v.cnt == vs[0].cnt && v.sum == vs[0].sum) ? "" : "   != first variant");
  }
  return 0;
}
