// How fast can a 12 GB file that sits in the page cache reach the device?  (The prepared-gallery loader of csrc/api_file.hip went
// through a mapping until round 5: 25 GB/s against 55 GB/s of pinned H2D; it is variant d now.)  Same file, same device buffer:
//   a  mmap(MAP_POPULATE) of everything, then hipMemcpy in 256 MiB pieces            (the loader as it was)
//   b  mmap without populate, T threads populate pieces ahead (MADV_POPULATE_READ), hipMemcpy of a piece when it is populated
//   c  mmap without populate, hipMemcpy straight away (the runtime faults the pages in)
//   d  T threads pread() pieces into a ring of pinned buffers, hipMemcpyAsync per piece   (copy_file_to_dev_parallel)
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 scripts/mapload_probe.hip -o build/mapload_probe -lpthread
//   mapload_probe <file> [GiB to write if the file does not exist]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
#define CK(e)                                                                            \
  do {                                                                                   \
    hipError_t _e = (e);                                                                 \
    if (_e != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/tmp/mapload_probe.bin";
  const size_t want = (size_t)(argc > 2 ? atof(argv[2]) : 11.5) << 30;
  struct stat st;
  if (stat(path, &st) != 0 || (size_t)st.st_size < want) {
    FILE* f = fopen(path, "wb");
    std::vector<char> blk(64 << 20);
    for (size_t i = 0; i < blk.size(); ++i) blk[i] = (char)(i * 131 + (i >> 12));
    for (size_t o = 0; o < want; o += blk.size()) fwrite(blk.data(), 1, blk.size(), f);
    fclose(f);
    stat(path, &st);
  }
  const size_t bytes = (size_t)st.st_size;
  const size_t piece = (size_t)256 << 20;
  const size_t npieces = (bytes + piece - 1) / piece;
  char* dev;
  CK(hipMalloc((void**)&dev, bytes));
  CK(hipMemset(dev, 0, bytes));
  CK(hipDeviceSynchronize());
  const int fd = open(path, O_RDONLY);
  printf("file %.2f GB\n", bytes / 1e9);
  for (int rep = 0; rep < 2; ++rep) {
    {  // a
      const double t0 = now();
      char* m = (char*)mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
      const double t1 = now();
      for (size_t o = 0; o < bytes; o += piece) CK(hipMemcpy(dev + o, m + o, std::min(piece, bytes - o), hipMemcpyHostToDevice));
      const double t2 = now();
      munmap(m, bytes);
      const double t3 = now();
      printf("a populate-all: map %.3f s, copy %.3f s (%.1f GB/s), unmap %.3f s, total %.3f s = %.1f GB/s\n", t1 - t0, t2 - t1,
             bytes / 1e9 / (t2 - t1), t3 - t2, t3 - t0, bytes / 1e9 / (t3 - t0));
    }
    for (int T : {2, 4, 8}) {  // b
      const double t0 = now();
      char* m = (char*)mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
      std::vector<std::atomic<int>> done(npieces);
      for (auto& d : done) d = 0;
      std::atomic<size_t> next{0};
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&] {
          for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= npieces) return;
            const size_t o = i * piece, len = std::min(piece, bytes - o);
            if (madvise(m + o, len, MADV_POPULATE_READ) != 0) {
              volatile char sink = 0;
              for (size_t p = 0; p < len; p += 4096) sink += m[o + p];
            }
            done[i].store(1, std::memory_order_release);
          }
        });
      for (size_t i = 0; i < npieces; ++i) {
        while (!done[i].load(std::memory_order_acquire)) std::this_thread::yield();
        const size_t o = i * piece;
        CK(hipMemcpy(dev + o, m + o, std::min(piece, bytes - o), hipMemcpyHostToDevice));
      }
      for (auto& t : th) t.join();
      const double t2 = now();
      munmap(m, bytes);
      const double t3 = now();
      printf("b populate ahead, %d threads: total %.3f s = %.1f GB/s (unmap %.3f s)\n", T, t3 - t0, bytes / 1e9 / (t3 - t0), t3 - t2);
    }
    {  // c
      const double t0 = now();
      char* m = (char*)mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
      for (size_t o = 0; o < bytes; o += piece) CK(hipMemcpy(dev + o, m + o, std::min(piece, bytes - o), hipMemcpyHostToDevice));
      const double t2 = now();
      munmap(m, bytes);
      printf("c no populate: total %.3f s = %.1f GB/s\n", now() - t0, bytes / 1e9 / (now() - t0));
      (void)t2;
    }
    for (int T : {4, 8, 12}) {  // d
      const size_t chunk = (size_t)32 << 20;
      const int ring = 2 * T;
      std::vector<char*> buf(ring);
      std::vector<hipEvent_t> ev(ring);
      hipStream_t s;
      CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      for (int i = 0; i < ring; ++i) {
        CK(hipHostMalloc((void**)&buf[i], chunk, hipHostMallocDefault));
        CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
      }
      const size_t nch = (bytes + chunk - 1) / chunk;
      const double t0 = now();
      std::vector<std::atomic<int>> filled(nch), freed(nch);      // chunk i read into its slot / its copy finished
      for (auto& d : filled) d = 0;
      for (auto& d : freed) d = 0;
      std::atomic<size_t> next{0};
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&] {
          for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nch) return;
            if (i >= (size_t)ring)
              while (!freed[i - ring].load(std::memory_order_acquire)) std::this_thread::yield();
            const size_t o = i * chunk, len = std::min(chunk, bytes - o);
            size_t got = 0;
            while (got < len) {
              const ssize_t r = pread(fd, buf[i % ring] + got, len - got, (off_t)(o + got));
              if (r <= 0) exit(3);
              got += (size_t)r;
            }
            filled[i].store(1, std::memory_order_release);
          }
        });
      size_t waited = 0;
      for (size_t i = 0; i < nch; ++i) {
        while (!filled[i].load(std::memory_order_acquire)) std::this_thread::yield();
        const size_t o = i * chunk, len = std::min(chunk, bytes - o);
        CK(hipMemcpyAsync(dev + o, buf[i % ring], len, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(ev[i % ring], s));
        // release slots whose copies have finished (in order)
        while (waited <= i && (i - waited >= (size_t)ring / 2 || i + 1 == nch)) {
          CK(hipEventSynchronize(ev[waited % ring]));
          freed[waited].store(1, std::memory_order_release);
          ++waited;
        }
      }
      for (auto& t : th) t.join();
      CK(hipStreamSynchronize(s));
      const double t1 = now();
      printf("d pread by %d threads into %d pinned 32 MiB buffers: %.3f s = %.1f GB/s\n", T, ring, t1 - t0, bytes / 1e9 / (t1 - t0));
      for (int i = 0; i < ring; ++i) {
        CK(hipHostFree(buf[i]));
        CK(hipEventDestroy(ev[i]));
      }
      CK(hipStreamDestroy(s));
    }
  }
  close(fd);
  unlink(path);
  return 0;
}
