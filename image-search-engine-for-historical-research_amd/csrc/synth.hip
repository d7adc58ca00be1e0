// Device twin of synth.synth_rows (bit-identical float32 values; integer arithmetic + one exact scaling).
#include "common.h"
#include "kernels.h"

namespace mi {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_fill_kernel(float* __restrict__ dst, uint64_t seed_off, int64_t row0,
                                                         int64_t total, int32_t d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t ctr = (uint64_t)row0 * (uint64_t)d + (uint64_t)i;
    const uint64_t h = splitmix64(ctr + seed_off);
    const int s = (int)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
    dst[i] = (float)s * 3.0517578125e-05f;   // 2^-15, exact
  }
}

void launch_synth_fill(float* dst, uint64_t seed, int64_t row0, int64_t nrows, int32_t d, hipStream_t stream) {
  const int64_t total = nrows * (int64_t)d;
  if (total <= 0) return;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst,
                     seed * 0xD1342543DE82EF95ull, row0, total, d);
}

}  // namespace mi
