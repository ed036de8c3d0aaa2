// Read-bandwidth ceilings of the box for the two access patterns the LFA kernels use:
//   stream : every lane reads consecutive 16-byte pieces (fully coalesced)
//   rows   : lane i of a wavefront owns row i (512 B) and reads it in 16-byte pieces, two
//            lanes per row - the pattern of the GEMM's A-fragment loads
// at an Infinity-Cache-sized and an HBM-sized footprint, U loads in flight per lane.
// Build: hipcc -O3 --offload-arch=gfx950 tools/readbw.hip -o tools/readbw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool ROWS>
__global__ __launch_bounds__(512) void read_kernel(const u32x4 *__restrict__ src, size_t n16,
                                                    unsigned *sink) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  if (ROWS) {
    // a wavefront takes 32 rows of 512 B (= 32 pieces each); lane (i, h) reads pieces 2q + h
    const size_t wave = tid >> 6, nwaves = nthreads >> 6;
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const size_t nblocks = n16 / (32 * 32);
    for (size_t b = wave; b < nblocks; b += nwaves) {
      const u32x4 *row = src + (b * 32 + i) * 32 + h;
      u32x4 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = __builtin_nontemporal_load(row + 2 * q);
#pragma unroll
      for (int q = 0; q < 16; ++q) acc ^= v[q];
    }
  } else {
    for (size_t k = tid; k + (U - 1) * nthreads < n16; k += U * nthreads) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[k + u * nthreads];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool ROWS>
static int run(const char *name, const u32x4 *src, size_t bytes, unsigned *sink, int blocks) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const size_t n16 = bytes / 16;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((read_kernel<U, ROWS>), dim3(blocks), dim3(512), 0, 0, src, n16, sink);
  CK(hipEventRecord(a, 0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((read_kernel<U, ROWS>), dim3(blocks), dim3(512), 0, 0, src, n16, sink);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-10s U=%2d blocks=%5d %6zu MB  %.2f TB/s\n", name, U, blocks, bytes >> 20,
         (double)bytes * reps / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  const size_t big = 2048ull << 20;
  u32x4 *src;
  unsigned *sink;
  CK(hipMalloc(&src, big));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(src, 1, big));
  CK(hipMemset(sink, 0, 4));
  const size_t sizes[] = {64ull << 20, 115ull << 20, big};
  for (size_t s : sizes) {
    for (int blocks : {256, 512, 1024, 2048}) {
      if (run<4, false>("stream", src, s, sink, blocks)) return 1;
      if (run<8, false>("stream", src, s, sink, blocks)) return 1;
      if (run<16, false>("stream", src, s, sink, blocks)) return 1;
    }
    for (int blocks : {256, 512, 1024})
      if (run<16, true>("rows", src, s, sink, blocks)) return 1;
  }
  return 0;
}
