// Rate of float atomic adds in the shape the backward's scatter uses: 32 lanes add one
// 128-byte run at a data-dependent row of an 80 MB buffer (the size of one chunk's grad_Y).
//   agent      : unsafeAtomicAdd as shipped (device scope)
//   workgroup  : the same instruction at workgroup scope (no sc1: executed in the XCD's L2)
//   *_own      : rows drawn only from the eighth of the buffer that belongs to the XCD the
//                wavefront runs on (HW_REG_XCC_ID), i.e. no address is touched by two XCDs
// Each run checks the buffer's sum afterwards: a scope that loses updates shows up there.
// Build: hipcc -O3 --offload-arch=gfx950 tools/atomicbw.hip -o tools/atomicbw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <bool WG_SCOPE, bool OWN>
__global__ __launch_bounds__(256) void add_kernel(float *buf, unsigned nrows, int iters,
                                                  unsigned *xcc_seen) {
  const unsigned lane = threadIdx.x & 63, l = lane & 31;
  const unsigned half = (blockIdx.x * 256 + threadIdx.x) >> 5;     // one run per half wavefront
  const unsigned xcc = xcc_id();
  if (lane == 0 && threadIdx.x == 0) atomicOr(xcc_seen + (blockIdx.x & 7), 1u << xcc);
  const unsigned per = nrows / 8;
  for (int it = 0; it < iters; ++it) {
    unsigned r = hash32(half * 7919u + it * 104729u + 1u);
    r = OWN ? xcc * per + r % per : r % nrows;
    float *p = buf + (size_t)r * 32 + l;
    if (WG_SCOPE)
      __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      unsafeAtomicAdd(p, 1.0f);
  }
}

template <bool WG_SCOPE, bool OWN>
static int run(const char *name, float *buf, size_t bytes, unsigned *seen) {
  const unsigned nrows = (unsigned)(bytes / 128);
  const int blocks = 256 * 16, iters = 64;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e30f;
  double sum = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(buf, 0, bytes));
    CK(hipMemset(seen, 0, 32));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((add_kernel<WG_SCOPE, OWN>), dim3(blocks), dim3(256), 0, 0, buf, nrows, iters, seen);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  std::vector<float> h(bytes / 4);
  CK(hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost));
  for (float v : h) sum += v;
  unsigned hs[8];
  CK(hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost));
  const double adds = (double)blocks * 256 * iters;
  printf("%-16s %8.3f ms  %7.1f GB/s of atomic bytes   sum %.0f of %.0f %s   xcc masks by blockIdx%%8:",
         name, best, adds * 4 / best / 1e6, sum, adds, sum == adds ? "ok" : "LOST UPDATES");
  for (int k = 0; k < 8; ++k) printf(" %02x", hs[k]);
  printf("\n");
  return 0;
}

int main() {
  const size_t bytes = 80u << 20;
  float *buf;
  unsigned *seen;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&seen, 32));
  if (run<false, false>("agent", buf, bytes, seen)) return 1;
  if (run<true, false>("workgroup", buf, bytes, seen)) return 1;
  if (run<false, true>("agent_own", buf, bytes, seen)) return 1;
  if (run<true, true>("workgroup_own", buf, bytes, seen)) return 1;
  return 0;
}
