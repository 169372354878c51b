// concurrency.hip -- do two kernels launched on two streams share the chip on gfx950, or does the second one wait for the
// first?  A = a pure VALU spin (no memory traffic; its workgroups are capped at W per CU by a dynamic LDS request), B = a pure
// write stream (fill of 2 GB).  Prints A alone, B alone, both (A first / B first) for W = 8, 4, 2, 1 workgroups of A per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/concurrency.hip -o /tmp/cc && /tmp/cc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_spin(unsigned *out, int iters) {
  extern __shared__ unsigned char dyn[];
  unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { a = a * 1664525u + b; b = b * 22695477u + c; c = c * 1103515245u + d; d = d * 69069u + a; }
  }
  if ((a ^ b ^ c ^ d) == 0x1234567u) out[0] = a + dyn[0];
}

__global__ __launch_bounds__(256) void k_fill(u32x4 *D, size_t n16, unsigned val) {
  const u32x4 v = {val, val + 1, val + 2, val + 3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) D[i] = v;
}

int main() {
  const size_t bytes = 2048ull << 20;
  u32x4 *D; unsigned *out;
  CK(hipMalloc(&D, bytes)); CK(hipMalloc(&out, 64));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int W : {8, 4, 2, 1}) {
    const size_t lds = W == 8 ? 0 : (size_t)(160 * 1024 / W) - 512;
    const int blocks = 256 * 8 * 4, iters = 180 * 8 / W / 2;  // the same total time per W: fewer resident waves issue proportionally faster only up to the issue rate
    auto A = [&] { hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), lds, s1, out, iters); };
    auto B = [&] { hipLaunchKernelGGL(k_fill, dim3(256 * 16), dim3(256), 0, s2, D, bytes / 16, 7u); };
    auto timed = [&](int mode) {
      double best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        if (mode == 0) A();
        else if (mode == 1) B();
        else if (mode == 2) { A(); B(); }
        else { B(); A(); }
        CK(hipDeviceSynchronize());
        best = std::min(best, now() - t0);
      }
      return best;
    };
    timed(2);
    const double a = timed(0), b = timed(1), ab = timed(2), ba = timed(3);
    printf("A at %d workgroups per CU: A %.3f ms, B %.3f ms, A then B %.3f ms, B then A %.3f ms (sum %.3f)\n", W, a, b, ab, ba, a + b);
  }
  return 0;
}
