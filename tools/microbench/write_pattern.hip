// write_pattern.hip -- what does the ORDER in which a kernel writes a batch of row-major u16 matrices (1024 pairs x 1000
// rows x 2000 bytes, the all-pairs Hamming output: 2.05 GB) cost on gfx950?  Every mode writes every byte exactly once with
// 16-byte (or 8-byte) stores and does nothing else; only the assignment of addresses to wavefronts and instructions changes.
//   0 linear       a workgroup streams a contiguous 128-row block: 1 KB contiguous per wave-instruction
//   1 rows8        k_hamming<4>'s order: lane = 4 columns (8 bytes), wave = 512 bytes of a row, workgroup = a whole row, row by row
//   2 seg256       the first matrix-core form: a wavefront owns 32 rows, per step 8 instructions of 4 rows x 256 bytes at
//                  column offset 256 c (16-byte aligned only: the row pitch is 2000)
//   3 seg256a      the same with every row's window shifted to its own 128-byte line boundaries (whole lines only)
//   4 seg512a      4 rows x 512... a wavefront owns 32 rows, windows of 512 bytes, line aligned (2 rows per instruction)
//   5 rowblock32   a workgroup streams 32 whole rows (64000 contiguous bytes), then the next 32: 1 KB per wave-instruction
//   6 seg128a      windows of 128 bytes (one line per row, 8 rows per instruction), line aligned
//   7 seg256b      as 3, windows on 256-byte boundaries
//   8 seg512b      as 4, windows on 512-byte boundaries
//   9 seg256b2     as 7, and a row's two partial pieces (head and tail, which share 256-byte blocks with the neighbouring
//                  rows) are both written in the last step
//  10 seg256a2     256-byte windows on 128-byte boundaries, heads deferred to the last step
//  11 seg128a2     128-byte windows on 128-byte boundaries (8 rows per instruction), heads deferred
//  12 gridstride   the whole grid sweeps memory front to back, 16 bytes per thread per trip (a fill kernel)
//  13 linear_nt    as 0 with non-temporal stores
// Prints ms and TB/s per mode.   hipcc -O3 --offload-arch=gfx950 tools/microbench/write_pattern.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kPairs = 1024, kRows = 1000;
#ifndef PITCH
#define PITCH 2000
#endif
constexpr long long kPairBytes = (long long)kRows * PITCH;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_write(unsigned char *D, unsigned val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char *base = D + (long long)blockIdx.z * kPairBytes;
  const int i0 = blockIdx.x * 128;  // 128 rows per workgroup
  const u32x4 v = {val, val + 1, val + 2, val + 3};
  if (MODE == 0) {
    const long long lo = (long long)i0 * PITCH, hi = std::min<long long>((long long)(i0 + 128) * PITCH, kPairBytes);
    for (long long o = lo + threadIdx.x * 16; o < hi; o += 4096) *reinterpret_cast<u32x4 *>(base + o) = v;
  } else if (MODE == 1) {
    for (int r = 0; r < 128 && i0 + r < kRows; r++) {
      const int o = threadIdx.x * 8;
      if (o < PITCH) *reinterpret_cast<u32x2 *>(base + (long long)(i0 + r) * PITCH + o) = u32x2{val, val + 1};
    }
  } else if (MODE == 7 || MODE == 9) {
    const int r0 = i0 + wave * 32;
    for (int c = 0; c < 9; c++) {
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int r = r0 + 4 * it + (lane >> 4);
        unsigned char *rowp = base + (long long)r * PITCH;
        const int si = (int)(reinterpret_cast<uintptr_t>(rowp) & 255);
        const int off = c * 256 - si + 16 * (lane & 15);
        bool ok = r < kRows && off >= 0 && off < PITCH;
        if (MODE == 9 && c == 0 && si != 0) ok = false;  // the head piece waits for the last step
        if (ok) *reinterpret_cast<u32x4 *>(rowp + off) = v;
        if (MODE == 9 && c == 8) {
          const int o2 = 16 * (lane & 15);
          if (r < kRows && si != 0 && o2 < 256 - si) *reinterpret_cast<u32x4 *>(rowp + o2) = v;
        }
      }
      __builtin_amdgcn_s_sleep(8);
    }
  } else if (MODE == 10 || MODE == 11) {  // 10: 256-byte windows on 128-byte boundaries, 11: 128-byte windows; heads deferred
    constexpr int W = MODE == 10 ? 256 : 128, LPR = W / 16, RPI = 64 / LPR, NIT = 32 / RPI, NC = (PITCH + 127) / W + 1;
    const int r0 = i0 + wave * 32;
    for (int c = 0; c < NC; c++) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int r = r0 + RPI * it + lane / LPR;
        unsigned char *rowp = base + (long long)r * PITCH;
        const int si = (int)(reinterpret_cast<uintptr_t>(rowp) & 127);
        const int off = c * W - si + 16 * (lane % LPR);
        bool ok = r < kRows && off >= 0 && off < PITCH;
        if (c == 0 && si != 0 && off < 128 - si) ok = false;  // the head piece waits for the last step
        if (ok) *reinterpret_cast<u32x4 *>(rowp + off) = v;
        if (c == NC - 1) {
          const int o2 = 16 * (lane % LPR);
          if (r < kRows && si != 0 && o2 < 128 - si) *reinterpret_cast<u32x4 *>(rowp + o2) = v;
        }
      }
      __builtin_amdgcn_s_sleep(MODE == 10 ? 8 : 4);
    }
  } else if (MODE == 12) {  // the whole grid sweeps memory front to back (what a fill kernel does)
    const long long total = (long long)kPairs * kPairBytes, nthreads = (long long)gridDim.x * gridDim.z * 256;
    const long long tid0 = ((long long)blockIdx.z * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    for (long long o = tid0 * 16; o < total; o += nthreads * 16) *reinterpret_cast<u32x4 *>(D + o) = v;
  } else if (MODE == 13) {  // as 0 with non-temporal stores
    const long long lo = (long long)i0 * PITCH, hi = std::min<long long>((long long)(i0 + 128) * PITCH, kPairBytes);
    for (long long o = lo + threadIdx.x * 16; o < hi; o += 4096) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(base + o));
  } else if (MODE == 8) {
    const int r0 = i0 + wave * 32;
    for (int c = 0; c < 5; c++) {
#pragma unroll
      for (int it = 0; it < 16; it++) {
        const int r = r0 + 2 * it + (lane >> 5);
        unsigned char *rowp = base + (long long)r * PITCH;
        const int si = (int)(reinterpret_cast<uintptr_t>(rowp) & 511);
        const int off = c * 512 - si + 16 * (lane & 31);
        if (r < kRows && off >= 0 && off < PITCH) *reinterpret_cast<u32x4 *>(rowp + off) = v;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  } else if (MODE == 2 || MODE == 3) {
    const int r0 = i0 + wave * 32;
    for (int c = 0; c < 9; c++) {
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int r = r0 + 4 * it + (lane >> 4);
        unsigned char *rowp = base + (long long)r * PITCH;
        const int si = MODE == 3 ? (int)(reinterpret_cast<uintptr_t>(rowp) & 127) : 0;
        const int off = c * 256 - si + 16 * (lane & 15);
        if (r < kRows && off >= 0 && off < PITCH && (MODE == 3 || c < 8)) *reinterpret_cast<u32x4 *>(rowp + off) = v;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  } else if (MODE == 4) {
    const int r0 = i0 + wave * 32;
    for (int c = 0; c < 5; c++) {
#pragma unroll
      for (int it = 0; it < 16; it++) {
        const int r = r0 + 2 * it + (lane >> 5);
        unsigned char *rowp = base + (long long)r * PITCH;
        const int si = (int)(reinterpret_cast<uintptr_t>(rowp) & 127);
        const int off = c * 512 - si + 16 * (lane & 31);
        if (r < kRows && off >= 0 && off < PITCH) *reinterpret_cast<u32x4 *>(rowp + off) = v;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  } else if (MODE == 5) {
    for (int b = 0; b < 4; b++) {
      const long long lo = (long long)(i0 + 32 * b) * PITCH, hi = std::min<long long>((long long)(i0 + 32 * b + 32) * PITCH, kPairBytes);
      for (long long o = lo + threadIdx.x * 16; o < hi; o += 4096) *reinterpret_cast<u32x4 *>(base + o) = v;
      __builtin_amdgcn_s_sleep(8);
    }
  } else if (MODE == 6) {
    const int r0 = i0 + wave * 32;
    for (int c = 0; c < 17; c++) {
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int r = r0 + 8 * it + (lane >> 3);
        unsigned char *rowp = base + (long long)r * PITCH;
        const int si = (int)(reinterpret_cast<uintptr_t>(rowp) & 127);
        const int off = c * 128 - si + 16 * (lane & 7);
        if (r < kRows && off >= 0 && off < PITCH) *reinterpret_cast<u32x4 *>(rowp + off) = v;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
}

template <int MODE>
void run(const char *name, unsigned char *D) {
  dim3 grid((kRows + 127) / 128, 1, kPairs);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int rep = 0; rep < 8; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_write<MODE>, grid, dim3(256), 0, 0, D, (unsigned)rep);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms);
  }
  std::sort(ts.begin() + 2, ts.end());
  const float m = ts[2 + 3];
  printf("%-12s %.3f ms  %.2f TB/s\n", name, m, kPairs * (double)kPairBytes / m / 1e9);
}

int main() {
  unsigned char *D;
  CK(hipMalloc(&D, kPairs * kPairBytes + 4096));
  printf("pitch %d\n", PITCH);
  run<0>("linear", D); run<1>("rows8", D); run<2>("seg256", D); run<3>("seg256a", D); run<4>("seg512a", D);
  run<5>("rowblock32", D); run<6>("seg128a", D); run<7>("seg256b", D); run<8>("seg512b", D); run<9>("seg256b2", D); run<10>("seg256a2", D); run<11>("seg128a2", D); run<12>("gridstride", D); run<13>("linear_nt", D);
  run<0>("linear", D);
  return 0;
}
