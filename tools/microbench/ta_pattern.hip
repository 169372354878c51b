// ta_pattern.hip -- what does a 16-byte-per-lane vector load cost the texture-address path on gfx950 (MI355X) as a
// function of HOW the 64 lanes' addresses are laid out?  Every wave issues the same number of global_load_dwordx4 from
// an L2-resident buffer (32 MB); only the address pattern changes:
//   0 contiguous   lanes 0..63 read one aligned 1 KB run
//   1 quads        16 scattered, 64-byte-aligned segments, 4 consecutive lanes each
//   2 pairs        32 scattered, 32-byte-aligned pieces, 2 consecutive lanes each (a 256-bit descriptor per lane pair)
//   3 triples      21 scattered rows of 48 contiguous bytes at a 16-byte-aligned start (k_describe's window rows)
//   4 triples64    the same, starts 64-byte aligned (the 48 bytes never straddle a 64-byte segment)
//   5 scattered    every lane its own 16-byte chunk somewhere (k_ba_pairs: lane = couple, chunk q of its block)
//   6 blocks9      7 scattered 144-byte blocks, 9 consecutive lanes each (cooperative load of W blocks)
//   7 octets       8 scattered, 128-byte-aligned lines, 8 consecutive lanes each
// Prints ns per wave-instruction per CU (all CUs busy, 8 waves per SIMD) and the implied lanes per clock.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/ta_pattern.hip -o /tmp/ta_pattern && /tmp/ta_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef VO_TA_BYTES
#define VO_TA_BYTES (2u << 20)
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLoads = 256;               // loads per wave
constexpr size_t kBytes = VO_TA_BYTES;      // buffer (2 MB: fits every XCD's 4 MB L2; 32 MB: memory side)

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_pat(const unsigned char *buf, unsigned *out, unsigned seed) {
  const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < kLoads; i += 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned key = seed + wave * 7919u + (unsigned)(i + u) * 104729u;
      size_t off;
      if (MODE == 0) off = (size_t)(hash(key) % (kBytes / 1024)) * 1024 + lane * 16;
      else if (MODE == 1) off = (size_t)(hash(key + (lane >> 2)) % (kBytes / 64)) * 64 + (lane & 3) * 16;
      else if (MODE == 2) off = (size_t)(hash(key + (lane >> 1)) % (kBytes / 32)) * 32 + (lane & 1) * 16;
      else if (MODE == 3) off = (size_t)(hash(key + lane / 3) % (kBytes / 16 - 8)) * 16 + (lane % 3) * 16;
      else if (MODE == 4) off = (size_t)(hash(key + lane / 3) % (kBytes / 64)) * 64 + (lane % 3) * 16;
      else if (MODE == 5) off = (size_t)(hash(key + lane) % (kBytes / 16)) * 16;
      else if (MODE == 6) off = (size_t)(hash(key + lane / 9) % (kBytes / 16 - 16)) * 16 + (lane % 9) * 16;
      else off = (size_t)(hash(key + (lane >> 3)) % (kBytes / 128)) * 128 + (lane & 7) * 16;
      v[u] = *reinterpret_cast<const u32x4 *>(buf + off);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

template <int MODE>
void run(const char *name, const unsigned char *buf, unsigned *out, int n_cu) {
  const int blocks = n_cu * 8;  // 8 x 256 threads per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pat<MODE>, dim3(blocks), dim3(256), 0, 0, buf, out, 1234u + rep);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double instr_per_cu = 32.0 * kLoads;  // 32 waves per CU
  const double ns = best * 1e6 / instr_per_cu;
  printf("%-12s %8.3f ms  %7.1f ns per wave-instruction per CU = %5.1f cycles at 2.4 GHz = %5.2f lanes/clk  (%6.2f TB/s of requested bytes)\n",
         name, best, ns, ns * 2.4, 64.0 / (ns * 2.4), (double)n_cu * instr_per_cu * 1024 / (best * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  unsigned char *buf; unsigned *out;
  CK(hipMalloc(&buf, kBytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, kBytes)); CK(hipMemset(out, 0, 64));
  const int n = prop.multiProcessorCount;
  printf("%s, %d CUs; %d loads of 16 B per lane per wave, 32 waves per CU, %zu MB buffer\n", prop.name, n, kLoads, kBytes >> 20);
  run<0>("contiguous", buf, out, n);
  run<1>("quads", buf, out, n);
  run<7>("octets", buf, out, n);
  run<2>("pairs", buf, out, n);
  run<3>("triples", buf, out, n);
  run<4>("triples64", buf, out, n);
  run<6>("blocks9", buf, out, n);
  run<5>("scattered", buf, out, n);
  return 0;
}
