// unaligned_load.hip -- are 16-byte vector loads at BYTE-aligned addresses legal and exact on gfx950 (global_load_dwordx4 and
// raw buffer loads with hardware range checking)?  Every lane loads 16 bytes at base + 64 * lane + shift for shift = 0..15 and
// the result is compared with a byte-wise copy; the buffer form is also asked for offsets before and behind its range.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/unaligned_load.hip -o /tmp/ul && /tmp/ul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_global(const unsigned char *buf, u32x4 *out, int shift) {
  const int lane = threadIdx.x;
  out[lane] = *reinterpret_cast<const u32x4 *>(buf + 64 * lane + shift);
}
__global__ void k_buffer(const unsigned char *buf, int bytes, u32x4 *out, int shift, int base_off) {
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, bytes, 0x00020000);
  out[lane] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + 64 * lane + shift, 0, 0);
}

int main() {
  const int N = 64 * 64 + 64;
  std::vector<unsigned char> h(N);
  for (int i = 0; i < N; i++) h[i] = (unsigned char)(i * 7 + (i >> 8) * 13 + 1);
  unsigned char *d; u32x4 *o;
  CK(hipMalloc(&d, N)); CK(hipMalloc(&o, 64 * 16));
  CK(hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice));
  std::vector<unsigned char> r(64 * 16);
  int bad = 0;
  for (int shift = 0; shift < 16; shift++) {
    hipLaunchKernelGGL(k_global, dim3(1), dim3(64), 0, 0, d, o, shift);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; l++) for (int b = 0; b < 16; b++) if (r[16 * l + b] != h[64 * l + shift + b]) bad++;
    hipLaunchKernelGGL(k_buffer, dim3(1), dim3(64), 0, 0, d, 64 * 64, o, shift, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; l++) for (int b = 0; b < 16; b++) {
      const int a = 64 * l + shift + b;
      const unsigned char want = a < 64 * 64 ? h[a] : 0;
      if (r[16 * l + b] != want) { if (bad < 8) printf("buffer shift %d lane %d byte %d: got %u want %u\n", shift, l, b, r[16 * l + b], want); bad++; }
    }
  }
  printf("byte-aligned 16-byte loads, global and buffer form: %s (%d mismatches)\n", bad ? "MISMATCH" : "exact", bad);
  // range checking: offsets before the base (negative) and behind the end
  for (int base_off : {-40, 64 * 64 - 20}) {
    hipLaunchKernelGGL(k_buffer, dim3(1), dim3(64), 0, 0, d, 64 * 64, o, 0, base_off);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost));
    printf("buffer load at offset %d (lane 0): ", base_off);
    for (int b = 0; b < 16; b++) printf("%u ", r[b]);
    printf("| expected in range: ");
    for (int b = 0; b < 16; b++) { const int a = base_off + b; printf("%d ", (a >= 0 && a < 64 * 64) ? h[a] : -1); }
    printf("\n");
  }
  return 0;
}
