#!/usr/bin/env python3
"""Generate tools/microbench/valu_survey.hip: wave64 issue rate of many VALU opcodes on gfx950 at 2 and 8 waves
per SIMD (one 512-thread block per CU / two 1024-thread blocks), wall-clock G wave-instr/s/SIMD and cycles."""
OPS = [
    # name, asm template ({a}=accumulator in/out, {b},{c} inputs)
    ("v_add_u32", "v_add_u32 {a}, {a}, {b}"),
    ("v_sub_u32", "v_sub_u32 {a}, {a}, {b}"),
    ("v_add3_u32", "v_add3_u32 {a}, {a}, {b}, {c}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {a}, {a}, 1, {b}"),
    ("v_and_b32", "v_and_b32 {a}, {a}, {b}"),
    ("v_or_b32", "v_or_b32 {a}, {a}, {b}"),
    ("v_xor_b32", "v_xor_b32 {a}, {a}, {b}"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {a}, 1, {a}"),
    ("v_lshrrev_b32", "v_lshrrev_b32 {a}, 1, {a}"),
    ("v_mov_b32", "v_mov_b32 {a}, {b}"),
    ("v_cndmask_b32", "v_cndmask_b32 {a}, {a}, {b}, vcc"),
    ("v_max_u32", "v_max_u32 {a}, {a}, {b}"),
    ("v_min_i32", "v_min_i32 {a}, {a}, {b}"),
    ("v_max_u16", "v_max_u16 {a}, {a}, {b}"),
    ("v_max_f32", "v_max_f32 {a}, {a}, {b}"),
    ("v_min_f32", "v_min_f32 {a}, {a}, {b}"),
    ("v_add_f32", "v_add_f32 {a}, {a}, {b}"),
    ("v_sub_f32", "v_sub_f32 {a}, {a}, {b}"),
    ("v_mul_f32", "v_mul_f32 {a}, {a}, {b}"),
    ("v_fma_f32", "v_fma_f32 {a}, {a}, {b}, {c}"),
    ("v_fmac_f32", "v_fmac_f32 {a}, {b}, {c}"),
    ("v_max3_f32", "v_max3_f32 {a}, {a}, {b}, {c}"),
    ("v_min3_f32", "v_min3_f32 {a}, {a}, {b}, {c}"),
    ("v_med3_f32", "v_med3_f32 {a}, {a}, {b}, {c}"),
    ("v_max3_u32", "v_max3_u32 {a}, {a}, {b}, {c}"),
    ("v_pk_add_f32", "v_pk_add_f32 {A}, {A}, {B}"),
    ("v_pk_mul_f32", "v_pk_mul_f32 {A}, {A}, {B}"),
    ("v_pk_fma_f32", "v_pk_fma_f32 {A}, {A}, {B}, {B}"),
    ("v_pk_add_u16", "v_pk_add_u16 {a}, {a}, {b}"),
    ("v_pk_max_u16", "v_pk_max_u16 {a}, {a}, {b}"),
    ("v_pk_add_f16", "v_pk_add_f16 {a}, {a}, {b}"),
    ("v_pk_max_f16", "v_pk_max_f16 {a}, {a}, {b}"),
    ("v_pk_min_f16", "v_pk_min_f16 {a}, {a}, {b}"),
    ("v_pk_fma_f16", "v_pk_fma_f16 {a}, {a}, {b}, {c}"),
    ("v_max_f16", "v_max_f16 {a}, {a}, {b}"),
    ("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 {a}, {b}"),
    ("v_cvt_f32_ubyte3", "v_cvt_f32_ubyte3 {a}, {b}"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {a}, {b}"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 {a}, {b}"),
    ("v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 {a}, {b}, 1, {a}"),
    ("v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 {a}, {a}, {b}"),
    ("v_perm_b32", "v_perm_b32 {a}, {a}, {b}, {c}"),
    ("v_alignbyte_b32", "v_alignbyte_b32 {a}, {a}, {b}, 3"),
    ("v_alignbit_b32", "v_alignbit_b32 {a}, {a}, {b}, 8"),
    ("v_bfe_u32", "v_bfe_u32 {a}, {a}, 8, 8"),
    ("v_bfi_b32", "v_bfi_b32 {a}, {b}, {a}, {c}"),
    ("v_and_or_b32", "v_and_or_b32 {a}, {a}, {b}, {c}"),
    ("v_xad_u32", "v_xad_u32 {a}, {a}, {b}, {c}"),
    ("v_sad_u8", "v_sad_u8 {a}, {a}, {b}, {c}"),
    ("v_msad_u8", "v_msad_u8 {a}, {a}, {b}, {c}"),
    ("v_lerp_u8", "v_lerp_u8 {a}, {a}, {b}, {c}"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 {a}, {b}, {c}, {a}"),
    ("v_dot2_u32_u16", "v_dot2_u32_u16 {a}, {b}, {c}, {a}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {a}, {a}, {b}, {c}"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {a}, {a}, {b}"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {a}, {a}, {b}"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {a}, {b}, {a}"),
    ("v_mbcnt_lo", "v_mbcnt_lo_u32_b32 {a}, {b}, {a}"),
    ("v_cmp_lt_u32", "v_cmp_lt_u32 vcc, {a}, {b}"),
    ("v_cmp_lt_f32", "v_cmp_lt_f32 vcc, {a}, {b}"),
    ("v_cmp_sdwa", "v_cmp_ne_u32_sdwa vcc, {a}, {b} src0_sel:BYTE_1 src1_sel:DWORD"),
    ("v_max_u32_sdwa", "v_max_u32_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"),
    ("v_max_f32_sdwa", "v_max_f32_sdwa {a}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"),
    ("v_mov_dpp_row_shr", "v_mov_b32_dpp {a}, {b} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_add_u32_dpp", "v_add_u32_dpp {a}, {a}, {b} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_max_f32_dpp", "v_max_f32_dpp {a}, {a}, {b} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_readlane", "v_readlane_b32 s20, {a}, 3"),
    ("v_add_f64", "v_add_f64 {A}, {A}, {B}"),
    ("v_mul_f64", "v_mul_f64 {A}, {A}, {B}"),
    ("v_fma_f64", "v_fma_f64 {A}, {A}, {B}, {B}"),
    ("v_max_f64", "v_max_f64 {A}, {A}, {B}"),
    ("v_rcp_f64", "v_rcp_f64 {A}, {B}"),
    ("v_rcp_f32", "v_rcp_f32 {a}, {b}"),
    ("v_rsq_f64", "v_rsq_f64 {A}, {B}"),
    ("v_mov_b64", "v_mov_b64 {A}, {B}"),
    ("v_pk_mov_b32", "v_pk_mov_b32 {A}, {A}, {B}"),
    ("v_accvgpr_write", "v_accvgpr_write_b32 a{i}, {b}"),
    ("v_accvgpr_read", "v_accvgpr_read_b32 {a}, a{i}"),
    ("v_min_u16", "v_min_u16 {a}, {a}, {b}"),
    ("v_add_u16", "v_add_u16 {a}, {a}, {b}"),
    ("v_sub_u16", "v_sub_u16 {a}, {a}, {b}"),
    ("v_max_i16", "v_max_i16 {a}, {a}, {b}"),
    ("v_min_i16", "v_min_i16 {a}, {a}, {b}"),
    ("v_mul_lo_u16", "v_mul_lo_u16 {a}, {a}, {b}"),
    ("v_lshlrev_b16", "v_lshlrev_b16 {a}, 1, {a}"),
    ("v_lshrrev_b16", "v_lshrrev_b16 {a}, 1, {a}"),
    ("v_add_f16", "v_add_f16 {a}, {a}, {b}"),
    ("v_min_f16", "v_min_f16 {a}, {a}, {b}"),
    ("v_mul_f16", "v_mul_f16 {a}, {a}, {b}"),
    ("v_cmp_lt_u16", "v_cmp_lt_u16 vcc, {a}, {b}"),
    ("v_max_u32_e64", "v_max_u32_e64 {a}, {a}, {b}"),
    ("v_max_u16_e64", "v_max_u16_e64 {a}, {a}, {b}"),
    ("v_add_f32_e64", "v_add_f32_e64 {a}, {a}, {b}"),
    ("v_max_u32_bc", "v_max_u32 {a}, {b}, {c}"),
    ("v_max_f32_bc", "v_max_f32 {a}, {b}, {c}"),
    ("v_and_b32_lit", "v_and_b32 {a}, 0xff00ff, {a}"),
    ("v_lshlrev_b32_v", "v_lshlrev_b32 {a}, {b}, {a}"),
    ("v_ashrrev_i32", "v_ashrrev_i32 {a}, 1, {a}"),
    ("v_subrev_u32", "v_subrev_u32 {a}, {a}, {b}"),
    ("v_not_b32", "v_not_b32 {a}, {b}"),
    ("v_bfrev_b32", "v_bfrev_b32 {a}, {b}"),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 {a}, {b}"),
    ("v_fma_f16", "v_fma_f16 {a}, {a}, {b}, {c}"),
    ("v_mad_u16", "v_mad_u16 {a}, {a}, {b}, {c}"),
    ("v_mad_legacy?", "v_mad_i32_i24 {a}, {a}, {b}, {c}"),
]
HEAD = r'''// valu_survey.hip -- GENERATED by tools/microbench/gen_valu_survey.py.  Issue rate of wave64 VALU opcodes on gfx950:
// 8 independent accumulators round-robin, 64 instructions per loop trip, at 2 waves per SIMD (one 512-thread
// workgroup per CU) and 8 (two 1024-thread workgroups per CU).  Rate = wall-clock wave-instructions per second per
// SIMD; cycles = shader clock (s_memtime / s_memrealtime of the median wave) / rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kTrips = 1024;
template <int KIND>
__global__ __launch_bounds__(1024) void k_issue(unsigned *out, unsigned long long *cyc, unsigned seed) {
  extern __shared__ unsigned lds[];
  unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  unsigned b = seed * 2654435761u + 1, c = seed ^ 0x01020304u;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, e = 1.0000001;
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < kTrips; t++) {
'''
TAIL = r'''  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  unsigned s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
  if (s == 0x12345 && ds == 1.5) out[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (size_t)gridDim.x * (blockDim.x >> 6);
    cyc[w] = t1 - t0;
    cyc[nw + w] = r0;
    cyc[2 * nw + w] = r1;
  }
}
template <int KIND>
void run(const char *name) {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  unsigned *out;
  unsigned long long *cyc;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * cus * 2 * 16 * 3));
  CK(hipFuncSetAttribute((const void *)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-20s", name);
  for (int k : {2, 4, 8}) {
    const int threads = k == 2 ? 512 : 1024, per_cu = k == 8 ? 2 : 1;
    const size_t ldsb = per_cu == 1 ? 160 * 1024 : 80 * 1024 - 512;
    const int grid = cus * per_cu, wpb = threads / 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(threads), ldsb, 0, out, cyc, 1u);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(threads), ldsb, 0, out, cyc, 2u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const size_t nw = (size_t)grid * wpb;
    std::vector<unsigned long long> all(3 * nw);
    CK(hipMemcpy(all.data(), cyc, sizeof(unsigned long long) * 3 * nw, hipMemcpyDeviceToHost));
    std::vector<double> clk(nw);
    for (size_t w = 0; w < nw; w++) clk[w] = (double)all[w] / (double)(all[2 * nw + w] - all[nw + w]) * 0.1;
    std::sort(clk.begin(), clk.end());
    const double ghz = clk[nw / 2];
    const double rate = (double)nw * kTrips * 64 / (ms * 1e-3) / (cus * 4) / 1e9;
    printf("  | %d waves/SIMD: %6.3f G/s/SIMD  %.2f GHz  %5.2f cycles", k, rate, ghz, ghz / rate);
  }
  printf("\n");
  CK(hipFree(out));
  CK(hipFree(cyc));
}
int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int first = argc > 1 ? atoi(argv[1]) : 0, last = argc > 2 ? atoi(argv[2]) : 1000;
'''
def body(i, tmpl):
    lines = []
    for j in range(64):
        r = j % 8
        if "{A}" in tmpl:
            ins = tmpl.format(A="%0", B="%1", i=r)
            lines.append(f'asm volatile("{ins}" : "+v"(d{r}) : "v"(e) : "vcc");')
        else:
            ins = tmpl.format(a="%0", b="%1", c="%2", i=r)
            clob = '"vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s30"' if ("s20" in ins or "s2" in tmpl) else '"vcc"'
            lines.append(f'asm volatile("{ins}" : "+v"(a{r}) : "v"(b), "v"(c) : {clob});')
    return "\n      ".join(lines)
src = HEAD
for i, (name, tmpl) in enumerate(OPS):
    src += f"    if (KIND == {i}) {{\n      {body(i, tmpl)}\n    }}\n"
src += TAIL
for i, (name, _) in enumerate(OPS):
    src += f'  if ({i} >= first && {i} <= last) run<{i}>("{name}");\n'
src += "  return 0;\n}\n"
import pathlib
pathlib.Path(__file__).with_name("valu_survey.hip").write_text(src)
print("wrote valu_survey.hip with", len(OPS), "ops")
