// ba.hip -- pose-only and local bundle adjustment on gfx950 (MI355X), FP64.
// Replaces myslam::Optimizer::solvePoseOnlySE3 / solveLocalBAPoseAndPoint (reference
// src/optimizer_ceres.cpp:157-314, 446-808) together with the Ceres solve they delegate to
// (TrustRegionMinimizer + LevenbergMarquardtStrategy + DENSE_SCHUR; contract in DESIGN.md).
//
// Device-resident LM: every iteration is a fixed sequence of kernels whose control decisions
// (accept / reject, radius, convergence) live in a small state struct in HBM, so the host never
// synchronises inside a solve and a multi-GPU driver only inserts two all-reduces per iteration.
//
//   k_ba_setup    start of a solve: edge activity -> problem membership (epoch stamps), optional
//                 chi2 classification, LM state reset, pose caches
//   k_ba_lin0     first linearisation: 16 lanes per map point evaluate its edges -> point block
//                 Hll/gl, Jacobi scale, rows of the K-major operand matrix Wt[3j+k][6c+a]
//   k_ba_gemm     Schur product  (W Hll^-1) W^T  (6nf x 3Np x 6nf) on the FP64 matrix cores
//                 (v_mfma_f64_16x16x4_f64), short K slices; extra blocks: camera blocks Hpp/gp
//   k_ba_reduce   (sharded) fixed-order sum of the partial slabs into the all-reduce payload
//   k_ba_solve    one workgroup: reduced camera system, blocked LDL^T in LDS, step, candidate poses
//   k_ba_backsub  back-substitution, candidate points, linearisation at the candidate, its cost;
//                 single GPU: the last block also runs the trust-region update
//   k_ba_reduce2 / k_ba_update  (sharded) second payload and trust-region bookkeeping
#include "ba_math.h"
#include "vo_common.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <numeric>
#include <type_traits>
#include <vector>

namespace {

using namespace vo;
using namespace vo::ba;

// ============================================================================================
// block reductions (fixed order => deterministic)
// ============================================================================================
// 64-lane sum, same value in every lane.  The four in-row steps are DPP moves of the two 32-bit halves (no LDS
// crossbar, a fraction of a ds_bpermute's latency); the cross-row steps use v_readlane of the row sums.  Fixed order.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror: every lane holds its row's sum
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
#ifdef VO_BA_STAMPS
// time stamp that cannot move above the computation of `dep`
__device__ __forceinline__ unsigned long long stamp_after(double &dep) {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep)::"memory");
  return t;
}
#endif
template <int N, int NW = 0>  // NW: wavefronts per block when known at compile time (0: blockDim.x / 64)
__device__ __forceinline__ void block_sum(double (&v)[N], double *lds /*>= nw*N*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = wave_sum(v[i]);
  const int nw = NW ? NW : (int)(blockDim.x >> 6);
  if (nw == 1) return;  // one wavefront: the wave sum is the block sum (no LDS, no barrier)
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; i++) lds[wave * N + i] = v[i];
  }
  __syncthreads();
  // all partials are read in one batch and summed in wave order (a runtime loop over the waves waits for LDS once
  // per term: 4 us for 27 sums)
  if (NW == 8 || (NW == 0 && nw == 8)) {
    double p[8][N];
#pragma unroll
    for (int w = 0; w < 8; w++)
#pragma unroll
      for (int i = 0; i < N; i++) p[w][i] = lds[w * N + i];
#pragma unroll
    for (int i = 0; i < N; i++)
      v[i] = (((((((0.0 + p[0][i]) + p[1][i]) + p[2][i]) + p[3][i]) + p[4][i]) + p[5][i]) + p[6][i]) + p[7][i];
    return;
  }
  if (NW == 2 || (NW == 0 && nw == 2)) {
    double p[2][N];
#pragma unroll
    for (int w = 0; w < 2; w++)
#pragma unroll
      for (int i = 0; i < N; i++) p[w][i] = lds[w * N + i];
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = (0.0 + p[0][i]) + p[1][i];
    return;
  }
  if (NW == 4 || (NW == 0 && nw == 4)) {
    double p[4][N];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int i = 0; i < N; i++) p[w][i] = lds[w * N + i];
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = (((0.0 + p[0][i]) + p[1][i]) + p[2][i]) + p[3][i];
    return;
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    double s = 0;
    for (int w = 0; w < nw; w++) s += lds[w * N + i];
    v[i] = s;
  }
}

// Inter-workgroup hand-off ("last block reduces") without fences, MI355X guide Guideline 16 form
// R1: EVERY handed-off byte is stored write-through (agent-scope relaxed atomic store = `sc1`) and
// loaded with an agent-scope relaxed atomic load (`sc1`, bypasses this CU's L1); every storing wave
// drains its stores (s_waitcnt vmcnt(0)), the block barriers, one lane takes a ticket with a relaxed
// agent-scope add.  The block that draws the last ticket reads the others' data.  Placement
// independent: per-XCD L2s are not coherent and a CU's L1 is never refreshed by other CUs.
__device__ __forceinline__ void st_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ bool arrive_and_check_last(unsigned int *counter, unsigned int expected, int *s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == expected - 1u);
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

// ============================================================================================
// Pose-only BA: Optimizer::solvePoseOnlySE3 (optimizer_ceres.cpp:157-314), one workgroup per frame
// ============================================================================================
struct PoseLm {
  double radius, decrease, x_cost, x_norm;
  int iterations, accepted, termination;
};

// A frame's observations (global memory, explicit address space: generic pointers in a struct turn every access
// into a flat load that waits for both counters).  Round 2 kept the first 768 of them in LDS in a compact form
// (one wavefront per SIMD cannot hide the latency of a load it waits for right away); with every load of a pass issued
// four trips ahead of its use the plain reads are as fast (0.284 against 0.286 ms per 1024 frames), and the 148 KB of
// LDS per CU the cache took go to the extraction kernels that run next to the solve (tracked step 4.39 -> 4.14 ms).
#define VO_GLOBAL __attribute__((address_space(1)))
constexpr int kPoseNd = 4;  // trips of 64 observations per register set of the one-wavefront pass (5 / 6: spills, round 3)
struct ObsView {
  const VO_GLOBAL double *pts, *obs, *isg;
  VO_GLOBAL uint8_t *outlier;  // the result, and the skip mask of a pass
  __device__ __forceinline__ void get(int i, double (&p)[3], double &ou, double &ov, double &our, double &is) const {
    p[0] = pts[3 * i], p[1] = pts[3 * i + 1], p[2] = pts[3 * i + 2];
    ou = obs[3 * i], ov = obs[3 * i + 1], our = obs[3 * i + 2];
    is = isg[i];
  }
};

// One observation's contribution to H (upper 21), g (6) and the cost; unscaled, loss-corrected.
//
// The pose Jacobian of edge_eval factors as J = A [I | X]: A = d r / d pc (rows u [a 0 c], v [0 b d], uR [a 0 e]) and
// X = -[pc]x, so with M = rho1 A^T A (3 x 3, M01 = 0) and m = rho1 A^T r
//     H += [ M    M X   ]      g += [ m     ]
//          [ .  X^T M X ]           [ X^T m ]
// and every column of X has two non-zeros: 15 + 14 + 12 multiply-adds and 14 adds for H (the dense row products of the
// 2-3 x 6 Jacobian: 60 + 15), the rotation half of J is never formed.
__device__ __forceinline__ void pose_obs_term(const PoseCache &P, const double (&pw)[3], double ou, double ov, double our,
                                              double is, const Cam &K, double hm, double hs, double (&acc)[28]) {
  // R p + t as three multiply-add chains that start from t (trans_point adds t last: a multiply and an add more per row)
  const double x = __builtin_fma(P.R[0], pw[0], __builtin_fma(P.R[1], pw[1], __builtin_fma(P.R[2], pw[2], P.t[0])));
  const double y = __builtin_fma(P.R[3], pw[0], __builtin_fma(P.R[4], pw[1], __builtin_fma(P.R[5], pw[2], P.t[1])));
  const double z = __builtin_fma(P.R[6], pw[0], __builtin_fma(P.R[7], pw[1], __builtin_fma(P.R[8], pw[2], P.t[2])));
  const double invz = inv_fast(z), invz2 = invz * invz;
  const bool stereo = !(our < 0);
  const double uhat = K.fx * x * invz + K.cx;
  const double r0 = (ou - uhat) * is;
  const double r1 = (ov - (K.fy * y * invz + K.cy)) * is;
  const double r2 = stereo ? (our - (uhat - K.bf * invz)) * is : 0.0;
  const double a = -invz * K.fx, b = -invz * K.fy, c = x * invz2 * K.fx, d = y * invz2 * K.fy;  // Jp[0], [7], [2], [8]
  const double a2 = stereo ? a : 0.0, e = stereo ? c - K.bf * invz2 : 0.0;                        // Jp[12], [14]
  const double s = r0 * r0 + r1 * r1 + r2 * r2;
  double rho0, rho1;
  huber(stereo ? hs : hm, s, rho0, rho1);
  acc[27] += 0.5 * rho0;
  const double wa = rho1 * a, wa2 = rho1 * a2, wb = rho1 * b, wc = rho1 * c, wd = rho1 * d, we = rho1 * e;
  const double M00 = __builtin_fma(wa, a, wa2 * a2), M02 = __builtin_fma(wa, c, wa2 * e), M11 = wb * b, M12 = wb * d;
  const double M22 = __builtin_fma(wc, c, __builtin_fma(wd, d, we * e));
  const double m0 = __builtin_fma(wa, r0, wa2 * r2), m1 = wb * r1, m2 = __builtin_fma(wc, r0, __builtin_fma(wd, r1, we * r2));
  // M X, X = [[0 z -y] [-z 0 x] [y -x 0]]
  const double X00 = y * M02, X10 = __builtin_fma(y, M12, -(z * M11)), X20 = __builtin_fma(y, M22, -(z * M12));
  const double X01 = __builtin_fma(z, M00, -(x * M02)), X11 = -(x * M12), X21 = __builtin_fma(z, M02, -(x * M22));
  const double X02 = -(y * M00), X12 = x * M11, X22 = __builtin_fma(x, M12, -(y * M02));
  // packed upper triangle, row by row: (0,b) 0..5, (1,b) 6..10, (2,b) 11..14, (3,b) 15..17, (4,b) 18..19, (5,5) 20
  acc[0] += M00, acc[2] += M02, acc[6] += M11, acc[7] += M12, acc[11] += M22;
  acc[3] += X00, acc[4] += X01, acc[5] += X02;
  acc[8] += X10, acc[9] += X11, acc[10] += X12;
  acc[12] += X20, acc[13] += X21, acc[14] += X22;
  // X^T (M X): column a of X against column b of M X
  acc[15] = __builtin_fma(y, X20, __builtin_fma(-z, X10, acc[15]));
  acc[16] = __builtin_fma(y, X21, __builtin_fma(-z, X11, acc[16]));
  acc[17] = __builtin_fma(y, X22, __builtin_fma(-z, X12, acc[17]));
  acc[18] = __builtin_fma(z, X01, __builtin_fma(-x, X21, acc[18]));
  acc[19] = __builtin_fma(z, X02, __builtin_fma(-x, X22, acc[19]));
  acc[20] = __builtin_fma(x, X12, __builtin_fma(-y, X02, acc[20]));
  acc[21] += m0, acc[22] += m1, acc[23] += m2;
  acc[24] = __builtin_fma(y, m2, __builtin_fma(-z, m1, acc[24]));
  acc[25] = __builtin_fma(z, m0, __builtin_fma(-x, m2, acc[25]));
  acc[26] = __builtin_fma(x, m1, __builtin_fma(-y, m0, acc[26]));
}

// A batch of observations of the one-wavefront form: ND trips of 64, in registers.
struct PoseOb { double pw[3], ou, ov, our, is; unsigned skip; };
// The loads take the wave-uniform bases from scalar registers and a 32-bit byte offset per lane (an int index costs
// twelve 64-bit address operations per observation); unconditional, the raw flag byte included: a bool would be
// compared, i.e. waited for, where it is loaded.
__device__ __forceinline__ void pose_request(const ObsView &V, unsigned base, unsigned last, PoseOb (&o)[kPoseNd]) {
  const unsigned lane = threadIdx.x;
  const VO_GLOBAL char *bp = (const VO_GLOBAL char *)V.pts, *bo = (const VO_GLOBAL char *)V.obs, *bi = (const VO_GLOBAL char *)V.isg;
  const VO_GLOBAL uint8_t *bs = V.outlier;
#pragma unroll
  for (int k = 0; k < kPoseNd; k++) {
    const unsigned i = min(base + 64u * k + lane, last);  // past the end: a harmless re-read
    const unsigned o24 = __umul24(i, 24u), o8 = i * 8u;  // (v_mul_lo_u32 is a quarter-rate instruction)
    o[k].pw[0] = *(const VO_GLOBAL double *)(bp + o24), o[k].pw[1] = *(const VO_GLOBAL double *)(bp + o24 + 8);
    o[k].pw[2] = *(const VO_GLOBAL double *)(bp + o24 + 16);
    o[k].ou = *(const VO_GLOBAL double *)(bo + o24), o[k].ov = *(const VO_GLOBAL double *)(bo + o24 + 8);
    o[k].our = *(const VO_GLOBAL double *)(bo + o24 + 16);
    o[k].is = *(const VO_GLOBAL double *)(bi + o8);
    o[k].skip = bs[i];
  }
}

// One linearisation pass over the observations that are not flagged.  WAVE (one wavefront per frame, nothing else on
// its SIMD to run while a load is in flight): observations travel in batches of four trips, one batch ahead of their
// use, into two register sets that swap roles in a loop unrolled by two (no "next becomes current" copies: 14 moves
// per observation).  `first` holds batch 0 on entry -- requested by the previous pass behind its last trip, so that it
// travels during the reduction and the 6 x 6 solve (the observations of a round do not change; a pass that requests
// its own first batch waits for it once per LM iteration) -- and again on exit.
template <bool WAVE>
__device__ __forceinline__ void pose_accumulate(const PoseCache &P, int n, const ObsView &V, const Cam &K, double hm, double hs,
                                                double (&acc)[28], PoseOb (&first)[kPoseNd]) {
#pragma unroll
  for (int i = 0; i < 28; i++) acc[i] = 0;
  if (WAVE) {
    const unsigned lane = threadIdx.x, last = (unsigned)(n - 1);
    constexpr int ND = kPoseNd;
    auto eval = [&](unsigned base, const PoseOb (&o)[ND]) {
#pragma unroll
      for (int k = 0; k < ND; k++)
        if (base + 64u * k + lane <= last && !o[k].skip) pose_obs_term(P, o[k].pw, o[k].ou, o[k].ov, o[k].our, o[k].is, K, hm, hs, acc);
    };
    PoseOb B[ND];
#pragma unroll 1
    for (unsigned base = 0; base <= last; base += 2 * ND * 64) {
      pose_request(V, base + ND * 64, last, B);
      eval(base, first);
      const unsigned nb = base + 2 * ND * 64;
      pose_request(V, nb > last ? 0u : nb, last, first);  // behind the last trip: batch 0 for the next pass (0.2275 -> 0.2235 ms)
      eval(base + ND * 64, B);  // (a batch wholly past the end evaluates nothing: every lane fails the range test)
    }
    return;
  }
#pragma unroll 1
  for (int i = threadIdx.x; i < n; i += (int)blockDim.x) {
    if (V.outlier[i]) continue;
    double pw[3], ou, ov, our, is;
    V.get(i, pw, ou, ov, our, is);
    pose_obs_term(P, pw, ou, ov, our, is, K, hm, hs, acc);
  }
}

// 6 x 6 SPD solve on the packed lower triangle (row by row: 00 10 11 20 21 22 ...), in place; b := A^-1 b.  Fully
// unrolled (every index a compile-time constant: the 21 + 6 + 6 values stay in registers); no divide and no square
// root: per column one Newton-refined v_rsq_f64 (an ulp or two from 1 / sqrt(d), like the per-observation arithmetic),
// the column and both substitutions multiply by it -- 6 reciprocal square roots instead of 6 IEEE square roots and 33 IEEE
// divides (about 1000 of the 3600 instructions an LM iteration spent outside the observation loop).
__device__ __forceinline__ constexpr int tri_l(int i, int j) { return i * (i + 1) / 2 + j; }            // j <= i
__device__ __forceinline__ constexpr int tri_u(int a, int b) { return a * 6 - a * (a - 1) / 2 + (b - a); }  // a <= b
__device__ __forceinline__ bool chol6_packed(double (&L)[21], double (&b)[6]) {
  double ri[6];  // 1 / L[j][j]
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double d = L[tri_l(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[tri_l(j, k)] * L[tri_l(j, k)];
    if (!(d > 0.0)) return false;
    ri[j] = rsqrt_fast(d);
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double t = L[tri_l(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= L[tri_l(i, k)] * L[tri_l(j, k)];
      L[tri_l(i, j)] = t * ri[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double t = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) t -= L[tri_l(i, k)] * b[k];
    b[i] = t * ri[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double t = b[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) t -= L[tri_l(k, i)] * b[k];
    b[i] = t * ri[i];
  }
  return true;
}

__device__ __forceinline__ void wave_lds_sync() {  // LDS hand-off between lanes of one wavefront
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 64 lanes x 28 partial sums -> 28 totals at dst (LDS), in a fixed order: a transpose through LDS instead of 28 x 6
// DPP / readlane steps (about 110 instructions for what those did in 1000).  Layout: value-major, scratch[k][lane] at a
// pitch of kPoseRedPitch doubles -- a lane's stores of one value land on consecutive doubles across the wavefront.
//   ONE_PASS (the one-wavefront kernel, round 5): all 28 values in one trip: lane 2 v + h sums rows h, h + 2, ... of value v
//   (32 loads in flight, a pairwise tree), the halves meet by one quad permute.  Pitch 66: the reads of a half-wave fall on
//   banks 4 v + 4 j + 2 h (mod 64), all distinct.  One store / hand-off / load / hand-off chain per linearisation instead of two.
//   Two passes of 14 values (the 256-thread form: four scratch areas must fit next to each other): lane 4 v + q sums rows
//   q, q + 4, ..., two quad permutes; pitch 68: banks 8 v + 2 q + 8 j, distinct inside each half-wave (the lane-major
//   [64][15] layout of round 3 read two-way conflicted: 49 % of the kernel's LDS cycles).
// The hand-offs wait for LDS only where it matters: a wavefront-scope fence also waits for vmcnt(0).
template <bool ONE_PASS>
struct PoseRed {
  static constexpr int kPitch = ONE_PASS ? 66 : 68;
  static constexpr int kScratch = (ONE_PASS ? 28 : 14) * kPitch;
};
__device__ __forceinline__ void wave_lds_handoff() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS only: the next pass's first batch stays in flight
  __builtin_amdgcn_wave_barrier();
}
template <bool ONE_PASS>
__device__ __forceinline__ void wave_reduce28(const double (&v)[28], double *scratch, double *dst) {
  constexpr int P = PoseRed<ONE_PASS>::kPitch;
  const int lane = threadIdx.x & 63;
  if (ONE_PASS) {
    const int vi = lane >> 1, h = lane & 1;
#pragma unroll
    for (int k = 0; k < 28; k++) scratch[k * P + lane] = v[k];
    wave_lds_handoff();
    double t = 0;
    if (lane < 56) {
      double u[32];
#pragma unroll
      for (int j = 0; j < 32; j++) u[j] = scratch[vi * P + 2 * j + h];
#pragma unroll
      for (int w = 16; w >= 1; w >>= 1)
#pragma unroll
        for (int j = 0; j < w; j++) u[j] += u[j + w];
      t = u[0];
    }
    t += dpp_f64<0xB1>(t);  // quad_perm [1,0,3,2]
    if (lane < 56 && h == 0) dst[vi] = t;
    wave_lds_handoff();
    return;
  }
  const int vi = lane >> 2, q = lane & 3;
#pragma unroll
  for (int c = 0; c < 2; c++) {
#pragma unroll
    for (int k = 0; k < 14; k++) scratch[k * P + lane] = v[14 * c + k];
    wave_lds_handoff();
    double t = 0;
    if (lane < 56) {
      // all sixteen loads in flight before the first add (a running sum over loads waits for LDS once per term),
      // summed pairwise in a fixed order
      double u[16];
#pragma unroll
      for (int j = 0; j < 16; j++) u[j] = scratch[vi * P + 4 * j + q];
#pragma unroll
      for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int j = 0; j < w; j++) u[j] += u[j + w];
      t = u[0];
    }
    t += dpp_f64<0xB1>(t);  // quad_perm [1,0,3,2]
    t += dpp_f64<0x4E>(t);  // quad_perm [2,3,0,1]
    if (lane < 56 && q == 0) dst[14 * c + vi] = t;
    wave_lds_handoff();
  }
}

// LDS doubles of a pose-only workgroup behind the observation cache: the reduction scratch, the linearisation at x
// and at the candidate
template <bool WAVE>
struct PoseLds {
  static constexpr int kScratch = PoseRed<WAVE>::kScratch;
  static constexpr int kRed = WAVE ? kScratch : 4 * kScratch + 4 * 28;  // per-wave scratch, then the wave totals
  double red[kRed];
  double acc[28];   // linearisation at x: 21 + 6 + 1 sums, uniform over the workgroup
  double cand[28];  // ... at the trial point
};

// -DVO_POSE_STAMPS (tools/pose_stamps.py): shader-clock cycles per phase of the LM loop, summed over the iterations and
// handed back in the summary's fields (initial_cost = solve, final_cost = plus, final_radius = pass, reserved = reduction,
// accepted = tests) -- a developer build, never the product.
#ifdef VO_POSE_STAMPS
#define POSE_STAMP(slot, dep)                                                                            \
  do {                                                                                                   \
    unsigned long long t_;                                                                               \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(dep)::"memory");                 \
    if ((slot) >= 0) st_[(slot) < 0 ? 0 : (slot)] += t_ - tp_;                                           \
    tp_ = t_;                                                                                            \
  } while (0)
#else
#define POSE_STAMP(slot, dep) do { } while (0)
#endif

// Ceres-style LM on one 6-dof pose.  The linearisations -- 21 + 6 + 1 sums each -- live in LDS, not in registers: a
// trial step accumulates the candidate's sums while the solve's temporaries are dead and vice versa (round 2 kept
// two sets of 28 accumulators next to a 6 x 6 system in every lane: 256 VGPR + 251 AGPR).  Every thread carries the
// (uniform) trust-region scalars in registers.
template <bool WAVE>
__device__ void pose_lm(double x[6], int n, const ObsView &V, const Cam &K, double hm, double hs, int max_it, PoseLds<WAVE> &S,
                        vo_lm_summary *sum) {
  // exp(x) is kept across the iterations (an accepted candidate's exp is the product se3_plus forms anyway) and the
  // residuals are evaluated from it: rotation matrix from the unit quaternion, t = V * upsilon -- what
  // se3TransPoint(x) computes through sin / cos of |omega|, up to rounding; no trigonometry per evaluation.
  Se3 Tx = se3_exp<true>(x);
#ifdef VO_POSE_STAMPS
  unsigned long long st_[5] = {0, 0, 0, 0, 0}, tp_ = 0;
#endif
  bool in_loop_ = false;  // (stamps)
  PoseOb first[kPoseNd];  // batch 0 of the next pass (one-wavefront form)
  if (WAVE) pose_request(V, 0, (unsigned)(n - 1), first);
  auto linearize = [&](const Se3 &T, double *dst) {  // sums of the linearisation at T -> dst (LDS)
    double v[28];
    pose_accumulate<WAVE>(pose_cache_se3(T), n, V, K, hm, hs, v, first);
    POSE_STAMP(in_loop_ ? 2 : -1, v[27]);
    if (WAVE) {
      wave_reduce28<true>(v, S.red, dst);
    } else {
      // every wavefront reduces its lanes through its own scratch (the same transpose as the one-wavefront kernel: a
      // tenth of the instructions of 28 DPP / readlane sums -- this path is the latency of ONE frame), then 28 threads add
      // the wave totals in wave order
      const int wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
      constexpr int kS = PoseLds<WAVE>::kScratch;
      double *tot = S.red + 4 * kS;
      __syncthreads();  // the previous pass's totals have been read
      wave_reduce28<false>(v, S.red + wave * kS, tot + wave * 28);
      __syncthreads();
      if (threadIdx.x < 28) {
        double t = 0;
        for (int w = 0; w < nw; w++) t += tot[w * 28 + threadIdx.x];
        dst[threadIdx.x] = t;
      }
      __syncthreads();
    }
  };
  // the linearisation at x and the one at the candidate swap roles when a step is accepted (no copy)
  double *cur = S.acc, *cnd = S.cand;
  linearize(Tx, cur);
  double scale[6];
#pragma unroll
  for (int a = 0; a < 6; a++) scale[a] = 1.0 / (1.0 + sqrt(cur[tri_u(a, a)]));
  // The trust-region scalars: radius and decrease with their reciprocals next to them (decrease is a power of two, so
  // radius * inv_decrease is the quotient exactly; 1 / radius is refreshed when an accepted step changes the radius) --
  // the damping of an iteration is six multiplications, not six IEEE divisions in front of the factorisation.
  double radius = 1e4, inv_radius = 1e-4, decrease = 2.0, inv_decrease = 0.5, x_cost = cur[27];
  const double initial_cost = x_cost;
  auto norm6 = [](const double (&v)[6]) {
    const double s2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
    return s2 > 1e-280 ? s2 * rsqrt_fast(s2) : 0.0;  // (|x| below 1e-140 counts as zero: the test adds 1e-8 to it)
  };
  double x_norm;
  {
    const double xv[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
    x_norm = norm6(xv);
  }
  int iterations = 0, accepted = 0, termination = 0, invalid = 0;
  bool last_ok = false;
  for (int it = 1;; it++) {
    if (it - 1 >= max_it) {
      termination = 0;
      break;
    }
    in_loop_ = true;
    POSE_STAMP(-1, x_cost);
    double h[27];  // one batch of LDS reads: the gradient test and the normal equations use the same values
#pragma unroll
    for (int i = 0; i < 27; i++) h[i] = cur[i];
    if (last_ok) {
      double gm = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) gm = fmax(gm, fabs(h[21 + a]));
      if (gm <= 1e-10) {
        termination = 3;
        break;
      }
    }
    if (radius < 1e-32) {
      termination = 4;
      break;
    }
    iterations = it;
    last_ok = false;
    // scaled normal equations  H'' = S H S, g'' = S g ; LM diagonal from clamp(diag H'')/radius
    double L[21], g[6], y[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
      for (int b2 = 0; b2 <= a; b2++) L[tri_l(a, b2)] = h[tri_u(b2, a)] * scale[b2] * scale[a];
      g[a] = h[21 + a] * scale[a];
      y[a] = g[a];
    }
    double model = 0, delta[6];
    {
      // step^T H'' step from the undamped matrix, before the factorisation overwrites it
      double Hs[21];
#pragma unroll
      for (int i = 0; i < 21; i++) Hs[i] = L[i];
#pragma unroll
      for (int a = 0; a < 6; a++) L[tri_l(a, a)] += fmin(fmax(L[tri_l(a, a)], 1e-6), 1e32) * inv_radius;
      bool ok = chol6_packed(L, y);
      if (ok) {
        double gs = 0, sHs = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
          if (!isfinite(y[a])) ok = false;
#pragma unroll
        for (int a = 0; a < 6; a++) {
          gs -= g[a] * y[a];  // step = -y
          double row = 0;
#pragma unroll
          for (int b2 = 0; b2 < 6; b2++) row += Hs[a >= b2 ? tri_l(a, b2) : tri_l(b2, a)] * y[b2];
          sHs += y[a] * row;
          delta[a] = -y[a] * scale[a];
        }
        model = -(gs + 0.5 * sHs);  // -m.(r + m/2) with m = J''step
      }
      if (!ok || !(model > 0.0)) {
        if (++invalid >= 5) {
          termination = 4;
          break;
        }
        radius *= inv_decrease, inv_radius *= decrease;
        decrease *= 2.0, inv_decrease *= 0.5;
        continue;
      }
    }
    invalid = 0;
    POSE_STAMP(0, model);
    double xc[6];
    Se3 Tc;
    se3_plus_keep(Tx, delta, xc, Tc);
    POSE_STAMP(1, xc[0]);
    // The candidate is linearised completely in the same pass (its cost is one of the 28 sums): an
    // accepted step -- the common case -- then needs no second sweep over the observations.
    linearize(Tc, cnd);
    double cand = cnd[27];
    POSE_STAMP(3, cand);
    if (!isfinite(cand)) cand = 1.7976931348623157e308;
    double sn = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) sn += (x[a] - xc[a]) * (x[a] - xc[a]);
    {
      const double tol = 1e-8 * (x_norm + 1e-8);  // |step| <= tol, compared as squares (no square root)
      if (sn <= tol * tol) {
        termination = 2;
        break;
      }
    }
    const double change = x_cost - cand;
    if (fabs(change) <= 1e-6 * x_cost) {
      termination = 1;
      break;
    }
    const double rel = change * inv_fast(model);
    if (rel > 1e-3) {
#pragma unroll
      for (int a = 0; a < 6; a++) x[a] = xc[a];
      Tx = Tc;
      x_norm = norm6(xc);
      double *t = cur;
      cur = cnd, cnd = t;  // the candidate's linearisation becomes the current one
      x_cost = cand;
      const double t2 = 2.0 * rel - 1.0;
      radius = fmin(radius * inv_fast(fmax(1.0 / 3.0, 1.0 - t2 * t2 * t2)), 1e16);
      inv_radius = inv_fast(radius);
      decrease = 2.0, inv_decrease = 0.5;
      accepted++;
      last_ok = true;
    } else {
      radius *= inv_decrease, inv_radius *= decrease;
      decrease *= 2.0, inv_decrease *= 0.5;
    }
    POSE_STAMP(4, radius);
  }
  if (sum && threadIdx.x == 0) {
    sum->iterations = iterations;
    sum->accepted = accepted;
    sum->termination = termination;
    sum->reserved = 0;
    sum->initial_cost = initial_cost;
    sum->final_cost = x_cost;
    sum->final_radius = radius;
#ifdef VO_POSE_STAMPS
    sum->initial_cost = (double)st_[0], sum->final_cost = (double)st_[1], sum->final_radius = (double)st_[2];
    sum->reserved = (int)st_[3], sum->accepted = (int)st_[4];
#endif
  }
}

// float chi2 test of optimizer_ceres.cpp:262-303 (Q-B2: deliberately float)
__device__ __forceinline__ bool pose_chi2_outlier(const double pc[3], double ou, double ov, double our, float fx,
                                                  float fy, float cx, float cy, float bf, double isg) {
  const double x = pc[0], y = pc[1], z = pc[2];
  const float invz = (float)(1.0f / z);
  const float u = (float)(fx * x * invz + cx);
  const float v = (float)(fy * y * invz + cy);
  const float eu = (float)(u - ou), ev = (float)(v - ov);
  const float e2 = eu * eu + ev * ev;
  const float is2 = (float)(isg * isg);
  if (our < 0) return !(e2 * is2 < 5.991f);
  const float ur = u - bf * invz;
  const float eur = (float)(ur - our);
  return !((e2 + eur * eur) * is2 < 7.815f);
}

// ranges != 0: problem p owns observations [offsets[2p], offsets[2p] + offsets[2p+1]) (frames at a fixed stride,
// vo_track_gather_dev); otherwise [offsets[p], offsets[p+1]).  WAVE: one wavefront per problem (batches); otherwise
// 128 or 256 threads per problem (a few problems: the observations are shared out).
template <bool WAVE>
__global__ __launch_bounds__(WAVE ? 64 : 256) void k_pose_only(const int *offsets, const double *pts, const double *obs,
                                                               const double *isg, const double *cam5, double *poses,
                                                               uint8_t *outlier, int *n_inliers, vo_lm_summary *sums,
                                                               int ranges) {
  __shared__ PoseLds<WAVE> S;
  __shared__ int s_cnt[4];
  const int p = blockIdx.x;
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;  // (workgroups of one wavefront leave three of the four slots unused)
  const int o0 = ranges ? offsets[2 * p] : offsets[p], n = ranges ? offsets[2 * p + 1] : offsets[p + 1] - o0;
  pts += 3 * (long long)o0, obs += 3 * (long long)o0, isg += o0, outlier += o0;
  Cam K{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
  const float fx = (float)K.fx, fy = (float)K.fy, cx = (float)K.cx, cy = (float)K.cy, bf = (float)K.bf;
  double x0[6], x[6];
  for (int a = 0; a < 6; a++) x0[a] = x[a] = poses[6 * p + a];
  if (n <= 0) {  // :204-205
    if (threadIdx.x == 0) n_inliers[p] = 0;
    return;
  }
  const ObsView V{(const VO_GLOBAL double *)pts, (const VO_GLOBAL double *)obs, (const VO_GLOBAL double *)isg, (VO_GLOBAL uint8_t *)outlier};
  const int stride = WAVE ? 64 : (int)blockDim.x;
  for (int i = threadIdx.x; i < n; i += stride) outlier[i] = 0;
  __syncthreads();
  int inl = 0;
  for (int round = 0; round < 2; round++) {
    for (int a = 0; a < 6; a++) x[a] = x0[a];  // :215
    const double hm = round == 0 ? (double)sqrtf(5.991f) : 0.0;
    const double hs = round == 0 ? (double)sqrtf(7.815f) : 0.0;
    pose_lm<WAVE>(x, n, V, K, hm, hs, 10, S, sums ? &sums[2 * p + round] : nullptr);
    __syncthreads();
    // classification with Tcw = exp(pose) (Sophus quaternion form, :256-257)
    const Se3 T = se3_exp(x);
    int local = 0;
#pragma unroll 1
    for (int i = threadIdx.x; i < n; i += stride) {
      double rp[3], pc[3], pw[3], ou, ov, our, is;
      V.get(i, pw, ou, ov, our, is);
      quat_rotate(T.q, pw, rp);
      pc[0] = rp[0] + T.t[0], pc[1] = rp[1] + T.t[1], pc[2] = rp[2] + T.t[2];
      const bool out = pose_chi2_outlier(pc, ou, ov, our, fx, fy, cx, cy, bf, is);
      outlier[i] = out ? 1 : 0;
      local += out ? 0 : 1;
    }
    for (int o = 32; o >= 1; o >>= 1) local += __shfl_xor(local, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = local;
    __syncthreads();
    inl = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
    if (inl < 10) {  // :306-307
      if (round == 0 && sums && threadIdx.x == 0) memset(&sums[2 * p + 1], 0, sizeof(vo_lm_summary));
      break;
    }
  }
  if (threadIdx.x == 0) {
    n_inliers[p] = inl;
    for (int a = 0; a < 6; a++) poses[6 * p + a] = x[a];
  }
}

// ============================================================================================
// Sim3 optimisation of a loop candidate: Optimizer::solveLoopSim3 (optimizer_ceres.cpp:810-1030),
// one workgroup per problem, same structure as k_pose_only.  NP = 6 (scale fixed, the only mode the
// reference uses: loopClosing.cpp:15) or 7.
// ============================================================================================
struct Sim3Prob {
  int n;
  const double *Pm, *pc, *isc, *Pc, *pm, *ism;  // cam_match(3n), pix_curr(2n), 1/sigma, cam_curr(3n), pix_match(2n), 1/sigma
  const uint8_t *skip;
  double cam[4];
  double huber;
};

template <int NP>
__device__ void sim3_accumulate(const double x[7], const Sim3Prob &Q, bool want_jac,
                                double (&acc)[NP * (NP + 1) / 2 + NP + 1]) {
  constexpr int NH = NP * (NP + 1) / 2;
#pragma unroll
  for (int i = 0; i < NH + NP + 1; i++) acc[i] = 0;
  const Sim3Frame F = sim3_frame(x, want_jac);
  for (int i = threadIdx.x; i < Q.n; i += blockDim.x) {
    if (Q.skip && Q.skip[i]) continue;
    double r[4], J[28];
    if (want_jac)
      sim3_eval<true>(F, Q.Pm + 3 * i, Q.pc[2 * i], Q.pc[2 * i + 1], Q.isc[i], Q.Pc + 3 * i, Q.pm[2 * i], Q.pm[2 * i + 1],
                      Q.ism[i], Q.cam, r, J);
    else
      sim3_eval<false>(F, Q.Pm + 3 * i, Q.pc[2 * i], Q.pc[2 * i + 1], Q.isc[i], Q.Pc + 3 * i, Q.pm[2 * i], Q.pm[2 * i + 1],
                       Q.ism[i], Q.cam, r, nullptr);
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {  // each 2-row block has its own loss (two AddResidualBlock calls, :888-895)
      const double *rb = r + 2 * blk, *Jb = J + 14 * blk;
      double rho0, rho1;
      huber(Q.huber, rb[0] * rb[0] + rb[1] * rb[1], rho0, rho1);
      acc[NH + NP] += 0.5 * rho0;
      if (!want_jac) continue;
      int t = 0;
#pragma unroll
      for (int a = 0; a < NP; a++) {
#pragma unroll
        for (int b = a; b < NP; b++) acc[t++] += rho1 * (Jb[a] * Jb[b] + Jb[7 + a] * Jb[7 + b]);
        acc[NH + a] += rho1 * (Jb[a] * rb[0] + Jb[7 + a] * rb[1]);
      }
    }
  }
}

template <int N>
__device__ bool chol_solve_n(double (&A)[N][N], double (&b)[N]) {
#pragma unroll
  for (int j = 0; j < N; j++) {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= A[j][k] * A[j][k];
    if (!(d > 0.0)) return false;
    d = sqrt(d);
    A[j][j] = d;
#pragma unroll
    for (int i = j + 1; i < N; i++) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) v -= A[i][k] * A[j][k];
      A[i][j] = v / d;
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) v -= A[i][k] * b[k];
    b[i] = v / A[i][i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    double v = b[i];
#pragma unroll
    for (int k = i + 1; k < N; k++) v -= A[k][i] * b[k];
    b[i] = v / A[i][i];
  }
  return true;
}

// Ceres-style LM (same contract as pose_lm) on the first NP entries of x with the plain additive update
template <int NP>
__device__ void sim3_lm(double x[7], const Sim3Prob &Q, int max_it, double *lds, vo_lm_summary *sum) {
  constexpr int NH = NP * (NP + 1) / 2, NA = NH + NP + 1;
  double acc[NA];
  sim3_accumulate<NP>(x, Q, true, acc);
  block_sum<NA>(acc, lds);
  double scale[NP];
  {
    int t = 0;
#pragma unroll
    for (int a = 0; a < NP; a++) {
      scale[a] = 1.0 / (1.0 + sqrt(acc[t]));
      t += NP - a;
    }
  }
  auto norm_free = [](const double *v) {
    double q = 0;
#pragma unroll
    for (int a = 0; a < NP; a++) q += v[a] * v[a];
    return sqrt(q);
  };
  double radius = 1e4, decrease = 2.0, x_cost = acc[NH + NP];
  const double initial_cost = x_cost;
  double x_norm = norm_free(x);
  int iterations = 0, accepted = 0, termination = 0, invalid = 0;
  bool last_ok = false;
  for (int it = 1;; it++) {
    if (it - 1 >= max_it) {
      termination = 0;
      break;
    }
    if (last_ok) {
      double gm = 0;
#pragma unroll
      for (int a = 0; a < NP; a++) gm = fmax(gm, fabs(acc[NH + a]));
      if (gm <= 1e-10) {
        termination = 3;
        break;
      }
    }
    if (radius < 1e-32) {
      termination = 4;
      break;
    }
    iterations = it;
    last_ok = false;
    double A[NP][NP], Hs[NP][NP], g[NP], y[NP];
    {
      int t = 0;
#pragma unroll
      for (int a = 0; a < NP; a++)
#pragma unroll
        for (int b = a; b < NP; b++) {
          const double v = acc[t++] * scale[a] * scale[b];
          Hs[a][b] = Hs[b][a] = v;
        }
    }
#pragma unroll
    for (int a = 0; a < NP; a++) {
      g[a] = acc[NH + a] * scale[a];
#pragma unroll
      for (int b = 0; b < NP; b++) A[a][b] = Hs[a][b];
      A[a][a] += fmin(fmax(Hs[a][a], 1e-6), 1e32) / radius;
      y[a] = g[a];
    }
    bool ok = chol_solve_n<NP>(A, y);
    double delta[7] = {0, 0, 0, 0, 0, 0, 0}, model = 0;
    if (ok) {
      double gs = 0, sHs = 0;
#pragma unroll
      for (int a = 0; a < NP; a++)
        if (!isfinite(y[a])) ok = false;
#pragma unroll
      for (int a = 0; a < NP; a++) {
        gs -= g[a] * y[a];
        double row = 0;
#pragma unroll
        for (int b = 0; b < NP; b++) row -= Hs[a][b] * y[b];
        sHs -= y[a] * row;
        delta[a] = -y[a] * scale[a];
      }
      model = -(gs + 0.5 * sHs);
    }
    if (!ok || !(model > 0.0)) {
      if (++invalid >= 5) {
        termination = 4;
        break;
      }
      radius /= decrease;
      decrease *= 2.0;
      continue;
    }
    invalid = 0;
    double xc[7];
#pragma unroll
    for (int a = 0; a < 7; a++) xc[a] = x[a] + delta[a];
    double cacc[NA];  // complete linearisation at the candidate (see pose_lm)
    sim3_accumulate<NP>(xc, Q, true, cacc);
    block_sum<NA>(cacc, lds);
    double cand = cacc[NH + NP];
    if (!isfinite(cand)) cand = 1.7976931348623157e308;
    double sn = 0;
#pragma unroll
    for (int a = 0; a < NP; a++) sn += (x[a] - xc[a]) * (x[a] - xc[a]);
    if (sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) {
      termination = 2;
      break;
    }
    const double change = x_cost - cand;
    if (fabs(change) <= 1e-6 * x_cost) {
      termination = 1;
      break;
    }
    const double rel = change / model;
    if (rel > 1e-3) {
#pragma unroll
      for (int a = 0; a < 7; a++) x[a] = xc[a];
      x_norm = norm_free(x);
#pragma unroll
      for (int i = 0; i < NA; i++) acc[i] = cacc[i];
      x_cost = acc[NH + NP];
      const double t2 = 2.0 * rel - 1.0;
      radius = fmin(radius / fmax(1.0 / 3.0, 1.0 - t2 * t2 * t2), 1e16);
      decrease = 2.0;
      accepted++;
      last_ok = true;
    } else {
      radius /= decrease;
      decrease *= 2.0;
    }
  }
  if (sum && threadIdx.x == 0) {
    sum->iterations = iterations;
    sum->accepted = accepted;
    sum->termination = termination;
    sum->reserved = 0;
    sum->initial_cost = initial_cost;
    sum->final_cost = x_cost;
    sum->final_radius = radius;
  }
}

template <int NP>
__global__ __launch_bounds__(256) void k_sim3(const int *offsets, const double *Pm, const double *pc, const double *isc,
                                              const double *Pc, const double *pm, const double *ism, const double *cam4,
                                              double *poses, double *scales, uint8_t *outlier, int *n_inliers,
                                              vo_lm_summary *sums) {
  __shared__ double lds[4 * 36];
  __shared__ int s_cnt[4];
  const int p = blockIdx.x;
  const int o0 = offsets[p], n = offsets[p + 1] - o0;
  Sim3Prob Q;
  Q.n = n, Q.Pm = Pm + 3 * (long long)o0, Q.pc = pc + 2 * (long long)o0, Q.isc = isc + o0;
  Q.Pc = Pc + 3 * (long long)o0, Q.pm = pm + 2 * (long long)o0, Q.ism = ism + o0;
  Q.skip = nullptr;
  for (int a = 0; a < 4; a++) Q.cam[a] = cam4[a];
  Q.huber = (double)sqrtf(10.0f);  // :880
  outlier += o0;
  double x[7], x_in[7];
  for (int a = 0; a < 6; a++) x[a] = x_in[a] = poses[6 * p + a];
  x[6] = x_in[6] = scales[p];
  for (int i = threadIdx.x; i < n; i += blockDim.x) outlier[i] = 0;
  __syncthreads();
  if (sums && threadIdx.x == 0) {
    vo_lm_summary z = {};
    sums[2 * p] = z, sums[2 * p + 1] = z;
  }
  if (n > 0) sim3_lm<NP>(x, Q, 10, lds, sums ? &sums[2 * p] : nullptr);
  auto classify = [&](bool keep_old) {  // returns the number of matches passing both chi2 tests now
    const Sim3Frame F = sim3_frame(x, false);
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const bool o = sim3_outlier(F, Q.Pm + 3 * i, Q.pc[2 * i], Q.pc[2 * i + 1], Q.isc[i], Q.Pc + 3 * i, Q.pm[2 * i],
                                  Q.pm[2 * i + 1], Q.ism[i], Q.cam);
      if (o || !keep_old) outlier[i] = o ? 1 : 0;
      cnt += o ? 0 : 1;
    }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) cnt += __shfl_xor(cnt, o2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    return s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  };
  const int ok1 = classify(false);
  int inliers = 0;
  if (ok1 < 10) {  // :950-951 returns before Scm is written
    for (int a = 0; a < 7; a++) x[a] = x_in[a];
  } else {
    Q.skip = outlier;  // problem 2: survivors only (:958-960); continues from problem 1's estimate
    __syncthreads();
    sim3_lm<NP>(x, Q, ok1 < n ? 10 : 5, lds, sums ? &sums[2 * p + 1] : nullptr);
    __syncthreads();
    inliers = classify(true);  // :996-1022 tests every match again
  }
  if (threadIdx.x == 0) {
    for (int a = 0; a < 6; a++) poses[6 * p + a] = x[a];
    scales[p] = x[6];
    n_inliers[p] = inliers;
    if (sums) sums[2 * p].reserved = ok1 < 10 ? 1 : 2;  // phase reached: the shim writes Scm only after phase 2
  }
}

// ============================================================================================
// Local BA
// ============================================================================================
constexpr int kGroupLog = 4;
constexpr int kGroup = 1 << kGroupLog;  // lanes per map point: one edge per lane at the usual <= 16 observations,
                                        // and twice the workgroups -- the point kernels are FP64-latency bound
                                        // with one wavefront per SIMD
constexpr int kPtsPerBlock = 256 / kGroup;
constexpr int kGemmThreads = 512;  // k_ba_gemm / k_ba_cams_large: two wavefronts per SIMD (each wave's K-slice is one 48-row trip)
constexpr int kCamChunk = 256;  // edges per camera-role block, one per thread of its first four wavefronts (the role shares a
                                // launch with the Schur tiles; 27 wave sums per wavefront are its cost, so the other four retire)
constexpr int kPcLds = 128;     // pose caches staged in LDS by the point kernels (12 KB)
constexpr int kMaxN = 128;       // reduced system size limit of the LDS Cholesky (6*nf + 1 <= kMaxN)

struct BaState {
  double radius, decrease, x_cost, cand_cost, initial_cost, x_norm2_c, cand_norm2_c, step_norm2_c, gdot_c,
      dquad_c, gmax;
  double hm, hs;  // Huber thresholds of the running solve (<= 0: no loss)
  int iter, accepted, termination, done, invalid, last_ok, cur, max_it, first, solve_failed;
};

struct BaDev {
  int n_cams, n_pts, n_edges, nf, n_local;  // n_local = points owned by this shard
  int Mpad;                                 // padded reduced size (multiple of 16), 6nf+1 <= Mpad
  int ksplit, kchunk;                       // split-K of the Schur GEMM
  int n_pblocks, n_cchunks;
  int n_shards, shard;
  Cam K;
  // problem
  const int *e_cam, *e_pt;
  const double *e_obs, *e_is;
  uint8_t *e_active;
  const int *pt_start;      // [n_pts+1] into the point-sorted edge arrays
  const int *local_pts;     // [n_local]
  const int *cam_slot;      // [n_cams] free index or -1
  const int *slot_cam;      // [nf]
  const int *cam_start;     // [nf+1]
  const int *cam_edges;     // edge ids per free camera (this shard's edges only)
  uint8_t *pt_in, *cam_in;  // == epoch: the block is in the current problem (no clearing pass between solves)
  int epoch;
  // state
  double *Xc[2], *Xp[2];
  double *PC[2];            // pose caches (R row-major 9 + t 3) of Xc[0/1], one per camera
  double *scale_c, *scale_p;
  double *hinv, *gl2, *dl;  // per point, for the current radius: inverse (6), scaled gradient (3), LM diagonal (3)
  // linearisation products, double-buffered like the state ([cur] = at x, [cur^1] = at the candidate)
  double *Wt[2];            // [3*n_pts][Mpad]  W_cj[a][k] * sp[k] at row 3j+k, column 6c+a; column 6nf = sp*gl
  double *hll[2];           // [n_pts][6] packed upper triangle of sum Jl^T Jl (unscaled)
  double *slab_pt[2];       // [n_pblocks][2]  cost, gmax
  double *slab_gemm;        // [ksplit][Mpad*Mpad]
  double *slab_cam;         // [nf][n_cchunks][27]
  double *payload;          // Mpad*Mpad + nf*27 + 1 + n_shards
  double *zc;               // [6nf] scale_c * y_c
  double *slab_bs;          // [n_pblocks][6]
  double *payload2;         // 6
  unsigned long long *dbg;  // optional stamps (VO_BA_STAMPS builds)
  BaState *st;
  BaState *hist;            // [2] states of earlier solves of the same schedule
  unsigned int *counters;   // [0] back-substitution arrivals, [1 + tile] GEMM K-slice arrivals
  unsigned div_np1;         // ceil(2^32 / (6 nf + 1)): e / (n + 1) = umulhi(e, div_np1) for e (n + 1) < 2^32
  int fused;                // single shard: in-kernel reductions replace k_ba_reduce / k_ba_reduce2
  // ---- large reduced systems (6 nf + 1 > kMaxN, e.g. a global BA over hundreds of key-frames): no dense
  // operand matrix; the camera-point blocks live per edge and the reduced system is a dense ld x ld
  // matrix in HBM, factored by the blocked Cholesky of csrc/pose_graph.hip
  int large, ld;
  double *We[2];            // [n_edges][3][6]  W_cj[a][k] * sp[k] at [k][a]
  double *glsc[2];          // [n_pts][3] scaled point gradient
  double *Sd;               // Cholesky storage: reduced system, rhs row, solution row, extras (large_ext_off)
  double *sc_v, *Dd_v, *gpp_v;  // per reduced column: Jacobi scale, LM diagonal, scaled gradient
  int *chol_fail;
  const int *pair_start;    // [n_pairs][2] begin / end in pair_e; pairs in work order (below), padded with empty ones
  int pair_diag_blocks;     // leading workgroups (4 pairs each) that hold the (c, c) pairs
  const int *pair_cc;       // [n_pairs][2] camera slots (c <= c')
  const int4 *pair_e;       // [..] (edge of c, edge of c', point, 0) at a shared point, in point order
  const int2 *ltiles;       // large systems: the 64 x 64 tiles (row, column | partial << 16) of the factor that exist (the plan's, fill included)
  int n_ltiles;
  int n_pairs;
  // per-rank segment factorisation of a sharded large system (build_device / run_lm_eager): points are owned by the rank
  // of the nested-dissection segment they touch, so a rank's segment columns are complete and only the separator block
  // [seg_row0, ld) is a partial sum; the replicated terms of that block (camera blocks, LM diagonal, padding identity,
  // gradient) are added by one rank (seg_lead)
  int seg_mode, seg_row0, seg_lead;
  unsigned long long seg_own;  // bit j: tile column j belongs to one of this rank's segments
};

// large path: what rides along with the reduced system in the same buffer (so that one all-reduce carries
// everything): row ld = right-hand side, row ld + 1 = solution, from row ld + 2 on: camera blocks
// [nf][27], cost, one gradient-max slot per shard
__device__ __host__ __forceinline__ long long large_ext_off(int ld) { return (long long)(ld + 2) * ld; }

// address of the W block row k of edge e (point j, camera slot): dense operand matrix or per-edge store
__device__ __forceinline__ double *w_row(const BaDev &B, int buf, int e, int j, int slot, int k) {
  return B.large ? B.We[buf] + 18LL * e + 6 * k : B.Wt[buf] + (long long)(3 * j + k) * B.Mpad + 6 * slot;
}

__device__ __forceinline__ PoseCache load_pc(const double *pc, int c) {
  PoseCache P;
  const double *q = pc + 12 * c;
#pragma unroll
  for (int i = 0; i < 9; i++) P.R[i] = q[i];
  P.t[0] = q[9], P.t[1] = q[10], P.t[2] = q[11];
  return P;
}
__device__ __forceinline__ void store_pc(double *pc, int c, const PoseCache &P) {
  double *q = pc + 12 * c;
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = P.R[i];
  q[9] = P.t[0], q[10] = P.t[1], q[11] = P.t[2];
}

__device__ __forceinline__ int payload_hpp_off(const BaDev &B) { return B.Mpad * B.Mpad; }
__device__ __forceinline__ int payload_cost_off(const BaDev &B) { return B.Mpad * B.Mpad + B.nf * 27; }

// --------------------------------------------------------------------------------------------
// k_ba_points: LocalBAProjectUV / LocalBAStereoProjectUVD::Evaluate (:320-444) for the edges of a
// point, Ceres' loss correction, the e-block E^T E + D and its inverse (SchurEliminator), and this
// point's rows of the GEMM operands.
// --------------------------------------------------------------------------------------------
// Linearisation of one map point at (point position `pt`, camera caches `PCs`) by the kGroup lanes of
// its group: point block Hll (unscaled) and gradient, Jacobi scale (first linearisation of a
// solve), and the point's rows of the K-major operand matrix W (point-scaled).  Nothing here
// depends on the trust-region radius, so the same routine linearises the *candidate* inside the
// back-substitution kernel; the radius-dependent inverse is formed later by k_ba_gemm.
__device__ __forceinline__ void point_linearize(const BaDev &B, const BaState &st, int j, int g, const double pt[3],
                                                const double *PCs, int buf, double &cost, double &gmax) {
  const int e0 = B.pt_start[j], e1 = B.pt_start[j + 1];
  double h[6] = {0, 0, 0, 0, 0, 0}, gl[3] = {0, 0, 0};
  double sp[3] = {1, 1, 1};
  const bool first = st.first != 0;
  if (!first) sp[0] = B.scale_p[3 * j], sp[1] = B.scale_p[3 * j + 1], sp[2] = B.scale_p[3 * j + 2];
  for (int e = e0 + g; e < e1; e += kGroup) {
    const int slot = B.cam_slot[B.e_cam[e]];
    if (!B.e_active[e]) {  // rows of a deactivated edge stay zero
      if (slot >= 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          double *w = w_row(B, buf, e, j, slot, k);
#pragma unroll
          for (int a = 0; a < 6; a++) w[a] = 0;
        }
      }
      continue;
    }
    const PoseCache P = load_pc(PCs, B.e_cam[e]);
    double r[3], Jp[18], Jl[9];
    const int m = edge_eval<true, true>(P, pt, B.e_obs[3 * e], B.e_obs[3 * e + 1], B.e_obs[3 * e + 2], B.e_is[e],
                                        B.K, r, Jp, Jl);
    double rho0, rho1;
    huber(m == 2 ? st.hm : st.hs, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho0, rho1);
    cost += 0.5 * rho0;
    h[0] += rho1 * (Jl[0] * Jl[0] + Jl[3] * Jl[3] + Jl[6] * Jl[6]);
    h[1] += rho1 * (Jl[0] * Jl[1] + Jl[3] * Jl[4] + Jl[6] * Jl[7]);
    h[2] += rho1 * (Jl[0] * Jl[2] + Jl[3] * Jl[5] + Jl[6] * Jl[8]);
    h[3] += rho1 * (Jl[1] * Jl[1] + Jl[4] * Jl[4] + Jl[7] * Jl[7]);
    h[4] += rho1 * (Jl[1] * Jl[2] + Jl[4] * Jl[5] + Jl[7] * Jl[8]);
    h[5] += rho1 * (Jl[2] * Jl[2] + Jl[5] * Jl[5] + Jl[8] * Jl[8]);
    gl[0] += rho1 * (Jl[0] * r[0] + Jl[3] * r[1] + Jl[6] * r[2]);
    gl[1] += rho1 * (Jl[1] * r[0] + Jl[4] * r[1] + Jl[7] * r[2]);
    gl[2] += rho1 * (Jl[2] * r[0] + Jl[5] * r[1] + Jl[8] * r[2]);
    if (!first && slot >= 0) {  // the scale is known: W rows can be written in the same sweep
#pragma unroll
      for (int k = 0; k < 3; k++) {
        double *w = w_row(B, buf, e, j, slot, k);
#pragma unroll
        for (int a = 0; a < 6; a++) w[a] = rho1 * (Jp[a] * Jl[k] + Jp[6 + a] * Jl[3 + k] + Jp[12 + a] * Jl[6 + k]) * sp[k];
      }
    }
  }
  // butterfly over the lanes of the group: every lane ends with the totals
#pragma unroll
  for (int o = 1; o < kGroup; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 6; i++) h[i] += __shfl_xor(h[i], o);
#pragma unroll
    for (int i = 0; i < 3; i++) gl[i] += __shfl_xor(gl[i], o);
  }
  if (first) {  // Jacobi scaling 1/(1+||column||), fixed for the rest of the solve
    sp[0] = 1.0 / (1.0 + sqrt(h[0])), sp[1] = 1.0 / (1.0 + sqrt(h[3])), sp[2] = 1.0 / (1.0 + sqrt(h[5]));
    if (g == 0) B.scale_p[3 * j] = sp[0], B.scale_p[3 * j + 1] = sp[1], B.scale_p[3 * j + 2] = sp[2];
  }
  gmax = fmax(gmax, fmax(fabs(gl[0]), fmax(fabs(gl[1]), fabs(gl[2]))));
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) B.hll[buf][6 * j + i] = h[i];
    // extra GEMM column 6nf carries the scaled gradient so that the same product yields sum_j Y_j gl''_j
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (B.large)
        B.glsc[buf][3 * j + k] = gl[k] * sp[k];
      else
        B.Wt[buf][(long long)(3 * j + k) * B.Mpad + 6 * B.nf] = gl[k] * sp[k];
    }
  }
  if (first) {  // second sweep, now that the scale exists
    for (int e = e0 + g; e < e1; e += kGroup) {
      const int slot = B.cam_slot[B.e_cam[e]];
      if (slot < 0 || !B.e_active[e]) continue;
      const PoseCache P = load_pc(PCs, B.e_cam[e]);
      double r[3], Jp[18], Jl[9];
      const int m = edge_eval<true, true>(P, pt, B.e_obs[3 * e], B.e_obs[3 * e + 1], B.e_obs[3 * e + 2], B.e_is[e],
                                          B.K, r, Jp, Jl);
      double rho0, rho1;
      huber(m == 2 ? st.hm : st.hs, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho0, rho1);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        double *w = w_row(B, buf, e, j, slot, k);
#pragma unroll
        for (int a = 0; a < 6; a++) w[a] = rho1 * (Jp[a] * Jl[k] + Jp[6 + a] * Jl[3 + k] + Jp[12 + a] * Jl[6 + k]) * sp[k];
      }
    }
  }
}

// block partials of a point pass: cost (sum) and gradient max-norm
__device__ __forceinline__ void store_point_partials(const BaDev &B, int buf, double cost, double gmax, double *lds,
                                                     double *lmx) {
  const int tid = threadIdx.x;
  double c[1] = {cost};
  block_sum<1>(c, lds);
  const double gm = wave_max(gmax);
  if ((tid & 63) == 0) lmx[tid >> 6] = gm;
  __syncthreads();
  if (tid == 0) {
    B.slab_pt[buf][2 * blockIdx.x] = c[0];
    B.slab_pt[buf][2 * blockIdx.x + 1] = fmax(fmax(lmx[0], lmx[1]), fmax(lmx[2], lmx[3]));
  }
}

// first linearisation of a solve, at the current state
__global__ __launch_bounds__(256) void k_ba_lin0(BaDev B) {
  __shared__ double lds[4];
  __shared__ double lmx[4];
  __shared__ double pcL[kPcLds * 12];
  const BaState st = *B.st;
  const int tid = threadIdx.x, g = tid & (kGroup - 1);
  const bool pc_lds = B.n_cams <= kPcLds;
  if (pc_lds)
    for (int i = tid; i < 12 * B.n_cams; i += 256) pcL[i] = B.PC[st.cur][i];
  __syncthreads();
  const int li = blockIdx.x * kPtsPerBlock + (tid >> kGroupLog);
  double cost = 0, gmax = 0;
  if (li < B.n_local) {
    const int j = B.local_pts[li];
    const double *Xp = B.Xp[st.cur];
    const double pt[3] = {Xp[3 * j], Xp[3 * j + 1], Xp[3 * j + 2]};
    point_linearize(B, st, j, g, pt, pc_lds ? pcL : B.PC[st.cur], st.cur, cost, gmax);
  }
  store_point_partials(B, st.cur, cost, gmax, lds, lmx);
}

// --------------------------------------------------------------------------------------------
// k_ba_cams: F^T F and F^T b of each free camera (the f-blocks of the Schur eliminator)
// --------------------------------------------------------------------------------------------
constexpr int kCamPitch = 265;  // LDS row pitch (doubles) of the transposed camera-block reduction
// kLdsReduce: the 27 block sums go through LDS transposed (lds >= 27 * kCamPitch doubles): 27 stores, then 216 threads
// add 32 terms each and finish over 8 lanes by DPP -- ~110 instructions per wavefront against ~700 for 27 butterfly
// wave sums, which is what the role's run time was made of.  Fixed order either way.
template <bool kLdsReduce>
__device__ __forceinline__ void ba_cams_role(const BaDev &B, const BaState &st, int slot, int chunk, double *lds) {
  const int tid = threadIdx.x;
  const int c = B.slot_cam[slot];
  const int s0 = B.cam_start[slot], s1 = B.cam_start[slot + 1];
  const double *Xp = B.Xp[st.cur];
  const PoseCache P = load_pc(B.PC[st.cur], c);
#ifdef VO_BA_STAMPS
  if (tid == 0 && slot == 0 && chunk == 0) B.dbg[32] = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) atomicMax(&B.dbg[42], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0;
  for (int idx = s0 + chunk * kCamChunk + tid; idx < min(s1, s0 + (chunk + 1) * kCamChunk); idx += kCamChunk) {
    const int e = B.cam_edges[idx];
    if (!B.e_active[e]) continue;
    const int j = B.e_pt[e];
    const double pt[3] = {Xp[3 * j], Xp[3 * j + 1], Xp[3 * j + 2]};
    double r[3], J[18];
    const int m = edge_eval<true, false>(P, pt, B.e_obs[3 * e], B.e_obs[3 * e + 1], B.e_obs[3 * e + 2], B.e_is[e],
                                         B.K, r, J, nullptr);
    double rho0, rho1;
    huber(m == 2 ? st.hm : st.hs, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho0, rho1);
    int t = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
      const double ja0 = J[a], ja1 = J[6 + a], ja2 = J[12 + a];
#pragma unroll
      for (int b = a; b < 6; b++) acc[t++] += rho1 * (ja0 * J[b] + ja1 * J[6 + b] + ja2 * J[12 + b]);
      acc[21 + a] += rho1 * (ja0 * r[0] + ja1 * r[1] + ja2 * r[2]);
    }
  }
#ifdef VO_BA_STAMPS
  {
    double dep = acc[0] + acc[26];
    const unsigned long long t = stamp_after(dep);
    if (dep == 1.2345e-300) acc[0] = 0;
    if (tid == 0 && slot == 0 && chunk == 0) B.dbg[33] = t;
    if ((tid & 63) == 0) atomicMax(&B.dbg[43], t);
  }
#endif
  double *o = B.slab_cam + ((long long)slot * B.n_cchunks + chunk) * 27;
  if (kLdsReduce) {
    const int col = tid + (tid >> 5);  // one pad per 32 columns: the 8 partial sums of an entry start in different banks
#pragma unroll
    for (int i = 0; i < 27; i++) lds[i * kCamPitch + col] = acc[i];
    __syncthreads();
    if (tid < 27 * 8) {
      const int i = tid >> 3, part = tid & 7;
      const double *src = lds + i * kCamPitch + part * 33;
      double v[32];
#pragma unroll
      for (int j = 0; j < 32; j++) v[j] = src[j];
      double sum = 0;
#pragma unroll
      for (int j = 0; j < 32; j++) sum += v[j];
      sum += dpp_f64<0xB1>(sum);   // quad_perm [1,0,3,2]
      sum += dpp_f64<0x4E>(sum);   // quad_perm [2,3,0,1]
      sum += dpp_f64<0x141>(sum);  // row_half_mirror: the other quad of the 8 lanes
      if (part == 0) o[i] = sum;
    }
  } else {
    block_sum<27, kCamChunk / 64>(acc, lds);
    if (tid == 0) {  // static indices only (a runtime-indexed acc[] would live in scratch memory)
#pragma unroll
      for (int i = 0; i < 27; i++) o[i] = acc[i];
    }
  }
#ifdef VO_BA_STAMPS
  if (tid == 0) atomicMax(&B.dbg[35], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// --------------------------------------------------------------------------------------------
// k_ba_gemm: partial tiles of  G = Y * W^T  with G[m][n] = sum_k Yt[k][m] * Wt[k][n] on the FP64
// matrix cores.  One 16x16 tile per wavefront per K-slice; A operand: lane l holds
// A[i = l&15][k = l>>4]; B operand: B[k = l>>4][j = l&15]; result: 4 doubles per lane at
// row (l>>4) + 4*reg, column l&15.
// --------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kChunkPts = 512;  // points per K-slice (LDS table of their damped inverses)

#ifdef VO_BA_STAMPS
#define STAMP(i) do { if (threadIdx.x == 0) B.dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP_DRAIN(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(i); } while (0)
#else
#define STAMP(i)
#define STAMP_DRAIN(i)
#endif
#ifdef VO_BA_STAMPS
#define STAMP0(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) B.dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// phase accumulators kept in registers (no store inside the timed loop)
#define PHASE_DECL unsigned long long ph_acc[5] = {0, 0, 0, 0, 0}, ph_prev = __builtin_amdgcn_s_memrealtime()
#define PHASE(i) do { const unsigned long long ph_t = __builtin_amdgcn_s_memrealtime(); ph_acc[i] += ph_t - ph_prev; ph_prev = ph_t; } while (0)
#define PHASE_STORE(base) do { if (threadIdx.x == 0) for (int ph_i = 0; ph_i < 5; ph_i++) B.dbg[(base) + ph_i] = ph_acc[ph_i]; } while (0)
#else
#define PHASE_DECL
#define PHASE(i)
#define PHASE_STORE(base)
#define STAMP0(i)
#endif
constexpr int kGemmWaves = kGemmThreads / 64;
constexpr int kGemmLdsTile = kGemmWaves * 256 + kChunkPts * 6;  // part, hinvL
static_assert(kChunkPts <= kGemmThreads, "ba_gemm_tile_role takes one point per thread");
constexpr int kGemmLdsDoubles = kGemmLdsTile > 27 * kCamPitch ? kGemmLdsTile : 27 * kCamPitch;  // or the camera role's 27 rows

__device__ __forceinline__ void ba_gemm_tile_role(const BaDev &B, double *sm, int ntiles, int tdim, int bid);

// Schur product tiles and camera blocks in one launch (independent roles).  Every block writes its
// partial result to a slab; the consumer sits behind the kernel boundary (k_ba_solve when the problem
// lives on one GPU, k_ba_reduce + all-reduce when it is sharded).  In-kernel hand-offs were measured
// and lose: an sc1 store -> ticket -> sc1 load chain costs ~8 us on MI355X and a single block pulls
// coherent loads at only ~12 GB/s, against ~5 us for a kernel boundary followed by cached loads.
__global__ __launch_bounds__(kGemmThreads) void k_ba_gemm(BaDev B) {
  extern __shared__ double sm[];  // part | hinvL | lds27
  STAMP0(16);
  const int tdim = B.Mpad / 16, ntiles = tdim * (tdim + 1) / 2;
  // The camera-block role (independent of the tiles, and the longer of the two) takes the FIRST workgroup indices: with
  // it behind the 240 tile blocks its last workgroup was observed to start 8.7 us into the kernel.
  const int ncam = B.nf * B.n_cchunks;
  if ((int)blockIdx.x < ncam) {
    const BaState st = *B.st;
    if (st.done) return;
    if (threadIdx.x >= kCamChunk) return;  // retired wavefronts do not take part in the role's barriers
    const int q = blockIdx.x;
    ba_cams_role<true>(B, st, q / B.n_cchunks, q % B.n_cchunks, sm);  // the role's blocks use none of the tile buffers
  } else {
    ba_gemm_tile_role(B, sm, ntiles, tdim, (int)blockIdx.x - ncam);  // reads the state itself, together with what does not depend on it
  }
  STAMP0(20);
}

__device__ __forceinline__ void ba_gemm_tile_role(const BaDev &B, double *sm, int ntiles, int tdim, int bid) {
  const BaState st = *B.st;  // issued first: waiting for it leaves the loads below in flight
  double(*part)[256] = reinterpret_cast<double(*)[256]>(sm);
  double *hinvL = sm + kGemmWaves * 256;
  const int ks = bid / ntiles;
  int tile = bid - ks * ntiles, tm = 0;
  while (tile >= tdim - tm) {  // upper-triangular tile index -> (tm <= tn)
    tile -= tdim - tm;
    tm++;
  }
  const int tn = tm + tile;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int K = 3 * B.n_pts;
  const int kb0 = ks * B.kchunk, kb1 = min(K, kb0 + B.kchunk);  // kchunk is a multiple of 48: whole points
  // ---- damped inverse of every point block of this K-slice (SchurEliminator: (E^T E + D)^-1).
  // Each tile block needs them for its own A operand; the tile-0 block also publishes them for the
  // back-substitution.
  const int j0 = kb0 / 3, npts = (kb1 - kb0) / 3;  // <= kChunkPts = kGemmThreads: one point per thread
  // The point blocks of BOTH linearisation buffers and the scales are requested before the state is looked at (which
  // buffer is current is the only thing the state decides here): one round trip instead of two.
  const int jp = j0 + min(tid, max(npts - 1, 0));
  double hb[2][6], spv[3];
#pragma unroll
  for (int i = 0; i < 6; i++) hb[0][i] = B.hll[0][6 * jp + i], hb[1][i] = B.hll[1][6 * jp + i];
#pragma unroll
  for (int i = 0; i < 3; i++) spv[i] = B.scale_p[3 * jp + i];
  if (st.done) return;
  const double *W = B.Wt[st.cur];
  if (tid < npts) {
    const int t = tid, j = j0 + t;
    const double sp0 = spv[0], sp1 = spv[1], sp2 = spv[2];
    double hl6[6];
#pragma unroll
    for (int i = 0; i < 6; i++) hl6[i] = st.cur ? hb[1][i] : hb[0][i];
    double hs[6] = {hl6[0] * sp0 * sp0, hl6[1] * sp0 * sp1, hl6[2] * sp0 * sp2,
                    hl6[3] * sp1 * sp1, hl6[4] * sp1 * sp2, hl6[5] * sp2 * sp2};
    const double d0 = fmin(fmax(hs[0], 1e-6), 1e32) / st.radius;
    const double d1 = fmin(fmax(hs[3], 1e-6), 1e32) / st.radius;
    const double d2 = fmin(fmax(hs[5], 1e-6), 1e32) / st.radius;
    hs[0] += d0, hs[3] += d1, hs[5] += d2;
    double hi[6];
    if (!inv3_sym(hs, hi)) {
      hi[0] = hi[3] = hi[5] = 0.0 / 0.0;  // poisons the step => invalid step handling
      hi[1] = hi[2] = hi[4] = 0;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) hinvL[6 * t + i] = hi[i];
    if (tm == 0 && tn == 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) B.hinv[6 * j + i] = hi[i];
      B.dl[3 * j] = d0, B.dl[3 * j + 1] = d1, B.dl[3 * j + 2] = d2;
#pragma unroll
      for (int k = 0; k < 3; k++) B.gl2[3 * j + k] = W[(long long)(3 * j + k) * B.Mpad + 6 * B.nf];
    }
  }
  __syncthreads();
  STAMP0(17);
  const int q = ((kb1 - kb0 + 4 * kGemmWaves - 1) / (4 * kGemmWaves)) * 4;  // rows per wave, multiple of 4
  const int k0 = kb0 + wave * q, k1 = min(kb1, k0 + q);
  double4_t acc = {0, 0, 0, 0};
  const int kk = lane >> 4, ii = lane & 15;
  const double *Wa = W + tm * 16 + ii, *Wb = W + tn * 16 + ii;
  // A[m][3j+c] = sum_c' W[3j+c'][m] * Hinv_j[c'][c]  is formed on the fly (Y is never stored)
  auto sym = [](const double *hv, int r, int c) {  // packed upper {00,01,02,11,12,22}
    const int a = r < c ? r : c, b = r < c ? c : r;
    return hv[a == 0 ? b : (a == 1 ? 2 + b : 5)];
  };
  int k = k0;
  // The loop is latency-bound (one wave per SIMD): 12 MFMA steps per trip keep 48 independent loads in flight per
  // lane, all issued before the first multiply.  (96-row trips are slower, 16.7 against 14.2 us for the kernel: vmcnt
  // counts at most 63 outstanding vector-memory operations per wave.)
  auto trip = [&](auto uc, int kt) {
    constexpr int U = decltype(uc)::value;
    double w0[U], w1[U], w2[U], b[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int kr = kt + 4 * u + kk;
      const long long r3 = (long long)(3 * (kr / 3)) * B.Mpad;
      w0[u] = Wa[r3], w1[u] = Wa[r3 + B.Mpad], w2[u] = Wa[r3 + 2 * B.Mpad];
      b[u] = Wb[(long long)kr * B.Mpad];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int kr = kt + 4 * u + kk;
      const int j = kr / 3, c = kr - 3 * j;
      const double *hv = &hinvL[6 * (j - j0)];
      const double a = w0[u] * sym(hv, 0, c) + w1[u] * sym(hv, 1, c) + w2[u] * sym(hv, 2, c);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[u], acc, 0, 0, 0);
    }
  };
  for (; k + 48 <= k1; k += 48) trip(std::integral_constant<int, 12>{}, k);
  for (; k + 16 <= k1; k += 16) {  // 4 MFMA steps: 16 independent loads in flight per lane
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int kr = k + 4 * u + kk;
      const int j = kr / 3, c = kr - 3 * j;
      const double *hv = &hinvL[6 * (j - j0)];
      const long long r3 = (long long)(3 * j) * B.Mpad;
      const double w0 = Wa[r3], w1 = Wa[r3 + B.Mpad], w2 = Wa[r3 + 2 * B.Mpad];
      a[u] = w0 * sym(hv, 0, c) + w1 * sym(hv, 1, c) + w2 * sym(hv, 2, c);
      b[u] = Wb[(long long)kr * B.Mpad];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  for (; k < k1; k += 4) {
    const int kr = k + kk;
    double a = 0, b = 0;
    if (kr < k1) {
      const int j = kr / 3, c = kr - 3 * j;
      const double *hv = &hinvL[6 * (j - j0)];
      const long long r3 = (long long)(3 * j) * B.Mpad;
      a = Wa[r3] * sym(hv, 0, c) + Wa[r3 + B.Mpad] * sym(hv, 1, c) + Wa[r3 + 2 * B.Mpad] * sym(hv, 2, c);
      b = Wb[(long long)kr * B.Mpad];
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  STAMP0(18);
#ifdef VO_BA_STAMPS
  if (threadIdx.x == 0) atomicMax(&B.dbg[23], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
#pragma unroll
  for (int r = 0; r < 4; r++) part[wave][(kk + 4 * r) * 16 + ii] = acc[r];
  __syncthreads();
  const int t = threadIdx.x & 255;  // element (t>>4, t&15) of the tile, fixed summation order (the upper half of the
                                    // block computes the same value and stores nothing)
  double v = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  if (kGemmWaves == 8) v += (part[4][t] + part[5][t]) + (part[6][t] + part[7][t]);
  const bool writer = threadIdx.x < 256;
  const long long eoff = (long long)(tm * 16 + (t >> 4)) * B.Mpad + tn * 16 + (t & 15);
  if (!B.fused) {  // sharded: k_ba_reduce sums the slabs into the payload that is all-reduced
    if (writer) B.slab_gemm[(long long)ks * B.Mpad * B.Mpad + eoff] = v;
    return;
  }
  // One GPU: the last K-slice block of a tile to arrive sums the tile's slabs (slab order:
  // deterministic) into the payload.  This hand-off (~8 us: write-through stores, ticket, coherent
  // loads) runs in parallel over the tiles and overlaps the longer camera-block role; the single
  // solving block behind the kernel boundary then reads 12 KB instead of ksplit x 12 KB.
  if (writer) st_sc1(&B.slab_gemm[(long long)ks * B.Mpad * B.Mpad + eoff], v);
  __shared__ int s_last;
  if (!arrive_and_check_last(&B.counters[1 + (tm * tdim + tn)], (unsigned)B.ksplit, &s_last)) return;
#ifdef VO_BA_STAMPS
  if (threadIdx.x == 0) atomicMax(&B.dbg[22], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
  if (!writer) return;  // no barrier below
  const long long M2 = (long long)B.Mpad * B.Mpad;
  double sv[32];
#pragma unroll
  for (int u = 0; u < 32; u++) sv[u] = ld_sc1(&B.slab_gemm[min(u, B.ksplit - 1) * M2 + eoff]);  // all in flight
  double sum = 0;
#pragma unroll
  for (int u = 0; u < 32; u++) sum += u < B.ksplit ? sv[u] : 0.0;
  if (writer) B.payload[eoff] = sum;
#ifdef VO_BA_STAMPS
  if (threadIdx.x == 0) atomicMax(&B.dbg[21], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// fixed-order reduction of the slabs into the payload that a multi-GPU run all-reduces
__global__ __launch_bounds__(256) void k_ba_reduce(BaDev B) {
  const BaState st = *B.st;
  if (st.done) return;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int i = gid >> 2, sub = gid & 3;  // 4 lanes share one payload entry
  const int nG = B.Mpad * B.Mpad, nH = B.nf * 27;
  double s = 0;
  if (i < nG) {
    const int r = i / B.Mpad, c = i - r * B.Mpad;
    if ((c >> 4) >= (r >> 4)) {  // upper tiles only
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {  // ksplit <= 32: up to 8 slabs per lane, all loads in flight together
        const int k = sub + 4 * u;
        v[u] = k < B.ksplit ? B.slab_gemm[(long long)k * nG + i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) s += v[u];
    }
  } else if (i < nG + nH) {
    const int q = i - nG, slot = q / 27, t = q - slot * 27;
    for (int c = sub; c < B.n_cchunks; c += 4) s += B.slab_cam[((long long)slot * B.n_cchunks + c) * 27 + t];
  }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  if (i < nG + nH) {
    if (sub == 0) B.payload[i] = s;
  } else if (i == nG + nH && sub == 0) {
    double cs = 0, m = 0;
    const double *sp = B.slab_pt[st.cur];
    for (int b = 0; b < B.n_pblocks; b++) {
      cs += sp[2 * b];
      m = fmax(m, sp[2 * b + 1]);
    }
    B.payload[i] = cs;
    for (int k = 0; k < B.n_shards; k++) B.payload[i + 1 + k] = (k == B.shard) ? m : 0.0;
  }
}

// --------------------------------------------------------------------------------------------
// k_ba_solve: reduced camera system S y = rhs (DenseSchurComplementSolver: dense Cholesky),
// camera step, candidate poses.  One workgroup; S lives in LDS.
// --------------------------------------------------------------------------------------------
// fast FP64 reciprocal: hardware estimate + two Newton steps (dependent chain of ~5 instructions
// instead of the ~12 of an IEEE division; relative error < 2^-50, far below the LM tolerances)
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = r * (2.0 - d * r);
  r = r * (2.0 - d * r);
  return r;
}

// 6x6 L D L^T of a diagonal block held in registers (packed lower triangle, row-major, 21).
// On return the strict lower part holds the unit-lower factor, the diagonal holds d_j and
// rd[j] = 1/d_j.  Returns false on a non-positive pivot.  Square-root free: the pivot chain is
// the latency floor of the whole solve.
__device__ __forceinline__ bool ldl6_packed(double L[21], double rd[6]) {
  bool ok = true;
  double u[21];  // u_ij = l_ij d_j
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double d = L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= u[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
    ok = ok && (d > 0.0) && isfinite(d);
    const double r = fast_rcp(d);
    L[j * (j + 1) / 2 + j] = d;
    rd[j] = r;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double v = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; k++) v -= u[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      u[i * (i + 1) / 2 + j] = v;
      L[i * (i + 1) / 2 + j] = v * r;
    }
  }
  return ok;
}

constexpr int kSolveThreads = 512;  // two wavefronts per SIMD: a lone wavefront issues one instruction per ~8 cycles
constexpr int kSolveRed = 16 * 5 + 48;  // block_sum<5> of up to 16 wavefronts
__device__ __forceinline__ void ba_solve_body(const BaDev &B, double *sm) {
  BaState *S = B.st;
  const int tid = threadIdx.x;
  constexpr int NT = kSolveThreads;
  // The state and everything whose address does not depend on it (the reduced system, the camera-block slabs) are
  // requested together: one round trip instead of two before the first barrier (~1.3 us).  The state's load is issued
  // first, so waiting for it (vmcnt counts in order) leaves the others in flight; only the poses need st.cur.
  const BaState st0 = *S;
  STAMP(0);
#ifdef VO_BA_STAMPS
  if (tid == 0) B.dbg[30] = __builtin_readcyclecounter();
#endif
  const int nb = B.nf, n = 6 * nb, ld = n + 1;
  double *A = sm;                   // (n+1) x ld: lower triangle of S'' in rows 0..n-1, rhs'' in row n
  double *sc = A + (n + 1) * ld;    // Jacobi scale
  double *Dd = sc + n;              // LM diagonal
  double *gpp = Dd + n;             // scaled gradient g''
  double *y = gpp + n;              // solution
  double *Ldg = y + n;              // nb x 21 factored diagonal blocks
  double *red = Ldg + nb * 21 + n;  // kSolveRed scratch (after the n reciprocal pivots)
  __shared__ int s_fail, s_stop;
  const double *G = B.payload;
  // camera blocks / cost / gradient-max: all-reduced payload in the sharded mode, summed into LDS here otherwise
  double *HPw = B.fused ? red + kSolveRed : B.payload + payload_hpp_off(B);
  // this thread's camera for the candidate-pose phase: pose and cache are read early, used after the solve
  const int pc_cam = tid < B.n_cams ? tid : 0;
  const int slotpre = B.cam_slot[pc_cam], cinpre = B.cam_in[pc_cam] == B.epoch;
  Se3 expx;
  double xpre[6];
  PoseCache pcpre;
  int cur0, first;
  double radius;
  // Raw Schur product into A (lower triangle, rhs in row n): entry (c, r >= c) of the payload --
  // all-reduced across GPUs (sharded) or summed over the K slices by k_ba_gemm's tile blocks (fused).
  // Unconditional loads, eight in flight per lane: lanes without an entry read the all-zero slab.
  // (A `cond ? load : 0` form is sunk into a branch by the compiler and followed by
  // s_waitcnt vmcnt(0): every load of the round would be serialised.)
  {
    const double *zero_slab = B.slab_gemm + (long long)B.ksplit * B.Mpad * B.Mpad;  // never written after creation
    // Only the entries that exist are enumerated (instruction issue, not latency, bounds this prologue): the packed
    // lower triangle row by row, e = r (r + 1) / 2 + c for r < n, then the rhs row n.
    const int ntri = n * (n + 1) / 2, ne = ntri + n;
    constexpr int RU = 4;  // entries per thread and round
    auto issue = [&](int e0, double (&v)[RU], int (&pos)[RU]) {
#pragma unroll
      for (int un = 0; un < RU; un++) {
        const int e = e0 + un * NT + tid;
        int r = (int)((__fsqrt_rn(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        r += ((r + 1) * (r + 2) / 2 <= e) ? 1 : 0;
        r -= (r * (r + 1) / 2 > e) ? 1 : 0;
        int c = e - r * (r + 1) / 2;
        if (e >= ntri) r = n, c = e - ntri;
        const bool valid = e < ne;
        pos[un] = valid ? r * ld + c : -1;
        v[un] = *(valid ? G + (long long)c * B.Mpad + r : zero_slab);
      }
    };
    auto commit = [&](const double (&v)[RU], const int (&pos)[RU]) {
#pragma unroll
      for (int un = 0; un < RU; un++)
        if (pos[un] >= 0) A[pos[un]] = v[un];
    };
    double v0[RU];
    int pos0[RU];
    issue(0, v0, pos0);
    // single shard (no k_ba_reduce): the camera-block slabs of k_ba_gemm's camera role are summed here, one entry
    // per thread, sixteen chunk loads in flight together with the round above
    const double *zslab = zero_slab;
    const int ci = tid < B.nf * 27 ? tid : 0;
    const int cslot = ci / 27;
    const double *sp = B.slab_cam + (long long)cslot * B.n_cchunks * 27 + (ci - cslot * 27);
    double cv[16];
    const bool slab_thread = B.fused && tid < B.nf * 27;  // whole wavefronts without an entry skip the sixteen loads
    if (slab_thread) {
#pragma unroll
      for (int q = 0; q < 16; q++) cv[q] = *(q < B.n_cchunks ? sp + q * 27 : zslab);
    }
    if (st0.done) return;  // (nothing has been written yet)
    cur0 = st0.cur, first = st0.first, radius = st0.radius;
#pragma unroll
    for (int a = 0; a < 6; a++) xpre[a] = B.Xc[cur0][6 * pc_cam + a];
    pcpre = load_pc(B.PC[cur0], pc_cam);
    commit(v0, pos0);
    for (int e0 = NT * RU; e0 < ne; e0 += NT * RU) {
      double v[RU];
      int pos[RU];
      issue(e0, v, pos);
      commit(v, pos);
    }
    if (slab_thread) {
      double a = 0;
#pragma unroll
      for (int q = 0; q < 16; q++) a += cv[q];
      for (int c0 = 16; c0 < B.n_cchunks; c0 += 16) {  // more than 16 chunks per camera: further rounds, chunk order
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = *(c0 + q < B.n_cchunks ? sp + (c0 + q) * 27 : zslab);
#pragma unroll
        for (int q = 0; q < 16; q++) a += v[q];
      }
      HPw[tid] = a;
    }
    // exp(x) of this thread's camera: se3_plus then only has exp(delta), the product and the log left to do
    if (tid < B.n_cams) expx = se3_exp(xpre);
  }
  STAMP_DRAIN(10);
  STAMP_DRAIN(11);
  const double sc_pre = (first || tid >= n) ? 0.0 : B.scale_c[tid];  // per reduced-system column
  STAMP_DRAIN(8);  // debug builds only: the prefetch round trip
  if (B.fused) {
    for (int i = tid + NT; i < B.nf * 27; i += NT) {  // nf * 27 > NT never happens (nf <= 21): kept for safety
      const int slot = i / 27, t = i - slot * 27;
      const double *sp2 = B.slab_cam + (long long)slot * B.n_cchunks * 27 + t;
      double a = 0;
      for (int c0 = 0; c0 < B.n_cchunks; c0++) a += sp2[c0 * 27];
      HPw[i] = a;
    }
    if (tid >= NT - 64) {  // last wave: cost (sum) and gradient max over the point blocks
      const int l = tid - (NT - 64);
      double cs = 0, m = 0;
      const double *spt = B.slab_pt[cur0];
      for (int b = l; b < B.n_pblocks; b += 64) {
        cs += spt[2 * b];
        m = fmax(m, spt[2 * b + 1]);
      }
      cs = wave_sum(cs);
      m = wave_max(m);
      if (l == 0) {
        HPw[B.nf * 27] = cs;
        HPw[B.nf * 27 + 1] = m;
      }
    }
    __syncthreads();
  }
  STAMP(9);
  const double *HP = HPw;
  if (tid == 0) s_fail = 0, s_stop = 0;
  double gm = 0;
  for (int i = tid; i < n; i += NT) {
    const int slot = i / 6, a = i - slot * 6;
    int t = 0;
    for (int q = 0; q < a; q++) t += 6 - q;  // index of (a,a) in the packed upper triangle
    const double hd = HP[slot * 27 + t];
    double s;
    if (first) {  // Jacobi scaling of the camera columns from the first linearisation
      s = 1.0 / (1.0 + sqrt(hd));
      B.scale_c[i] = s;
    } else {
      s = i == tid ? sc_pre : B.scale_c[i];
    }
    sc[i] = s;
    Dd[i] = fmin(fmax(hd * s * s, 1e-6), 1e32) / radius;
    const double gp = HP[slot * 27 + 21 + a];
    gm = fmax(gm, fabs(gp));
    gpp[i] = s * gp;
  }
  __syncthreads();
  STAMP(1);
  // S'' = diag(sc) (Hpp - Y W^T) diag(sc) + D in place on the raw product; rhs'' = g'' - sc * (Y g_l)
  {
    const int tx = tid & 15, ty = tid >> 4;
    for (int c = ty; c < n; c += NT / 16)
      for (int r = c - (c & 15) + tx; r < n; r += 16) {
        if (r < c) continue;
        double v = -A[r * ld + c];
        if (r / 6 == c / 6) {
          const int slot = r / 6, a = c % 6, b = r % 6;  // a <= b
          int t = 0;
          for (int q = 0; q < a; q++) t += 6 - q;
          v += HP[slot * 27 + t + (b - a)];
        }
        v *= sc[r] * sc[c];
        if (r == c) v += Dd[r];
        A[r * ld + c] = v;
      }
  }
  for (int i = tid; i < n; i += NT) A[n * ld + i] = gpp[i] - sc[i] * A[n * ld + i];
  gm = wave_max(gm);
  if ((tid & 63) == 0) red[tid >> 6] = gm;
  __syncthreads();
  if (tid == 0) {
    double m = 0;
    for (int w = 0; w < NT / 64; w++) m = fmax(m, red[w]);
    const double *CP = HP + B.nf * 27;  // cost, then one gradient-max slot per shard
    for (int k = 0; k < B.n_shards; k++) m = fmax(m, CP[1 + k]);
    S->gmax = m;
    const double cost = CP[0];
    S->x_cost = cost;
    if (first) S->initial_cost = cost;
    // FinalizeIterationAndCheckIfMinimizerCanContinue of the previous iteration
    if (st0.last_ok && m <= 1e-10) {
      S->termination = 3;
      S->done = 1;
      s_stop = 1;
    } else {
      S->iter = st0.iter + 1;
    }
    S->first = 0;
  }
  __syncthreads();
  if (s_stop) return;
  STAMP(2);
  // Blocked (6x6) right-looking L D L^T with the rhs carried as row n (forward substitution for
  // free).  Rows below the diagonal block store u_rt = l_rt d_t.  Per block column: every thread
  // factors the diagonal block redundantly in registers (no barrier needed for it), one thread per
  // panel row does its substitution, barrier, rank-6 trailing update, barrier.
  // (Measured alternative: the trailing update as 16x16 tiles on the FP64 matrix cores, K = 6 padded
  // to two v_mfma_f64_16x16x4 steps.  At n = 54 it is slower -- 15.0 us against 11.3 us for the whole
  // factorisation: whole tiles do not shrink with the trailing matrix and the panel rows must be
  // published twice -- so the scalar form stays; the pivot chain and the panel substitution, not the
  // update, are the latency floor.)
  double *rdv = Ldg + nb * 21;  // 1/d_j of all n pivots
  // Look-ahead: while wavefronts 1.. update the trailing matrix with block column k, wavefront 0 updates only the next
  // diagonal block, factors it and publishes L / 1/d in LDS, so the ~0.4 us pivot chain of ldl6 is off the critical
  // path of every block column but the first.  Same operations on the same operands as the plain right-looking form.
  PHASE_DECL;
  const int wave = tid >> 6;
  if (nb > 0) {  // (all cameras fixed: a points-only problem has no reduced system)
    double L0[21], rd0[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) L0[i * (i + 1) / 2 + j] = A[i * ld + j];
    const bool ok0 = ldl6_packed(L0, rd0);  // every thread, redundantly: nothing to wait for
    if (tid == 0) {
      if (!ok0) s_fail = 1;
#pragma unroll
      for (int i = 0; i < 21; i++) Ldg[i] = L0[i];
#pragma unroll
      for (int i = 0; i < 6; i++) rdv[i] = rd0[i];
    }
    __syncthreads();
  }
  for (int k = 0; k < nb; k++) {
    const int K0 = 6 * k, R0 = K0 + 6;
    double L[21], rd[6];
    if (R0 + (tid & ~63) <= n) {  // wavefronts that own a panel row (LDS bandwidth is per CU: no redundant loads)
#pragma unroll
      for (int i = 0; i < 21; i++) L[i] = Ldg[k * 21 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) rd[i] = rdv[K0 + i];
    PHASE(0);
    for (int r = R0 + tid; r <= n; r += NT) {  // u_rt = A[r][t] - sum_{q<t} u_rq l_tq
      double x[6];
#pragma unroll
      for (int t = 0; t < 6; t++) {
        double v = A[r * ld + K0 + t];
#pragma unroll
        for (int q = 0; q < t; q++) v -= x[q] * L[t * (t + 1) / 2 + q];
        x[t] = v;
      }
#pragma unroll
      for (int t = 0; t < 6; t++) A[r * ld + K0 + t] = x[t];
    }
    PHASE(2);
    __syncthreads();
    PHASE(3);
    if (wave == 0) {
      if (k + 1 < nb) {  // rows R0 .. R0+5 of the trailing matrix are exactly the next diagonal block
        int i = 0, j = 0;  // lane e < 21 -> (i, j), j <= i
        {
          int e = tid < 21 ? tid : 0;
          while (e > i) e -= ++i;
          j = e;
        }
        const int r = R0 + i, c = R0 + j;
        double acc = 0;
#pragma unroll
        for (int t = 0; t < 6; t++) acc += (A[r * ld + K0 + t] * rd[t]) * A[c * ld + K0 + t];
        if (tid < 21) A[r * ld + c] -= acc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double Ln[21], rdn[6];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int bq = 0; bq <= a; bq++) Ln[a * (a + 1) / 2 + bq] = A[(R0 + a) * ld + R0 + bq];
        PHASE(1);
        const bool okn = ldl6_packed(Ln, rdn);
        if (tid == 0) {
          if (!okn) s_fail = 1;
#pragma unroll
          for (int a = 0; a < 21; a++) Ldg[(k + 1) * 21 + a] = Ln[a];
#pragma unroll
          for (int a = 0; a < 6; a++) rdv[R0 + a] = rdn[a];
        }
      }
    } else {
      const int tt = tid - 64, tx = tt & 15, ty = tt >> 4;
      for (int r = R0 + 6 + ty; r <= n; r += (NT - 64) / 16) {
        double w[6];
#pragma unroll
        for (int t = 0; t < 6; t++) w[t] = A[r * ld + K0 + t] * rd[t];
        for (int c = R0 + tx; c <= r && c < n; c += 16) {
          double acc = 0;
#pragma unroll
          for (int t = 0; t < 6; t++) acc += w[t] * A[c * ld + K0 + t];
          A[r * ld + c] -= acc;
        }
      }
    }
    PHASE(4);
    __syncthreads();
    PHASE(3);
  }
  PHASE_STORE(48);
  STAMP(3);
  // back substitution  L^T y = D^-1 w  (w = row n), block by block from the bottom; row n keeps
  // w_i - sum_{r>i} u_ri y_r and is scaled by 1/d_i when its block is solved
  if (n < 64) {
    // One wavefront, lane = column, no barrier: w_i lives in a register, the pivots' entries come by v_readlane, and the
    // next block's operands (L, 1/d, the six u_ri of this lane) are in flight while the current block is solved.
    if (wave == 0) {
      const int i = tid;
      double wi = A[n * ld + (i < n ? i : 0)], yi = 0;
      auto loadblk = [&](int kb, double (&Lb)[21], double (&rb)[6], double (&ab)[6]) {
        const int Kb = 6 * kb, ii = i < Kb ? i : 0;
#pragma unroll
        for (int q = 0; q < 21; q++) Lb[q] = Ldg[kb * 21 + q];
#pragma unroll
        for (int t = 0; t < 6; t++) rb[t] = rdv[Kb + t], ab[t] = A[(Kb + t) * ld + ii];
      };
      auto step = [&](int kb, const double (&Lc)[21], const double (&rc)[6], const double (&ac)[6], double (&Ln)[21],
                      double (&rn)[6], double (&an)[6]) {
        const int Kb = 6 * kb;
        loadblk(kb > 0 ? kb - 1 : 0, Ln, rn, an);
        double yk[6];
#pragma unroll
        for (int t = 5; t >= 0; t--) {
          double v = readlane_f64(wi, Kb + t) * rc[t];
#pragma unroll
          for (int q = t + 1; q < 6; q++) v -= Lc[q * (q + 1) / 2 + t] * yk[q];
          yk[t] = v;
        }
#pragma unroll
        for (int t = 0; t < 6; t++) yi = i == Kb + t ? yk[t] : yi;
        double acc = 0;
#pragma unroll
        for (int t = 0; t < 6; t++) acc += ac[t] * yk[t];
        wi = i < Kb ? wi - acc : wi;
      };
      double LA[21], rA[6], aA[6], LB[21], rB[6], aB[6];
      int kb = nb - 1;
      if (kb >= 0) {
        loadblk(kb, LA, rA, aA);
        for (;;) {
          step(kb, LA, rA, aA, LB, rB, aB);
          if (--kb < 0) break;
          step(kb, LB, rB, aB, LA, rA, aA);
          if (--kb < 0) break;
        }
      }
      if (i < n) y[i] = yi;
    }
    __syncthreads();
  } else
  for (int k = nb - 1; k >= 0; k--) {
    const int K0 = 6 * k;
    double L[21], yk[6];
#pragma unroll
    for (int i = 0; i < 21; i++) L[i] = Ldg[k * 21 + i];
#pragma unroll
    for (int t = 5; t >= 0; t--) {
      double v = A[n * ld + K0 + t] * rdv[K0 + t];
#pragma unroll
      for (int q = t + 1; q < 6; q++) v -= L[q * (q + 1) / 2 + t] * yk[q];
      yk[t] = v;
    }
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) y[K0 + i] = yk[i];
    }
    for (int i = tid; i < K0; i += NT) {
      double acc = 0;
#pragma unroll
      for (int t = 0; t < 6; t++) acc += A[(K0 + t) * ld + i] * yk[t];
      A[n * ld + i] -= acc;
    }
    __syncthreads();
  }
  STAMP(4);
  // camera part of  g''.step  and  step^T D step  (step = -y)
  double gdot = 0, dquad = 0;
  int bad = 0;
  for (int i = tid; i < n; i += NT) {
    const double stp = -y[i];
    if (!isfinite(stp)) bad = 1;
    gdot += gpp[i] * stp;
    dquad += Dd[i] * stp * stp;
    B.zc[i] = sc[i] * y[i];
  }
  if (bad) s_fail = 1;
  __syncthreads();
  const int failed = s_fail;
  STAMP(5);
  // candidate poses (PoseLocalParameterization::Plus) + their caches + norms
  const double *X = B.Xc[cur0];
  double *Xn = B.Xc[cur0 ^ 1];
  double xn2 = 0, cn2 = 0, sn2 = 0;
  for (int c = tid; c < B.n_cams; c += NT) {
    const bool pre = c == tid;  // first round: operands were prefetched at the top of the kernel
    const int slot = pre ? slotpre : B.cam_slot[c];
    const int cin = pre ? cinpre : (int)(B.cam_in[c] == B.epoch);
    double x0[6], xc[6];
#pragma unroll
    for (int a = 0; a < 6; a++) x0[a] = pre ? xpre[a] : X[6 * c + a];
    if (slot >= 0 && !failed) {
      double d[6];
#pragma unroll
      for (int a = 0; a < 6; a++) d[a] = -y[6 * slot + a] * sc[6 * slot + a];
      if (pre)
        se3_plus_exp(expx, d, xc);
      else
        se3_plus(x0, d, xc);
      store_pc(B.PC[cur0 ^ 1], c, pose_cache(xc));
    } else {
#pragma unroll
      for (int a = 0; a < 6; a++) xc[a] = x0[a];
      store_pc(B.PC[cur0 ^ 1], c, pre ? pcpre : load_pc(B.PC[cur0], c));
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
      Xn[6 * c + a] = xc[a];
      if (slot >= 0 && cin) {
        xn2 += x0[a] * x0[a];
        cn2 += xc[a] * xc[a];
        sn2 += (xc[a] - x0[a]) * (xc[a] - x0[a]);
      }
    }
  }
  STAMP(6);
  double v5[5] = {gdot, dquad, xn2, cn2, sn2};
  if (n <= 64 && B.n_cams <= 64) {  // only wavefront 0 holds non-zero terms: no LDS pass, no barrier
    if (tid < 64) {
#pragma unroll
      for (int i = 0; i < 5; i++) v5[i] = wave_sum(v5[i]);
    }
  } else {
    block_sum<5, kSolveThreads / 64>(v5, red);
  }
  if (tid == 0) {
    S->gdot_c = v5[0];
    S->dquad_c = v5[1];
    S->x_norm2_c = v5[2];
    S->cand_norm2_c = v5[3];
    S->step_norm2_c = v5[4];
    S->solve_failed = failed;
  }
  STAMP(7);
#ifdef VO_BA_STAMPS
  if (tid == 0) B.dbg[31] = __builtin_readcyclecounter();
#endif
}

__global__ __launch_bounds__(kSolveThreads) void k_ba_solve(BaDev B) {
  extern __shared__ double sm[];
  ba_solve_body(B, sm);
}

// ============================================================================================
// Large reduced camera systems (6 nf + 1 > kMaxN: a global BA over hundreds of key-frames).
// The linearisation and the back-substitution are the kernels above (per-edge W blocks instead of
// the dense operand matrix); the Schur complement is gathered per camera pair and the dense ld x ld
// system goes through the blocked Cholesky of csrc/pose_graph.hip.
//   k_ba_hinv_large  damped inverses of the point blocks for the current radius
//   k_ba_cams_large  camera blocks Hpp / gp (same role as inside k_ba_gemm)
//   k_ba_pairs       one wavefront per covisible camera pair (c <= c'): the 6x6 block
//                    sum_j (W_cj Hll_j^-1) W_c'j^T over the shared points, lanes run over the points,
//                    fixed-shape butterfly at the end (deterministic, no atomics)
//   k_ba_prestep_large / k_ba_assemble_large / k_ba_poststep_large   what k_ba_solve does around
//                    its factorisation, split around the grid-wide Cholesky
// ============================================================================================
__global__ __launch_bounds__(256) void k_ba_hinv_large(BaDev B) {
  const BaState st = *B.st;
  if (st.done) return;
  const int li = blockIdx.x * 256 + threadIdx.x;
  if (li >= B.n_local) return;
  const int j = B.local_pts[li];
  const double *hl = B.hll[st.cur];
  const double sp0 = B.scale_p[3 * j], sp1 = B.scale_p[3 * j + 1], sp2 = B.scale_p[3 * j + 2];
  double hs[6] = {hl[6 * j] * sp0 * sp0, hl[6 * j + 1] * sp0 * sp1, hl[6 * j + 2] * sp0 * sp2,
                  hl[6 * j + 3] * sp1 * sp1, hl[6 * j + 4] * sp1 * sp2, hl[6 * j + 5] * sp2 * sp2};
  const double d0 = fmin(fmax(hs[0], 1e-6), 1e32) / st.radius;
  const double d1 = fmin(fmax(hs[3], 1e-6), 1e32) / st.radius;
  const double d2 = fmin(fmax(hs[5], 1e-6), 1e32) / st.radius;
  hs[0] += d0, hs[3] += d1, hs[5] += d2;
  double hi[6];
  if (!inv3_sym(hs, hi)) {
    hi[0] = hi[3] = hi[5] = 0.0 / 0.0;  // poisons the step => invalid step handling
    hi[1] = hi[2] = hi[4] = 0;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) B.hinv[6 * j + i] = hi[i];
  B.dl[3 * j] = d0, B.dl[3 * j + 1] = d1, B.dl[3 * j + 2] = d2;
#pragma unroll
  for (int k = 0; k < 3; k++) B.gl2[3 * j + k] = B.glsc[st.cur][3 * j + k];
}

__global__ __launch_bounds__(kCamChunk) void k_ba_cams_large(BaDev B) {
  __shared__ double lds27[4 * 27];
  const BaState st = *B.st;
  if (st.done) return;
  // the camera blocks depend on the linearisation only, not on the radius: after a rejected (or invalid) step the point of
  // linearisation has not moved and this shard's slabs of the previous iteration stand (39 us per iteration at config 4)
  if (!st.first && !st.last_ok) return;
  ba_cams_role<false>(B, st, blockIdx.x / B.n_cchunks, blockIdx.x % B.n_cchunks, lds27);
}

typedef double double2_t __attribute__((ext_vector_type(2)));

// 36 per-lane partial sums -> their 64-lane totals, one per lane: a reduce-scatter butterfly (each level halves the
// number of values a lane carries: 38 exchanges instead of the 216 of 36 full butterflies).  Lane l ends up with the
// total of entry bitreverse6(l) when that is < 36.  Fixed order.
__device__ __forceinline__ double reduce_scatter36(const double (&v0)[36], int lane, int &index) {
  auto level = [&](auto nin, const double *in, double *out, int bit, int width) {
    constexpr int N = decltype(nin)::value;
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; i++) {
      const double lo = in[2 * i], hi = 2 * i + 1 < N ? in[2 * i + 1] : 0.0;
      const double keep = bit ? hi : lo, send = bit ? lo : hi;
      out[i] = keep + __shfl_xor(send, width);
    }
  };
  double v1[18], v2[9], v3[5], v4[3], v5[2], v6[1];
  level(std::integral_constant<int, 36>{}, v0, v1, lane & 32, 32);
  level(std::integral_constant<int, 18>{}, v1, v2, lane & 16, 16);
  level(std::integral_constant<int, 9>{}, v2, v3, lane & 8, 8);
  level(std::integral_constant<int, 5>{}, v3, v4, lane & 4, 4);
  level(std::integral_constant<int, 3>{}, v4, v5, lane & 2, 2);
  level(std::integral_constant<int, 2>{}, v5, v6, lane & 1, 1);
  index = (int)(__brev((unsigned)lane) >> 26);
  return v6[0];
}

__global__ __launch_bounds__(256) void k_ba_pairs(BaDev B) {
  const BaState st = *B.st;
  if (st.done) return;
  const int lane = threadIdx.x & 63;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: behind the diagonal pairs every XCD takes one
  // contiguous eighth of the list, so the W blocks of the cameras it is working on are fetched into one L2, not eight.
  int blk = blockIdx.x;
  if (blk >= B.pair_diag_blocks) {
    const int q = blk - B.pair_diag_blocks, per_xcd = ((int)gridDim.x - B.pair_diag_blocks) >> 3;
    blk = B.pair_diag_blocks + (q & 7) * per_xcd + (q >> 3);
  }
  const int pr = blk * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int c = B.pair_cc[2 * pr], cp = B.pair_cc[2 * pr + 1];
  const int t0 = B.pair_start[2 * pr], t1 = B.pair_start[2 * pr + 1];
  if (t1 <= t0) return;  // padding
  const double *We = B.We[st.cur], *gls = B.glsc[st.cur];
  double acc[36], rh[6];
#pragma unroll
  for (int i = 0; i < 36; i++) acc[i] = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) rh[i] = 0;
  // The walk is a chain of dependent loads (couple -> flags, point block, two W blocks); the kernel is bound by that
  // latency, not by arithmetic or bandwidth.  The couple carries its point index (one load instead of two dependent
  // ones), the next couple is requested before the current one is used, and every load is unconditional (clamped
  // index; an inactive or out-of-range couple gets a zero point block, so it adds exact zeros).
  int4 rec = B.pair_e[min(t0 + lane, t1 - 1)];
  for (int t = t0 + lane; t - lane < t1; t += 64) {
    const int4 nxt = B.pair_e[min(t + 64, t1 - 1)];
    const int e = rec.x, ep = rec.y, j = rec.z;
    // (no activity test: the linearisation writes all-zero W rows for a deactivated edge -- point_linearize --, so its
    // couples add exact zeros; the two scattered byte loads it took were two of the ~7 cache lines a couple touches)
    const bool ok = t < t1;
    // 16-byte loads: the texture-address unit spends ~1.5 cycles per lane and instruction on these scattered blocks
    // and was busy 91 % of the kernel with 8-byte loads (TA_TA_BUSY); the blocks are 48 and 144 bytes, 16-byte aligned
    const double2_t *hv2 = reinterpret_cast<const double2_t *>(B.hinv + 6 * j);
    const double2_t ha = hv2[0], hb = hv2[1], hc = hv2[2];
    const double h00 = ok ? ha.x : 0.0, h01 = ok ? ha.y : 0.0, h02 = ok ? hb.x : 0.0, h11 = ok ? hb.y : 0.0,
                 h12 = ok ? hc.x : 0.0, h22 = ok ? hc.y : 0.0;
    double w[18], wp[18];
    {
      const double2_t *w2 = reinterpret_cast<const double2_t *>(We + 18LL * e);
      const double2_t *wp2 = reinterpret_cast<const double2_t *>(We + 18LL * ep);
#pragma unroll
      for (int q = 0; q < 9; q++) {
        const double2_t a2 = w2[q], b2 = wp2[q];
        w[2 * q] = a2.x, w[2 * q + 1] = a2.y, wp[2 * q] = b2.x, wp[2 * q + 1] = b2.y;
      }
    }
    double Y[6][3];  // (W_e Hinv)[a][k'] with W_e[a][k] = w[6 k + a]
#pragma unroll
    for (int a = 0; a < 6; a++) {
      const double w0 = w[a], w1 = w[6 + a], w2 = w[12 + a];
      Y[a][0] = w0 * h00 + w1 * h01 + w2 * h02;
      Y[a][1] = w0 * h01 + w1 * h11 + w2 * h12;
      Y[a][2] = w0 * h02 + w1 * h12 + w2 * h22;
    }
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) acc[6 * a + b] += Y[a][0] * wp[b] + Y[a][1] * wp[6 + b] + Y[a][2] * wp[12 + b];
    if (c == cp) {  // diagonal pair: the camera's own edges -> its part of the right-hand side
      const double g0 = gls[3 * j], g1 = gls[3 * j + 1], g2 = gls[3 * j + 2];
#pragma unroll
      for (int a = 0; a < 6; a++) rh[a] += Y[a][0] * g0 + Y[a][1] * g1 + Y[a][2] * g2;
    }
    rec = nxt;
  }
  // block (row c', column c) of the lower triangle: entry (6c'+b, 6c+a) = -G[a][b]
  int idx;
  const double mine = reduce_scatter36(acc, lane, idx);
  if (idx < 36) {
    const int a = idx / 6, b = idx - 6 * a;
    B.Sd[(long long)(6 * cp + b) * B.ld + 6 * c + a] = -mine;
  }
  if (c == cp) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
      for (int i = 0; i < 6; i++) rh[i] += __shfl_xor(rh[i], o);
    }
    double r1 = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) r1 = (lane == i) ? rh[i] : r1;
    if (lane < 6) B.Sd[(long long)B.ld * B.ld + 6 * c + lane] = -r1;  // row ld: -(sum_e Y_e gl''_j); completed by k_ba_assemble_large
  }
}

// k_ba_pairs_lds -- the same sums with the blocks staged through LDS (round 6).  k_ba_pairs' lane = couple mapping makes every one
// of its 21 loads per couple touch 64 different cache lines (9 + 9 + 3 instructions of 16 bytes per lane, each lane in another
// block), holds the 42 loaded doubles of the couple in flight in registers (208 registers, two wavefronts per SIMD) and walks a
// pair as a chain of dependent round trips.  Here a wavefront takes 32 couples per step and
//  * fetches their blocks COOPERATIVELY by global_load_dwordx4 ... lds: nine consecutive lanes read one 144-byte W block (seven
//    blocks per instruction, ~14 lines instead of 64), three lanes one 48-byte point block; the data never pass through registers,
//    the block indices reach the fetching lanes by ds_bpermute from the lanes that hold the couple records;
//  * computes with lane = (couple, half): lanes l and l + 32 share a couple and produce rows 0..2 / 3..5 of its 6 x 6 product from
//    LDS reads (stride 144 / 48 bytes: conflict-free for 16-byte reads), 18 accumulators each;
//  * ends with a reduce-scatter butterfly over the 32 couples' lanes (20 exchanges; entry bitreverse5(l & 31) of the half).
// ~100 registers: three workgroups per CU by LDS (4 x 10.5 KB each).  Deterministic; the order of the sum differs from
// k_ba_pairs', so the two agree to rounding (1e-16 relative), not bitwise.
constexpr int kPrStep = 32;                       // couples per step
constexpr int kPrLds = kPrStep * (144 + 144 + 48);  // 10752 bytes per wavefront

__device__ __forceinline__ double reduce_scatter18(const double (&v0)[18], int r, int &index) {
  auto level = [&](auto nin, const double *in, double *out, int bit, int width) {
    constexpr int N = decltype(nin)::value;
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; i++) {
      const double lo = in[2 * i], hi = 2 * i + 1 < N ? in[2 * i + 1] : 0.0;
      const double keep = bit ? hi : lo, send = bit ? lo : hi;
      out[i] = keep + __shfl_xor(send, width);
    }
  };
  double v1[9], v2[5], v3[3], v4[2], v5[1];
  level(std::integral_constant<int, 18>{}, v0, v1, r & 16, 16);
  level(std::integral_constant<int, 9>{}, v1, v2, r & 8, 8);
  level(std::integral_constant<int, 5>{}, v2, v3, r & 4, 4);
  level(std::integral_constant<int, 3>{}, v3, v4, r & 2, 2);
  level(std::integral_constant<int, 2>{}, v4, v5, r & 1, 1);
  index = (int)(__brev((unsigned)r) >> 27);
  return v5[0];
}

__global__ __launch_bounds__(256) void k_ba_pairs_lds(BaDev B) {
  __shared__ __attribute__((aligned(16))) unsigned char pr_lds[4][kPrLds];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef const __attribute__((address_space(1))) unsigned char gmem_u8;
  const BaState st = *B.st;
  if (st.done) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  int blk = blockIdx.x;  // (XCD split as in k_ba_pairs)
  if (blk >= B.pair_diag_blocks) {
    const int q = blk - B.pair_diag_blocks, per_xcd = ((int)gridDim.x - B.pair_diag_blocks) >> 3;
    blk = B.pair_diag_blocks + (q & 7) * per_xcd + (q >> 3);
  }
  const int pr = blk * 4 + wave;
  const int c = B.pair_cc[2 * pr], cp = B.pair_cc[2 * pr + 1];
  const int t0 = B.pair_start[2 * pr], t1 = B.pair_start[2 * pr + 1];
  if (t1 <= t0) return;  // padding (the kernel has no workgroup barrier)
  gmem_u8 *We = (gmem_u8 *)B.We[st.cur], *Hi = (gmem_u8 *)B.hinv;
  const double *gls = B.glsc[st.cur];
  lds_u8 *const LW = (lds_u8 *)pr_lds[wave], *const LP = LW + kPrStep * 144, *const LH = LP + kPrStep * 144;
  const int r = lane & 31, half = lane >> 5;
  // roles in the fetch: lane = 9 block + chunk for the W blocks (7 blocks per instruction), 3 block + chunk for the point blocks (21)
  const int c7 = lane / 9, ch9 = (lane - 9 * c7) * 16, c21 = lane / 3, ch3 = (lane - 3 * c21) * 16;
  double acc[18], rh[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 18; i++) acc[i] = 0;
  const bool diag = c == cp;
  int4 rec = B.pair_e[min(t0 + r, t1 - 1)];  // (couple of lanes r and r + 32; past the end: the last one, its sums are zeroed)
  for (int t = t0; t < t1; t += kPrStep) {
    const int4 nxt = B.pair_e[min(t + kPrStep + r, t1 - 1)];
    double g0 = 0, g1 = 0, g2 = 0;
    if (diag) g0 = gls[3 * rec.z], g1 = gls[3 * rec.z + 1], g2 = gls[3 * rec.z + 2];
#pragma unroll
    for (int q = 0; q < 5; q++) {
      const int ci = min(7 * q + c7, kPrStep - 1);
      const unsigned e = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ci, rec.x), ep = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ci, rec.y);
      if (c7 < 7 && 7 * q + c7 < kPrStep) {
        __builtin_amdgcn_global_load_lds(We + ((unsigned long long)e * 144u + (unsigned)ch9), LW + 1008 * q, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(We + ((unsigned long long)ep * 144u + (unsigned)ch9), LP + 1008 * q, 16, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int ci = min(21 * q + c21, kPrStep - 1);
      const unsigned j = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ci, rec.z);
      if (c21 < 21 && 21 * q + c21 < kPrStep) __builtin_amdgcn_global_load_lds(Hi + ((unsigned long long)j * 48u + (unsigned)ch3), LH + 1008 * q, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool ok = t + r < t1;
    typedef __attribute__((address_space(3))) double2_t lds_d2;
    typedef __attribute__((address_space(3))) double lds_d;
    const lds_d2 *hv = (const lds_d2 *)(LH + 48 * r);
    const double2_t ha = hv[0], hb = hv[1], hc = hv[2];
    const double h00 = ok ? ha.x : 0.0, h01 = ok ? ha.y : 0.0, h02 = ok ? hb.x : 0.0, h11 = ok ? hb.y : 0.0, h12 = ok ? hc.x : 0.0,
                 h22 = ok ? hc.y : 0.0;
    const lds_d *wl = (const lds_d *)(LW + 144 * r) + 3 * half;  // W_e[a][k] = w[6 k + a], rows a = 3 half + i
    double Y[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double w0 = wl[i], w1 = wl[6 + i], w2 = wl[12 + i];
      Y[i][0] = w0 * h00 + w1 * h01 + w2 * h02;
      Y[i][1] = w0 * h01 + w1 * h11 + w2 * h12;
      Y[i][2] = w0 * h02 + w1 * h12 + w2 * h22;
    }
    const lds_d2 *wp2 = (const lds_d2 *)(LP + 144 * r);
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int bb = 0; bb < 3; bb++) {
        const double2_t p = wp2[3 * k + bb];  // W_e'[2 bb][k], W_e'[2 bb + 1][k]
#pragma unroll
        for (int i = 0; i < 3; i++) {
          acc[6 * i + 2 * bb] += Y[i][k] * p.x;
          acc[6 * i + 2 * bb + 1] += Y[i][k] * p.y;
        }
      }
    }
    if (diag) {  // the camera's own edges -> its part of the right-hand side
#pragma unroll
      for (int i = 0; i < 3; i++) rh[i] += Y[i][0] * g0 + Y[i][1] * g1 + Y[i][2] * g2;
    }
    rec = nxt;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next step's blocks overwrite these)
    __builtin_amdgcn_wave_barrier();
  }
  // block (row c', column c) of the lower triangle: entry (6c'+b, 6c+a) = -G[a][b]
  int idx;
  const double mine = reduce_scatter18(acc, r, idx);
  if (idx < 18) {
    const int i = idx / 6, b = idx - 6 * i, a = 3 * half + i;
    B.Sd[(long long)(6 * cp + b) * B.ld + 6 * c + a] = -mine;
  }
  if (diag) {
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
#pragma unroll
      for (int i = 0; i < 3; i++) rh[i] += __shfl_xor(rh[i], o);
    }
    double r1 = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) r1 = (r == i) ? rh[i] : r1;
    if (r < 3) B.Sd[(long long)B.ld * B.ld + 6 * c + 3 * half + r] = -r1;  // row ld: -(sum_e Y_e gl''_j); completed by k_ba_assemble_large
  }
}

__global__ __launch_bounds__(256) void k_ba_partials_large(BaDev B) {
  __shared__ double red[8];
  const BaState st0 = *B.st;
  if (st0.done) return;
  const int tid = threadIdx.x;
  double *ext = B.Sd + large_ext_off(B.ld);
  // camera-block sums: one entry per thread over the whole grid; the cost / gradient part on workgroup 0
  for (int i = blockIdx.x * 256 + tid; i < B.nf * 27; i += gridDim.x * 256) {
    const int slot = i / 27, t = i - slot * 27;
    const double *sp = B.slab_cam + (long long)slot * B.n_cchunks * 27 + t;
    double a = 0;
    for (int cchunk = 0; cchunk < B.n_cchunks; cchunk++) a += sp[cchunk * 27];  // chunk order
    ext[i] = a;
  }
  if (blockIdx.x != 0) return;
  double cs = 0, m = 0;
  const double *spt = B.slab_pt[st0.cur];
  for (int b = tid; b < B.n_pblocks; b += 256) cs += spt[2 * b], m = fmax(m, spt[2 * b + 1]);
  double v2[2] = {cs, 0};
  block_sum<2>(v2, red);
  m = wave_max(m);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    ext[B.nf * 27] = v2[0];
    const double gm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    for (int k = 0; k < B.n_shards; k++) ext[B.nf * 27 + 1 + k] = (k == B.shard) ? gm : 0.0;
  }
}


// Sharded large systems: what a shard contributes to the reduced system is the 64 x 64 tiles of the matrix that hold a
// covisible pair (the plan's input pattern: ~260 of 1128 at config 4), the right-hand-side row and the camera-block
// extras.  The collective carries exactly that, packed (dir 0: storage -> payload, 1: payload -> storage), instead of the
// whole Cholesky storage (72 MB -> 8.6 MB at 500 key-frames).
__global__ __launch_bounds__(256) void k_ba_pack_large(BaDev B, const int2 *tiles, int n_tiles, double *payload, int dir) {
  const BaState st = *B.st;
  if (st.done) return;
  const int tid = threadIdx.x, blk = blockIdx.x;
  constexpr int T = vo::kCholPanel;
  if (blk < n_tiles) {
    const int2 t = tiles[blk];
    double *pk = payload + (long long)blk * T * T;
    for (int i = tid; i < T * T; i += 256) {
      double *a = B.Sd + (long long)(T * t.x + i / T) * B.ld + T * t.y + (i % T);
      if (dir == 0) pk[i] = *a;
      else *a = pk[i];
    }
    return;
  }
  double *pk = payload + (long long)n_tiles * T * T;
  const int n_ext = B.nf * 27 + 1 + B.n_shards;
  double *rhs = B.Sd + (long long)B.ld * B.ld, *ext = B.Sd + large_ext_off(B.ld);
  for (int i = (blk - n_tiles) * 256 + tid; i < B.ld + n_ext; i += (gridDim.x - n_tiles) * 256) {
    double *a = i < B.ld ? rhs + i : ext + (i - B.ld);
    if (dir == 0) pk[i] = *a;
    else *a = pk[i];
  }
}

// Segment solve, the two collectives around the separator phase (dir 0: storage -> payload, 1: payload -> storage).
//  what 1: the separator block -- the tiles of the factor with both indices behind the segments (Schur fill included) and
//          the right-hand side from seg_row0 on;
//  what 2: the solution row: a rank contributes the unknowns of its own segment columns (the lead rank also the
//          separators'), zeros elsewhere, so that the sum is the whole step; one more slot carries the ranks' fail flags
//          (a rank whose segment was not positive definite must fail the step on every rank).
__global__ __launch_bounds__(256) void k_ba_pack_seg(BaDev B, const int2 *tiles, int n_tiles, double *payload, int dir, int what) {
  const BaState st = *B.st;
  if (st.done) return;
  const int tid = threadIdx.x, blk = blockIdx.x;
  constexpr int T = vo::kCholPanel;
  if (what == 1) {
    if (blk < n_tiles) {
      const int2 t = tiles[blk];
      double *pk = payload + (long long)blk * T * T;
      for (int i = tid; i < T * T; i += 256) {
        double *a = B.Sd + (long long)(T * t.x + i / T) * B.ld + T * t.y + (i % T);
        if (dir == 0) pk[i] = *a;
        else *a = pk[i];
      }
      return;
    }
    double *pk = payload + (long long)n_tiles * T * T, *rhs = B.Sd + (long long)B.ld * B.ld + B.seg_row0;
    for (int i = (blk - n_tiles) * 256 + tid; i < B.ld - B.seg_row0; i += (gridDim.x - n_tiles) * 256) {
      if (dir == 0) pk[i] = rhs[i];
      else rhs[i] = pk[i];
    }
    return;
  }
  double *x = B.Sd + (long long)(B.ld + 1) * B.ld;
  for (int i = blk * 256 + tid; i <= B.ld; i += gridDim.x * 256) {
    if (i == B.ld) {
      if (dir == 0) payload[i] = *B.chol_fail != 0 ? 1.0 : 0.0;
      else if (payload[i] != 0.0) *B.chol_fail = 1;
      continue;
    }
    const int tj = i / T;
    const bool mine = i < B.seg_row0 ? ((B.seg_own >> tj) & 1ull) != 0 : B.seg_lead != 0;
    if (dir == 0) payload[i] = mine ? x[i] : 0.0;
    else x[i] = payload[i];
  }
}

// one workgroup: Jacobi scale, LM diagonal, cost / gradient-max bookkeeping from the (all-reduced) extras
__global__ __launch_bounds__(256) void k_ba_prestep_large(BaDev B) {
  __shared__ double red[4];
  BaState *S = B.st;
  const BaState st0 = *S;
  if (st0.done) return;
  const int tid = threadIdx.x, n = 6 * B.nf;
  const double *hp = B.Sd + large_ext_off(B.ld);
  if (tid == 0) *B.chol_fail = 0;
  double gm = 0;
  for (int i = tid; i < n; i += 256) {
    const int slot = i / 6, a = i - slot * 6;
    int t = 0;
    for (int q = 0; q < a; q++) t += 6 - q;
    const double hd = hp[slot * 27 + t];
    double sc;
    if (st0.first) {
      sc = 1.0 / (1.0 + sqrt(hd));
      B.scale_c[i] = sc;
    } else {
      sc = B.scale_c[i];
    }
    B.sc_v[i] = sc;
    B.Dd_v[i] = fmin(fmax(hd * sc * sc, 1e-6), 1e32) / st0.radius;
    const double gp = hp[slot * 27 + 21 + a];
    gm = fmax(gm, fabs(gp));
    B.gpp_v[i] = sc * gp;
  }
  gm = wave_max(gm);
  if ((tid & 63) == 0) red[tid >> 6] = gm;
  __syncthreads();
  if (tid == 0) {
    double gmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    for (int k = 0; k < B.n_shards; k++) gmax = fmax(gmax, hp[B.nf * 27 + 1 + k]);
    const double cost = hp[B.nf * 27];
    S->gmax = gmax;
    S->x_cost = cost;
    if (st0.first) S->initial_cost = cost;
    if (st0.last_ok && gmax <= 1e-10) {
      S->termination = 3;
      S->done = 1;
    } else {
      S->iter = st0.iter + 1;
    }
    S->first = 0;
  }
}

// Start of a linearisation: zeros in the tiles of the plan (what k_ba_pairs does not write must read as zero: fill
// tiles, the parts of a tile without a covisible pair) and in the right-hand-side / solution rows.  Tiles outside the
// plan are never read or written by anything: 15.5 MB instead of the 74 MB memset of the whole storage at config 4.
__global__ __launch_bounds__(256) void k_ba_zero_large(BaDev B) {
  if (B.st->done) return;
  constexpr int T = vo::kCholPanel;
  const int tid = threadIdx.x, blk = blockIdx.x;
  if (blk < B.n_ltiles) {
    int2 t = B.ltiles[blk];
    t.y &= 0xffff;
    for (int i = tid; i < T * T / 2; i += 256) {
      const int r = i / (T / 2), c2 = i - r * (T / 2);
      *reinterpret_cast<double2 *>(B.Sd + (long long)(T * t.x + r) * B.ld + T * t.y + 2 * c2) = make_double2(0.0, 0.0);
    }
    return;
  }
  for (int i = (blk - B.n_ltiles) * 256 + tid; i < 2 * B.ld; i += (gridDim.x - B.n_ltiles) * 256) B.Sd[(long long)B.ld * B.ld + i] = 0.0;
}

// A = sc_r (Hpp - G) sc_c + D on the lower triangle of the plan's tiles (identity on the padding), rhs'' = g'' - sc (Y gl'')
__global__ __launch_bounds__(256) void k_ba_assemble_large(BaDev B) {
  if (B.st->done) return;
  constexpr int T = vo::kCholPanel;
  const double *hp = B.Sd + large_ext_off(B.ld);
  const int n = 6 * B.nf, tid = threadIdx.x, blk = blockIdx.x;
  if (blk >= B.n_ltiles) {  // right-hand side, row ld: in place on this shard-summed -(Y gl'')
    for (int r = (blk - B.n_ltiles) * 256 + tid; r < B.ld; r += (gridDim.x - B.n_ltiles) * 256) {
      double *rh = B.Sd + (long long)B.ld * B.ld + r;
      const bool whole = !B.seg_mode || r < B.seg_row0 || B.seg_lead;  // (separator rows of a segment solve: one rank adds g'')
      *rh = r < n ? (whole ? B.gpp_v[r] : 0.0) + B.sc_v[r] * *rh : 0.0;
    }
    return;
  }
  int2 t = B.ltiles[blk];
  const bool partial = (t.y >> 16) != 0;  // segment solve: a separator tile on a rank that does not add the replicated terms
  t.y &= 0xffff;
  for (int i = tid; i < T * T; i += 256) {
    const int r = T * t.x + i / T, c = T * t.y + (i % T);
    if (c > r) continue;
    const long long idx = (long long)r * B.ld + c;
    double v;
    if (r < n) {
      v = B.Sd[idx];
      if (r / 6 == c / 6 && !partial) {
        const int slot = r / 6, a = c % 6, b = r % 6;  // a <= b
        int q0 = 0;
        for (int q = 0; q < a; q++) q0 += 6 - q;
        v += hp[slot * 27 + q0 + (b - a)];
      }
      v *= B.sc_v[r] * B.sc_v[c];
      if (r == c && !partial) v += B.Dd_v[r];
    } else {
      v = (r == c && !partial) ? 1.0 : 0.0;
    }
    B.Sd[idx] = v;
  }
}

// one workgroup: step, candidate poses and the camera part of the trust-region quantities
__global__ __launch_bounds__(256) void k_ba_poststep_large(BaDev B) {
  __shared__ double red[4 * 5];
  BaState *S = B.st;
  const BaState st0 = *S;
  if (st0.done) return;
  const int tid = threadIdx.x, n = 6 * B.nf;
  const double *y = B.Sd + (long long)(B.ld + 1) * B.ld;  // solution row of the Cholesky storage
  int bad = *B.chol_fail;
  double gdot = 0, dquad = 0;
  for (int i = tid; i < n; i += 256) {
    const double stp = -y[i];
    if (!isfinite(stp)) bad = 1;
    gdot += B.gpp_v[i] * stp;
    dquad += B.Dd_v[i] * stp * stp;
    B.zc[i] = B.sc_v[i] * y[i];
  }
  const int failed = __syncthreads_or(bad);
  const int cur0 = st0.cur;
  const double *X = B.Xc[cur0];
  double *Xn = B.Xc[cur0 ^ 1];
  double xn2 = 0, cn2 = 0, sn2 = 0;
  for (int c = tid; c < B.n_cams; c += 256) {
    const int slot = B.cam_slot[c];
    double x0[6], xc[6];
#pragma unroll
    for (int a = 0; a < 6; a++) x0[a] = X[6 * c + a];
    if (slot >= 0 && !failed) {
      double d[6];
#pragma unroll
      for (int a = 0; a < 6; a++) d[a] = -y[6 * slot + a] * B.sc_v[6 * slot + a];
      se3_plus(x0, d, xc);
      store_pc(B.PC[cur0 ^ 1], c, pose_cache(xc));
    } else {
#pragma unroll
      for (int a = 0; a < 6; a++) xc[a] = x0[a];
      store_pc(B.PC[cur0 ^ 1], c, load_pc(B.PC[cur0], c));
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
      Xn[6 * c + a] = xc[a];
      if (slot >= 0 && B.cam_in[c] == B.epoch) {
        xn2 += x0[a] * x0[a];
        cn2 += xc[a] * xc[a];
        sn2 += (xc[a] - x0[a]) * (xc[a] - x0[a]);
      }
    }
  }
  double v5[5] = {gdot, dquad, xn2, cn2, sn2};
  block_sum<5>(v5, red);
  if (tid == 0) {
    S->gdot_c = v5[0];
    S->dquad_c = v5[1];
    S->x_norm2_c = v5[2];
    S->cand_norm2_c = v5[3];
    S->step_norm2_c = v5[4];
    S->solve_failed = failed;
  }
}

// TrustRegionMinimizer step evaluation + LevenbergMarquardtStrategy radius update (one thread)
// Works on a register copy of the state (`s`, as read at the start of the launch: nothing else writes it
// in between) and the six reduced sums `p`; the caller stores the result back in one piece.
__device__ __forceinline__ void ba_update_logic(BaState &s, const double p[6]) {
  BaState *S = &s;
  const double cand_cost = p[0];
  const double model = -0.5 * (S->gdot_c + p[1]) + 0.5 * (S->dquad_c + p[2]);
  const double step_norm = sqrt(S->step_norm2_c + p[3]);
  const double x_norm = sqrt(S->x_norm2_c + p[4]);
  S->cand_cost = cand_cost;
  S->last_ok = 0;
  if (S->solve_failed || !(model > 0.0) || !isfinite(model)) {
    if (++S->invalid >= 5) {
      S->termination = 4;
      S->done = 1;
      return;
    }
    S->radius /= S->decrease;
    S->decrease *= 2.0;
  } else {
    S->invalid = 0;
    const double cc = isfinite(cand_cost) ? cand_cost : 1.7976931348623157e308;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) {
      S->termination = 2;
      S->done = 1;
      return;
    }
    const double change = S->x_cost - cc;
    if (fabs(change) <= 1e-6 * S->x_cost) {
      S->termination = 1;
      S->done = 1;
      return;
    }
    const double rel = change / model;
    if (rel > 1e-3) {
      S->cur ^= 1;
      S->x_cost = cc;
      const double t2 = 2.0 * rel - 1.0;
      S->radius = fmin(S->radius / fmax(1.0 / 3.0, 1.0 - t2 * t2 * t2), 1e16);
      S->decrease = 2.0;
      S->accepted += 1;
      S->last_ok = 1;
    } else {
      S->radius /= S->decrease;
      S->decrease *= 2.0;
    }
  }
  if (S->iter >= S->max_it) {
    S->termination = 0;
    S->done = 1;
  } else if (S->radius < 1e-32) {
    S->termination = 4;
    S->done = 1;
  }
}

// --------------------------------------------------------------------------------------------
// k_ba_backsub: SchurEliminator::BackSubstitute, candidate points and the candidate cost
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ba_backsub(BaDev B) {
  __shared__ double lds[4 * 6];
  __shared__ double lmx[4];
  __shared__ double pcL[kPcLds * 12];  // pose caches of the candidate poses
  __shared__ double zcL[kMaxN];        // camera step
  const BaState st = *B.st;
  if (st.done) return;
  STAMP0(36);
  const int tid = threadIdx.x, g = tid & (kGroup - 1);
  // every edge evaluation reads a 12-double pose cache: from LDS it costs one level of the load
  // chain less and no global traffic (falls back to HBM for more cameras than the table holds)
  const bool pc_lds = B.n_cams <= kPcLds;
  if (pc_lds)
    for (int i = tid; i < 12 * B.n_cams; i += 256) pcL[i] = B.PC[st.cur ^ 1][i];
  const bool zc_lds = 6 * B.nf <= kMaxN;
  if (zc_lds)
    for (int i = tid; i < 6 * B.nf; i += 256) zcL[i] = B.zc[i];
  __syncthreads();
  const double *PCcand = pc_lds ? pcL : B.PC[st.cur ^ 1];
  const double *zcv = zc_lds ? zcL : B.zc;
  const int li = blockIdx.x * kPtsPerBlock + (tid >> kGroupLog);
  const bool valid = li < B.n_local;
  const int j = valid ? B.local_pts[li] : 0;
  const double *Xp = B.Xp[st.cur];
  double *Xpn = B.Xp[st.cur ^ 1];
  double v[6] = {0, 0, 0, 0, 0, 0};  // cand_cost, gdot_l, dquad_l, step2, xnorm2, candnorm2
  double ccost = 0, cgmax = 0;
  if (valid) {
    const int e0 = B.pt_start[j], e1 = B.pt_start[j + 1];
    double rr[3] = {0, 0, 0};
    for (int e = e0 + g; e < e1; e += kGroup) {
      if (!B.e_active[e]) continue;
      const int slot = B.cam_slot[B.e_cam[e]];
      if (slot < 0) continue;
      const double *w0 = w_row(B, st.cur, e, j, slot, 0), *w1 = w_row(B, st.cur, e, j, slot, 1),
                   *w2 = w_row(B, st.cur, e, j, slot, 2);
#pragma unroll
      for (int a = 0; a < 6; a++) {
        const double z = zcv[6 * slot + a];
        rr[0] += w0[a] * z;
        rr[1] += w1[a] * z;
        rr[2] += w2[a] * z;
      }
    }
#pragma unroll
    for (int o = 1; o < kGroup; o <<= 1)
#pragma unroll
      for (int i = 0; i < 3; i++) rr[i] += __shfl_xor(rr[i], o);
    const double *hi = B.hinv + 6 * j;
    const double g2[3] = {B.gl2[3 * j], B.gl2[3 * j + 1], B.gl2[3 * j + 2]};
    const double q[3] = {g2[0] - rr[0], g2[1] - rr[1], g2[2] - rr[2]};
    const double y0 = hi[0] * q[0] + hi[1] * q[1] + hi[2] * q[2];
    const double y1 = hi[1] * q[0] + hi[3] * q[1] + hi[4] * q[2];
    const double y2 = hi[2] * q[0] + hi[4] * q[1] + hi[5] * q[2];
    const double stp[3] = {-y0, -y1, -y2};
    const bool ptin = B.pt_in[j] == B.epoch;
    const bool in = ptin && !st.solve_failed;
    double pn[3];
    for (int k = 0; k < 3; k++) pn[k] = Xp[3 * j + k] + (in ? stp[k] * B.scale_p[3 * j + k] : 0.0);
    if (g == 0) {
      for (int k = 0; k < 3; k++) Xpn[3 * j + k] = pn[k];
      if (ptin) {
        for (int k = 0; k < 3; k++) {
          v[1] += g2[k] * stp[k];
          v[2] += B.dl[3 * j + k] * stp[k] * stp[k];
          v[3] += (pn[k] - Xp[3 * j + k]) * (pn[k] - Xp[3 * j + k]);
          v[4] += Xp[3 * j + k] * Xp[3 * j + k];
          v[5] += pn[k] * pn[k];
        }
      }
    }
    // candidate cost and, in the same sweep, the complete (radius-independent) linearisation at the
    // candidate into the other buffer set: if the step is accepted the next iteration starts from it
#ifdef VO_BA_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP0(37);
    point_linearize(B, st, j, g, pn, PCcand, st.cur ^ 1, ccost, cgmax);
    v[0] = ccost;
  }
#ifdef VO_BA_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  STAMP0(38);
  // one pass for everything the block hands on: the six step sums, and (v[0] doubles as it) the
  // candidate cost and gradient max-norm of the candidate linearisation
  {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = wave_sum(v[i]);
    const double gm = wave_max(cgmax);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) lds[wave * 6 + i] = v[i];
      lmx[wave] = gm;
    }
    __syncthreads();
    if (tid < 6) {
      const double t = ((lds[tid] + lds[6 + tid]) + lds[12 + tid]) + lds[18 + tid];
      if (B.fused)
        st_sc1(&B.slab_bs[6 * blockIdx.x + tid], t);
      else
        B.slab_bs[6 * blockIdx.x + tid] = t;
      if (tid == 0) {
        B.slab_pt[st.cur ^ 1][2 * blockIdx.x] = t;
        B.slab_pt[st.cur ^ 1][2 * blockIdx.x + 1] = fmax(fmax(lmx[0], lmx[1]), fmax(lmx[2], lmx[3]));
      }
    }
  }
  STAMP0(39);
  if (!B.fused) return;
  // single shard: the last block to arrive sums the slabs (fixed order) and runs the update
  __shared__ int s_last;
  if (!arrive_and_check_last(&B.counters[0], gridDim.x, &s_last)) return;
#ifdef VO_BA_STAMPS
  if (tid == 0) B.dbg[40] = __builtin_amdgcn_s_memrealtime();
#endif
  if (tid < 64) {
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int b0 = 0; b0 < B.n_pblocks; b0 += 256) {  // 24 coherent loads in flight per lane, added in block order
      double w[4][6];
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < 6; i++) w[q][i] = ld_sc1(&B.slab_bs[6 * min(b0 + 64 * q + tid, B.n_pblocks - 1) + i]);
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] += b0 + 64 * q + tid < B.n_pblocks ? w[q][i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) a[i] = wave_sum(a[i]);
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) B.payload2[i] = a[i];
      BaState ns = st;  // the solve kernel's fields (gdot_c, ...) are in the launch-time copy
      ba_update_logic(ns, a);
      *B.st = ns;
#ifdef VO_BA_STAMPS
      B.dbg[41] = __builtin_amdgcn_s_memrealtime();
#endif
    }
  }
}

__global__ __launch_bounds__(64) void k_ba_reduce2(BaDev B) {
  const BaState st = *B.st;
  if (st.done) return;
  const int t = threadIdx.x;
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int b = t; b < B.n_pblocks; b += 64)
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] += B.slab_bs[6 * b + i];
#pragma unroll
  for (int i = 0; i < 6; i++) v[i] = wave_sum(v[i]);
  if (t < 6) B.payload2[t] = v[t];
}

// --------------------------------------------------------------------------------------------
// k_ba_update: TrustRegionMinimizer step evaluation + LevenbergMarquardtStrategy radius update
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ba_update(BaDev B) {
  if (B.st->done || threadIdx.x != 0) return;
  BaState ns = *B.st;
  double p[6];
  for (int i = 0; i < 6; i++) p[i] = B.payload2[i];
  ba_update_logic(ns, p);
  *B.st = ns;
}

// flags reset (epoch wrap-around, and the outlier mask of a local BA stopped before problem 2)
__global__ void k_ba_clear(BaDev B, int set_active, uint8_t *out_or_null) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B.n_pts) B.pt_in[i] = 0;
  if (i < B.n_cams) B.cam_in[i] = 0;
  if (i < B.n_edges) {
    if (set_active) B.e_active[i] = 1;
    if (out_or_null) out_or_null[i] = 0;
  }
}

// float chi2 classification of one edge, optimizer_ceres.cpp:618-689 (mode 0: writes
// e_active = inlier and out = outlier) and :703-755 (mode 1: out |= outlier).  Q-B2.
__device__ __forceinline__ void classify_edge(const BaDev &B, int cur, int mode, int e, uint8_t *out) {
  if (mode == 1 && out[e]) return;
  const PoseCache P = pose_cache(B.Xc[cur] + 6 * B.e_cam[e]);
  double pc[3];
  trans_point(P, B.Xp[cur] + 3 * B.e_pt[e], pc);
  const float fx = (float)B.K.fx, fy = (float)B.K.fy, cx = (float)B.K.cx, cy = (float)B.K.cy, bf = (float)B.K.bf;
  const float x = (float)pc[0], y = (float)pc[1], z = (float)pc[2];
  const double ou = B.e_obs[3 * e], ov = B.e_obs[3 * e + 1], our = B.e_obs[3 * e + 2];
  bool outl;
  if (z < 0.0f) {
    outl = true;
  } else {
    const float invz = 1.0f / z;
    const float u = fx * x * invz + cx, v = fy * y * invz + cy;
    float eu, ev;
    if (mode == 0)
      eu = u - (float)ou, ev = v - (float)ov;
    else
      eu = (float)(u - ou), ev = (float)(v - ov);
    const float e2 = eu * eu + ev * ev;
    const float is2 = (float)(B.e_is[e] * B.e_is[e]);
    const bool mono = mode == 0 ? ((float)our < 0) : (our < 0);
    if (mono) {
      outl = e2 * is2 > 5.991f;
    } else {
      const float ur = u - bf * invz;
      const float eur = mode == 0 ? ur - (float)our : (float)(ur - our);
      outl = (e2 + eur * eur) * is2 > 7.815f;
    }
  }
  out[e] = outl;
  if (mode == 0) B.e_active[e] = !outl;
}

__global__ void k_ba_classify(BaDev B, int mode, uint8_t *out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < B.n_edges) classify_edge(B, B.st->cur, mode, e, out);
}

// Start of an LM solve in ONE launch (was clear / mark / begin, ~5 us of dispatch gap each):
//  * per edge: optional first-pass classification (local BA, between its two problems) or "all edges
//    active"; then edge activity -> which points / cameras are in the problem (Ceres drops unused
//    blocks), stamped with this solve's epoch so nothing has to be cleared first;
//  * block 0: reset the LM state (keeping the ping-pong index) and refresh the pose caches.
__global__ __launch_bounds__(256) void k_ba_setup(BaDev B, int set_active, uint8_t *classify0_out, int max_it,
                                                  int archive_slot, double hm, double hs) {
  BaState *S = B.st;
  const int cur = S->cur;  // never changes during this launch (the reset below keeps it)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < B.n_edges) {
    if (classify0_out) classify_edge(B, cur, 0, e, classify0_out);
    else if (set_active) B.e_active[e] = 1;
    if (B.e_active[e]) {
      B.pt_in[B.e_pt[e]] = (uint8_t)B.epoch;
      if (B.cam_slot[B.e_cam[e]] >= 0) B.cam_in[B.e_cam[e]] = (uint8_t)B.epoch;
    }
  }
  if (blockIdx.x != 0) return;
  for (int c = threadIdx.x; c < B.n_cams; c += blockDim.x) store_pc(B.PC[cur], c, pose_cache(B.Xc[cur] + 6 * c));
  if (threadIdx.x != 0) return;
  if (archive_slot >= 0) B.hist[archive_slot] = *S;
  BaState z;
  memset(&z, 0, sizeof(BaState));
  z.cur = cur;
  z.radius = 1e4;
  z.decrease = 2.0;
  z.max_it = max_it;
  z.first = 1;
  z.hm = hm;
  z.hs = hs;
  *S = z;  // `cur` is rewritten with its own value: readers in other blocks never see a different one
}

}  // namespace

// ============================================================================================
// host
// ============================================================================================
// process-wide developer knobs (vo_set_option)
static std::atomic<int> g_opt_ba_graph{0}, g_opt_pose_block{0}, g_opt_pairs_kernel{0};

// A host array that lives either in its own vector or -- the sorted edge arrays of an LDS-sized problem -- directly in the
// page-locked block build_device uploads from (ba_fill_problem writes them once, where the DMA reads them: at config 3 the
// intermediate copy of 1.1 MB was ~0.05 ms of every vo_ba_reset -> vo_ba_local_ba call).
template <class T>
struct HostArr {
  T *p = nullptr;
  size_t n = 0;
  std::vector<T> own;
  void resize(size_t m) { own.resize(m), p = own.data(), n = m; }
  void bind(T *ext, size_t m) { p = ext, n = m; }
  T *data() { return p; }
  const T *data() const { return p; }
  size_t size() const { return n; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};

struct vo_ba {
  int n_cams = 0, n_pts = 0, n_edges = 0, nf = 0;
  std::vector<double> poses, points;
  std::vector<uint8_t> cam_fixed;
  HostArr<int> e_cam, e_pt;  // sorted by point
  std::vector<int> perm;     // perm[sorted] = caller index
  HostArr<double> e_obs, e_is;
  std::vector<int> pt_start, cam_slot, slot_cam, fill_tmp, cam_count;  // (cam_count: edges per free-camera slot, all points)
  double cam[5];
  int shard = 0, n_shards = 1;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool built = false;
  vo::PinnedBuf pin;  // page-locked landing block of vo_ba_local_ba_finish (results of a solve)
  vo::PinnedBuf up_pin;   // page-locked staging of build_device's uploads: ONE block ...
  size_t up_used = 0;
  size_t pre_copied = 0;  // bytes at the start of the staging block whose copy to the device ba_fill_problem has already enqueued
  vo::DevBuf b_up_arena;  // ... mirrored by ONE device block: the uploaded buffers are views into it, one copy per build
  vo::DevBuf b_zero_arena;  // the buffers a build starts at zero, views into one block: one memset per build
  size_t zero_used = 0;
  bool arenas = false;    // this build uses the two arenas (LDS-sized systems; large ones keep separate buffers)
  bool state_cached = false;           // pin holds the final poses / points of the last vo_ba_local_ba (vo_ba_get_state without a round trip)
  size_t cache_xc_off = 0, cache_xp_off = 0;
  BaDev D{};
  vo::DevBuf b_ecam, b_ept, b_eobs, b_eis, b_eact, b_ptstart, b_local, b_camslot, b_slotcam, b_camstart,
      b_camedges, b_ptin, b_camin, b_xc0, b_xc1, b_xp0, b_xp1, b_pc0, b_pc1, b_sc, b_sp, b_hinv, b_gl2, b_dl, b_wt, b_wt1, b_hll0, b_hll1, b_spt1,
      b_sgemm, b_scam, b_spt, b_payload, b_zc, b_sbs, b_payload2, b_state, b_out, b_dbg, b_cnt;
  size_t solve_lds = 0, gemm_lds = 0;
  vo::DevBuf b_merge;  // sharded handles: payload of the closing point / erase-mask all-reduce
  vo::DevBuf b_we0, b_we1, b_glsc0, b_glsc1, b_Sd, b_scv, b_ddv, b_gppv, b_cholfail, b_pairstart, b_paircc, b_paire;  // large reduced systems
  int lm_max_it = 0;
  double *ext_payload = nullptr, *ext_payload2 = nullptr;
  int archive_slot = -1;
  bool lba_second = false;
  vo_allreduce_fn allreduce = nullptr;  // sums a device buffer over the shards, ordered on the handle's stream
  void *allreduce_user = nullptr;
  std::map<int, hipGraphExec_t> graphs;  // LM iteration sequences captured per iteration count
  vo::CholPlan *chol_plan = nullptr;     // large reduced systems: tile structure under the chosen key-frame order
  vo::DevBuf b_ltiles;                   // large systems: tiles of the factor (zeroed / assembled every iteration)
  vo::DevBuf b_packtiles, b_pack;        // sharded large systems: tiles of the matrix that exist, packed all-reduce payload
  int n_pack_tiles = 0;
  std::vector<int> pt_owner;             // shard of every point (p % n_shards; by nested-dissection segment in a segment solve)
  bool collectives = false;              // the LM loop runs its sharded form: n_shards > 1, or ONE shard with a callback and
                                         // VO_BA_COLLECTIVES_AT_ONE_RANK=1 (the callback path end to end on a one-GPU box)
  bool seg_mode = false;                 // sharded large system: per-rank segment factorisation
  int seg_c0 = 0;                        // first separator tile column
  vo::CholPlan *seg_plan[3] = {nullptr, nullptr, nullptr};  // chol_plan_create_split phases 1..3
  vo::DevBuf b_segtiles, b_segpack;      // separator tiles of the factor; payload of the separator / solution collectives
  int n_seg_tiles = 0;
  int order_parts = 1, order_cyclic = 0, order_sep = 0, order_depth = 0, order_tiles = 0;  // what choose_camera_order picked
  // vo_ba_set_option (before the first use of the handle; the same on every rank of a sharded solve -- checked by the
  // handshake all-reduce of build_device)
  int opt_segments = 0;       // VO_BA_OPT_SEGMENTS: per-rank segment factorisation
  int opt_collectives_1 = 0;  // VO_BA_OPT_COLLECTIVES_AT_ONE_RANK: one shard + callback runs the sharded form of the loop
  int opt_order_parts = -1;   // VO_BA_OPT_ORDER_PARTS: force the number of nested-dissection parts (1 = natural order)
};

namespace {

int upload(vo::DevBuf &b, const void *src, size_t bytes) {
  VO_CHECK(b.reserve(std::max<size_t>(bytes, 64)));
  if (bytes) VO_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  return VO_OK;
}
// A device buffer that grows with headroom: a handle that is re-used for problem after problem (vo_ba_reset) stops
// allocating after the first few (hipMalloc costs tens of microseconds, the hipFree behind a growth synchronises the device).
int reserve_grow(vo::DevBuf &b, size_t bytes) {
  if (b.view) b.release();  // a view into an earlier build's arena is not this buffer's storage (it kept aliasing the old offset)
  if (bytes <= b.bytes) return VO_OK;
  return b.reserve(bytes + bytes / 2 + 256);
}
// build_device's uploads of a handle: through ONE page-locked staging block (reserved by build_device for the whole build,
// so that it never moves under a copy in flight) and asynchronous on the handle's stream -- a hipMemcpy from pageable memory
// is a synchronous round trip of 10-30 us apiece, fourteen of them per handle
static size_t up_pin_bytes(int n_edges, int n_pts, int n_cams) {
  return 65536 + (size_t)n_edges * 56 + (size_t)n_pts * 64 + (size_t)n_cams * 256;
}
int upload(vo_ba *h, vo::DevBuf &b, const void *src, size_t bytes) {
  const size_t off = (h->up_used + 255) & ~(size_t)255;
  if (!h->arenas || off + std::max<size_t>(bytes, 64) > h->up_pin.bytes) {  // large systems; or more than build_device reserved
    VO_CHECK(reserve_grow(b, std::max<size_t>(bytes, 64)));
    if (bytes) VO_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return VO_OK;
  }
  if (bytes && src != h->up_pin.data() + off) memcpy(h->up_pin.data() + off, src, bytes);  // (ba_fill_problem may have put it there)
  h->up_used = off + std::max<size_t>(bytes, 64);
  b.set_view(static_cast<uint8_t *>(h->b_up_arena.p) + off, std::max<size_t>(bytes, 64));  // (the one copy follows at the end of the build)
  return VO_OK;
}
// a buffer that starts the build at zero: a view into the zero arena where this build uses it (one memset for all of them)
int zalloc(vo_ba *h, vo::DevBuf &b, size_t bytes) {
  bytes = std::max<size_t>(bytes, 64);
  const size_t off = (h->zero_used + 255) & ~(size_t)255;
  if (!h->arenas || off + bytes > h->b_zero_arena.bytes) {
    VO_CHECK(reserve_grow(b, bytes));
    VO_HIP_CHECK(hipMemsetAsync(b.p, 0, bytes, h->stream));
    return VO_OK;
  }
  h->zero_used = off + bytes;
  b.set_view(static_cast<uint8_t *>(h->b_zero_arena.p) + off, bytes);
  return VO_OK;
}

// ---- key-frame order of a large reduced system: vo::chol_choose_order (chol.hip) over the covisibility graph ------
vo::CholOrder choose_camera_order(const vo_ba *h, int m) {
  const int nf = h->nf;
  // covisible pairs of free key-frames over ALL points (every shard of a sharded solve must pick the same order, and
  // the factored matrix is the sum over the shards)
  std::vector<std::pair<int, int>> pairs;
  std::vector<uint8_t> seen((size_t)nf * nf, 0);
  for (int j = 0; j < h->n_pts; j++)
    for (int a = h->pt_start[j]; a < h->pt_start[j + 1]; a++) {
      const int sa = h->cam_slot[h->e_cam[a]];
      if (sa < 0) continue;
      for (int b = a + 1; b < h->pt_start[j + 1]; b++) {
        const int sb = h->cam_slot[h->e_cam[b]];
        if (sb < 0 || sb == sa) continue;
        const int lo = std::min(sa, sb), hi = std::max(sa, sb);
        if (!seen[(size_t)hi * nf + lo]) seen[(size_t)hi * nf + lo] = 1, pairs.push_back({hi, lo});
      }
    }
  return vo::chol_choose_order(nf, 6, pairs, m, h->opt_order_parts);  // (-1: choose; vo_ba_set_option(VO_BA_OPT_ORDER_PARTS))
}

// Every device-side initialisation below is enqueued on the handle's own (non-blocking) stream: a hipMemset on the NULL
// stream is asynchronous for device memory and does not order with that stream -- the zeroing of the LM state was seen
// to land after k_ba_setup had initialised it (a fresh handle whose allocations reuse recently freed memory).
int build_device(vo_ba *h) {
  if (h->built) return VO_OK;
  BaDev &D = h->D;
  // staging for every upload of this build (never moved while copies are in flight: reserved once, before the first)
  // (a previous build's copies out of the block have landed -- unless ba_fill_problem has just synchronised, filled the edge
  //  arrays and started THEIR copy, which this build neither touches nor has to wait for)
  if (!h->pre_copied) VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  VO_CHECK(h->up_pin.reserve(up_pin_bytes(h->n_edges, h->n_pts, h->n_cams)));  // (no-op when ba_fill_problem placed the edge arrays)
  h->up_used = 0;
  D.n_cams = h->n_cams, D.n_pts = h->n_pts, D.n_edges = h->n_edges, D.nf = h->nf;
  D.n_shards = h->n_shards, D.shard = h->shard;
  h->collectives = h->n_shards > 1 || (h->allreduce && h->opt_collectives_1);
  D.K = Cam{h->cam[0], h->cam[1], h->cam[2], h->cam[3], h->cam[4]};
  D.large = 6 * h->nf + 1 > kMaxN ? 1 : 0;
  D.ld = (6 * h->nf + vo::kCholPanel - 1) / vo::kCholPanel * vo::kCholPanel;
  if (D.large && D.ld > 4096) {
    vo::set_error("BA with %d free key-frames: the large-system path handles 6 nf <= 4096", h->nf);
    return VO_ERR_CAPACITY;
  }
  D.Mpad = D.large ? 16 : std::max(16, (6 * h->nf + 1 + 15) / 16 * 16);  // large: no dense operand matrix
  // LDS-sized systems (the per-key-frame local BA): the uploaded buffers are views into ONE device block filled by one copy
  // from the page-locked staging, the zero-initialised ones views into another cleared by one memset -- fourteen copies and
  // nine memsets otherwise, each a dependent ~3-5 us step in front of the first LM kernel
  h->arenas = !D.large;
  if (!h->arenas) h->pre_copied = 0;
  h->zero_used = 0;
  if (h->arenas) {
    VO_CHECK(reserve_grow(h->b_up_arena, h->up_pin.bytes));
    const size_t np_ = (size_t)std::max(1, h->n_pts);
    VO_CHECK(reserve_grow(h->b_zero_arena, np_ + (size_t)h->n_cams + 2 * np_ * 48 + 2 * 3 * np_ * D.Mpad * 8 +
                                               33 * (size_t)D.Mpad * D.Mpad * 8 + 4096 + 3 * sizeof(BaState) + 16 * 256 + 1024));
  }
  if (D.large) {
    const int m = D.ld / vo::kCholPanel;
    const vo::CholOrder o = choose_camera_order(h, m);
    std::vector<int> cams(h->slot_cam);  // natural free index -> camera
    for (int f = 0; f < h->nf; f++) {
      h->cam_slot[cams[f]] = o.slot_of[f];
      h->slot_cam[o.slot_of[f]] = cams[f];
    }
    h->order_parts = o.parts, h->order_cyclic = o.cyclic, h->order_sep = o.sep, h->order_depth = o.depth, h->order_tiles = o.tiles;
    h->chol_plan = vo::chol_plan_create(m, o.pattern.data());
    if (!h->chol_plan) {
      vo::set_error("BA: could not create the factorisation plan");
      return VO_ERR_HIP;
    }
    std::vector<unsigned long long> lmask((size_t)m);
    vo::chol_symbolic(m, o.pattern.data(), lmask.data(), nullptr, nullptr);
    h->pt_owner.resize((size_t)h->n_pts);
    for (int j = 0; j < h->n_pts; j++) h->pt_owner[j] = j % h->n_shards;
    // ---- per-rank segment factorisation (sharded, all-reduce callback registered, a nested-dissection order whose
    // segments end on a tile boundary).  A point is owned by the rank of the segment it touches -- the separators are as
    // wide as the covisibility band, so no point touches two segments (checked) --, points seen by separator key-frames
    // only are dealt out to even the load.  Then a rank's segment columns (separator rows included) are complete without
    // any exchange, and what has to be summed over the ranks is the separator block after the segments' elimination.
    {
      // Opt-in (vo_ba_set_option(h, VO_BA_OPT_SEGMENTS, 1)): measured with emulated ranks on one MI355X (tools/gba_seg_run.py) a
      // rank computes MORE per LM iteration this way than with the replicated factorisation of the all-reduced system
      // (config 4: 1.05 / 0.98 / 1.03 ms against 0.94 / 0.86 / 0.85 at 2 / 4 / 8 ranks) -- the separators' chain of dependent
      // tile columns, which every rank still runs, is three quarters of the factorisation, and the split adds launches,
      // flag resets and two small collectives.
      bool seg = h->n_shards > 1 && h->allreduce && !h->ext_payload && h->opt_segments && o.parts >= 2 &&
                 (int)o.part_of.size() == h->nf && o.seg_slots > 0 && (6 * o.seg_slots) % vo::kCholPanel == 0 &&
                 6 * o.seg_slots / vo::kCholPanel < m;
      std::vector<int> part_rank, pt_part;
      if (seg) {
        part_rank.resize((size_t)o.parts);
        for (int g = 0; g < o.parts; g++) part_rank[g] = g % h->n_shards;
        pt_part.assign((size_t)h->n_pts, -1);
        for (int j = 0; j < h->n_pts && seg; j++)
          for (int a = h->pt_start[j]; a < h->pt_start[j + 1]; a++) {
            const int sl = h->cam_slot[h->e_cam[a]];
            if (sl < 0 || sl >= o.seg_slots) continue;
            const int g = o.part_of[sl];
            if (pt_part[j] >= 0 && pt_part[j] != g) seg = false;  // (cannot happen with separators as wide as the band)
            pt_part[j] = g;
          }
        // a segment's tile columns: whole tiles (segment lengths are multiples of 32 key-frames, the end was checked)
        for (int sl = 0; sl + 1 < o.seg_slots && seg; sl++)
          if (o.part_of[sl] != o.part_of[sl + 1] && (6 * (sl + 1)) % vo::kCholPanel != 0) seg = false;
      }
      if (seg) {
        std::vector<long long> load((size_t)h->n_shards, 0);
        for (int j = 0; j < h->n_pts; j++)
          if (pt_part[j] >= 0) {
            h->pt_owner[j] = part_rank[pt_part[j]];
            load[h->pt_owner[j]] += h->pt_start[j + 1] - h->pt_start[j];
          }
        for (int j = 0; j < h->n_pts; j++)
          if (pt_part[j] < 0) {
            int best = 0;
            for (int r = 1; r < h->n_shards; r++)
              if (load[r] < load[best]) best = r;
            h->pt_owner[j] = best;
            load[best] += h->pt_start[j + 1] - h->pt_start[j];
          }
        h->seg_mode = true;
        h->seg_c0 = 6 * o.seg_slots / vo::kCholPanel;
        unsigned long long own = 0;
        for (int sl = 0; sl < o.seg_slots; sl++)
          if (part_rank[o.part_of[sl]] == h->shard) own |= (1ull << (6 * sl / vo::kCholPanel)) | (1ull << ((6 * sl + 5) / vo::kCholPanel));
        for (int ph = 0; ph < 3; ph++) {
          h->seg_plan[ph] = vo::chol_plan_create_split(m, o.pattern.data(), h->seg_c0, own, ph + 1);
          if (!h->seg_plan[ph]) {
            vo::set_error("BA: could not create the segment factorisation plans");
            return VO_ERR_HIP;
          }
        }
        D.seg_mode = 1, D.seg_row0 = vo::kCholPanel * h->seg_c0, D.seg_lead = h->shard == 0 ? 1 : 0, D.seg_own = own;
        std::vector<int2> st;
        for (int i = h->seg_c0; i < m; i++)
          for (int j = h->seg_c0; j <= i; j++)
            if ((lmask[i] >> j) & 1ull) st.push_back(make_int2(i, j));
        h->n_seg_tiles = (int)st.size();
        VO_CHECK(upload(h, h->b_segtiles, st.data(), st.size() * sizeof(int2)));
        VO_CHECK(reserve_grow(h->b_segpack, ((size_t)st.size() * vo::kCholPanel * vo::kCholPanel + (size_t)D.ld + 8) * 8));
      }
    }
    {
      // the tiles this handle zeroes and assembles: all of the factor's -- or, in a segment solve, its own segments'
      // columns and the separator block (flagged partial where another rank adds the replicated terms)
      std::vector<int2> lt;
      for (int i = 0; i < m; i++)
        for (int j = 0; j <= i; j++) {
          if (!((lmask[i] >> j) & 1ull)) continue;
          if (h->seg_mode && j < h->seg_c0 && !((D.seg_own >> j) & 1ull)) continue;
          const int partial = h->seg_mode && j >= h->seg_c0 && !D.seg_lead ? 1 : 0;
          lt.push_back(make_int2(i, j | (partial << 16)));
        }
      VO_CHECK(upload(h, h->b_ltiles, lt.data(), lt.size() * sizeof(int2)));
      D.n_ltiles = (int)lt.size();
    }
    if (h->collectives && !h->seg_mode) {
      std::vector<int2> tiles;
      for (int i = 0; i < m; i++)
        for (int j = 0; j <= i; j++)
          if ((o.pattern[i] >> j) & 1ull) tiles.push_back(make_int2(i, j));
      h->n_pack_tiles = (int)tiles.size();
      VO_CHECK(upload(h, h->b_packtiles, tiles.data(), tiles.size() * sizeof(int2)));
      VO_CHECK(reserve_grow(h->b_pack, ((size_t)tiles.size() * vo::kCholPanel * vo::kCholPanel + D.ld + (size_t)h->nf * 27 + 1 + h->n_shards) * 8));
    }
  }
  if (h->pt_owner.empty()) {
    h->pt_owner.resize((size_t)h->n_pts);
    for (int j = 0; j < h->n_pts; j++) h->pt_owner[j] = j % h->n_shards;
  }
  const bool all_local = h->n_shards == 1;  // (every point is this shard's: no owner look-ups, and ba_fill_problem has counted the cameras' edges)
  std::vector<int> local;
  if (all_local) {
    local.resize((size_t)h->n_pts);
    std::iota(local.begin(), local.end(), 0);
  } else {
    for (int j = 0; j < h->n_pts; j++)
      if (h->pt_owner[j] == h->shard) local.push_back(j);
  }
  D.n_local = (int)local.size();
  // per-camera edge lists restricted to this shard's points
  std::vector<int> cstart(h->nf + 1, 0), cedges;
  {
    // counting sort of this shard's edges by free-camera slot (edge order within a camera = sorted edge order)
    if (all_local && !D.large && (int)h->cam_count.size() == h->nf + 1) {  // (a large system has re-ordered its camera slots above)
      for (int sl = 0; sl < h->nf; sl++) cstart[sl + 1] = h->cam_count[sl];
    } else {
      for (int e = 0; e < h->n_edges; e++) {
        const int sl = h->cam_slot[h->e_cam[e]];
        if (sl >= 0 && h->pt_owner[h->e_pt[e]] == h->shard) cstart[sl + 1]++;
      }
    }
    int mx = 0;
    for (int sl = 0; sl < h->nf; sl++) mx = std::max(mx, cstart[sl + 1]), cstart[sl + 1] += cstart[sl];
    cedges.resize((size_t)cstart[h->nf]);
    std::vector<int> fill(cstart.begin(), cstart.end() - 1);
    const int *slot = h->cam_slot.data(), *ec = h->e_cam.data();
    if (all_local) {
      for (int e = 0; e < h->n_edges; e++) {
        const int sl = slot[ec[e]];
        if (sl >= 0) cedges[(size_t)fill[sl]++] = e;
      }
    } else {
      for (int e = 0; e < h->n_edges; e++) {
        const int sl = slot[ec[e]];
        if (sl >= 0 && h->pt_owner[h->e_pt[e]] == h->shard) cedges[(size_t)fill[sl]++] = e;
      }
    }
    D.n_cchunks = std::max(1, (mx + kCamChunk - 1) / kCamChunk);
  }
  D.n_pblocks = std::max(1, (D.n_local + kPtsPerBlock - 1) / kPtsPerBlock);
  const int K = 3 * h->n_pts;
  const int tiles = (D.Mpad / 16) * (D.Mpad / 16 + 1) / 2;
  // the MFMA loop is latency-bound per wave (~1.5 us per 48-row trip): many short K slices, <= 32 slabs
  int ks = std::max(1, std::min(32, 256 / tiles));
  ks = std::min(ks, std::max(1, K / 128));
  D.kchunk = ((K + ks - 1) / ks + 47) / 48 * 48;  // multiple of 16 (MFMA slices) and of 3 (whole points)
  D.ksplit = std::max(1, (K + D.kchunk - 1) / D.kchunk);
  if (!D.large && (D.kchunk / 3 > kChunkPts || D.ksplit > 32)) {
    vo::set_error("local BA with %d points exceeds this round's dense Schur path (%d points per K slice, 32 slices)",
                  h->n_pts, kChunkPts);
    return VO_ERR_CAPACITY;
  }
  VO_CHECK(upload(h, h->b_ecam, h->e_cam.data(), h->e_cam.size() * 4));
  VO_CHECK(upload(h, h->b_ept, h->e_pt.data(), h->e_pt.size() * 4));
  VO_CHECK(upload(h, h->b_eobs, h->e_obs.data(), h->e_obs.size() * 8));
  VO_CHECK(upload(h, h->b_eis, h->e_is.data(), h->e_is.size() * 8));
  VO_CHECK(reserve_grow(h->b_eact, std::max<size_t>(h->n_edges, 64)));
  VO_CHECK(reserve_grow(h->b_out, std::max<size_t>(h->n_edges, 64)));
  VO_CHECK(upload(h, h->b_ptstart, h->pt_start.data(), h->pt_start.size() * 4));
  VO_CHECK(upload(h, h->b_local, local.data(), local.size() * 4));
  VO_CHECK(upload(h, h->b_camslot, h->cam_slot.data(), h->cam_slot.size() * 4));
  VO_CHECK(upload(h, h->b_slotcam, h->slot_cam.data(), h->slot_cam.size() * 4));
  VO_CHECK(upload(h, h->b_camstart, cstart.data(), cstart.size() * 4));
  VO_CHECK(upload(h, h->b_camedges, cedges.data(), cedges.size() * 4));
  VO_CHECK(zalloc(h, h->b_ptin, std::max<size_t>(h->n_pts, 64)));  // epoch stamps start below epoch 1
  VO_CHECK(zalloc(h, h->b_camin, std::max<size_t>(h->n_cams, 64)));
  D.epoch = 0;
  VO_CHECK(upload(h, h->b_xc0, h->poses.data(), h->poses.size() * 8));
  VO_CHECK(upload(h, h->b_xc1, h->poses.data(), h->poses.size() * 8));
  VO_CHECK(upload(h, h->b_xp0, h->points.data(), h->points.size() * 8));
  VO_CHECK(upload(h, h->b_xp1, h->points.data(), h->points.size() * 8));
  VO_CHECK(reserve_grow(h->b_pc0, (size_t)h->n_cams * 12 * 8));
  VO_CHECK(reserve_grow(h->b_pc1, (size_t)h->n_cams * 12 * 8));
  VO_CHECK(reserve_grow(h->b_sc, (size_t)std::max(1, h->n_cams) * 6 * 8));
  VO_CHECK(reserve_grow(h->b_sp, (size_t)std::max(1, h->n_pts) * 3 * 8));
  VO_CHECK(reserve_grow(h->b_hinv, (size_t)std::max(1, h->n_pts) * 6 * 8));
  VO_CHECK(reserve_grow(h->b_gl2, (size_t)std::max(1, h->n_pts) * 3 * 8));
  VO_CHECK(reserve_grow(h->b_dl, (size_t)std::max(1, h->n_pts) * 3 * 8));
  const size_t wt_rows = D.large ? 1 : (size_t)std::max(1, K);
  // operand matrices start out all-zero; only (point, camera) pairs that have an edge are ever written
  VO_CHECK(zalloc(h, h->b_wt, wt_rows * D.Mpad * 8));
  VO_CHECK(zalloc(h, h->b_wt1, wt_rows * D.Mpad * 8));
  VO_CHECK(zalloc(h, h->b_hll0, (size_t)std::max(1, h->n_pts) * 6 * 8));
  VO_CHECK(zalloc(h, h->b_hll1, (size_t)std::max(1, h->n_pts) * 6 * 8));
  if (D.large) D.ksplit = 1;
  VO_CHECK(zalloc(h, h->b_sgemm, (size_t)(D.ksplit + 1) * D.Mpad * D.Mpad * 8));  // (the slab behind the last one must be all-zero)
  if (D.large) {
    // covisible camera pairs (c <= c') and, per pair, the (edge of c, edge of c') couples at their shared
    // points in point order: the gather lists of k_ba_pairs
    struct Couple {
      long long key;
      int e, ep;
    };
    std::vector<Couple> cp;
    for (int j : local) {
      for (int a = h->pt_start[j]; a < h->pt_start[j + 1]; a++) {
        const int sa = h->cam_slot[h->e_cam[a]];
        if (sa < 0) continue;
        for (int b = h->pt_start[j]; b < h->pt_start[j + 1]; b++) {
          const int sb = h->cam_slot[h->e_cam[b]];
          if (sb < sa || (sb == sa && b != a)) continue;  // c <= c'; a camera sees a point once
          cp.push_back({(long long)sa * h->nf + sb, a, b});
        }
      }
    }
    std::stable_sort(cp.begin(), cp.end(), [](const Couple &x, const Couple &y) { return x.key < y.key; });
    std::vector<int> pstart, pcc, pe, first;
    pe.reserve(4 * cp.size());
    for (size_t i = 0; i < cp.size(); i++) {
      if (i == 0 || cp[i].key != cp[i - 1].key) first.push_back((int)i);
      pe.push_back(cp[i].e), pe.push_back(cp[i].ep), pe.push_back(h->e_pt[cp[i].e]), pe.push_back(0);
    }
    first.push_back((int)cp.size());
    D.n_pairs = (int)first.size() - 1;
    // Work order.  First the cameras' own "pairs" (c, c): 5-10 x longer lists than the rest, they must not be what
    // the kernel ends on.  Then the covisible pairs in list order (c, then c'): neighbouring cameras share their
    // points, so consecutive pairs read the same W blocks -- measured 322 us in this order against 377 us sorted by
    // length and 427 us with the diagonal pairs in place.  Both parts are padded with empty pairs to whole groups of
    // 8 workgroups; the kernel deals the second part to the XCDs in contiguous eighths.
    std::vector<int> order;
    auto is_diag = [&](int w) { return cp[first[w]].key / h->nf == cp[first[w]].key % h->nf; };
    for (int w = 0; w < D.n_pairs; w++)
      if (is_diag(w)) order.push_back(w);
    while (order.size() % 32) order.push_back(-1);
    D.pair_diag_blocks = (int)order.size() / 4;
    for (int w = 0; w < D.n_pairs; w++)
      if (!is_diag(w)) order.push_back(w);
    while (order.size() % 32) order.push_back(-1);
    D.n_pairs = (int)order.size();
    for (int w : order) {
      if (w < 0) {
        pstart.push_back(0), pstart.push_back(0), pcc.push_back(0), pcc.push_back(0);
        continue;
      }
      pstart.push_back(first[w]), pstart.push_back(first[w + 1]);
      pcc.push_back((int)(cp[first[w]].key / h->nf)), pcc.push_back((int)(cp[first[w]].key % h->nf));
    }
    if (pe.empty()) pe.assign(4, 0);
    VO_CHECK(upload(h, h->b_pairstart, pstart.data(), pstart.size() * 4));
    VO_CHECK(upload(h, h->b_paircc, pcc.data(), pcc.size() * 4));
    VO_CHECK(upload(h, h->b_paire, pe.data(), pe.size() * 4));
    VO_CHECK(reserve_grow(h->b_we0, (size_t)std::max(1, h->n_edges) * 18 * 8));
    VO_CHECK(reserve_grow(h->b_we1, (size_t)std::max(1, h->n_edges) * 18 * 8));
    VO_HIP_CHECK(hipMemsetAsync(h->b_we0.p, 0, (size_t)std::max(1, h->n_edges) * 18 * 8, h->stream));
    VO_HIP_CHECK(hipMemsetAsync(h->b_we1.p, 0, (size_t)std::max(1, h->n_edges) * 18 * 8, h->stream));
    VO_CHECK(reserve_grow(h->b_glsc0, (size_t)std::max(1, h->n_pts) * 3 * 8));
    VO_CHECK(reserve_grow(h->b_glsc1, (size_t)std::max(1, h->n_pts) * 3 * 8));
    VO_CHECK(reserve_grow(h->b_Sd, (size_t)(D.ld + vo::kCholPanel) * D.ld * 8));  // + right-hand side / solution rows
    // (once: the tiles outside the plan are never written again, and a caller's all-reduce of the whole storage -- the
    // split-phase interface -- must not sum uninitialised memory)
    VO_HIP_CHECK(hipMemsetAsync(h->b_Sd.p, 0, (size_t)(D.ld + vo::kCholPanel) * D.ld * 8, h->stream));
    VO_CHECK(reserve_grow(h->b_scv, (size_t)D.ld * 8));
    VO_CHECK(reserve_grow(h->b_ddv, (size_t)D.ld * 8));
    VO_CHECK(reserve_grow(h->b_gppv, (size_t)D.ld * 8));
    VO_CHECK(reserve_grow(h->b_cholfail, vo::chol_workspace_bytes(D.ld)));
    D.We[0] = h->b_we0.as<double>(), D.We[1] = h->b_we1.as<double>();
    D.glsc[0] = h->b_glsc0.as<double>(), D.glsc[1] = h->b_glsc1.as<double>();
    D.Sd = h->ext_payload ? h->ext_payload : h->b_Sd.as<double>();
    if (h->ext_payload) VO_HIP_CHECK(hipMemsetAsync(h->ext_payload, 0, (size_t)(D.ld + vo::kCholPanel) * D.ld * 8, h->stream));
    D.sc_v = h->b_scv.as<double>(), D.Dd_v = h->b_ddv.as<double>(), D.gpp_v = h->b_gppv.as<double>();
    D.chol_fail = h->b_cholfail.as<int>();
    D.pair_start = h->b_pairstart.as<int>(), D.pair_cc = h->b_paircc.as<int>(), D.pair_e = h->b_paire.as<int4>();
    D.ltiles = h->b_ltiles.as<int2>();
  }
  VO_CHECK(reserve_grow(h->b_scam, (size_t)std::max(1, h->nf) * D.n_cchunks * 27 * 8));
  VO_CHECK(reserve_grow(h->b_spt, (size_t)D.n_pblocks * 2 * 8));
  VO_CHECK(reserve_grow(h->b_spt1, (size_t)D.n_pblocks * 2 * 8));
  VO_CHECK(reserve_grow(h->b_payload, ((size_t)D.Mpad * D.Mpad + (size_t)h->nf * 27 + 1 + h->n_shards) * 8));
  VO_CHECK(reserve_grow(h->b_zc, (size_t)std::max(1, 6 * h->nf) * 8));
  VO_CHECK(reserve_grow(h->b_sbs, (size_t)D.n_pblocks * 6 * 8));
  VO_CHECK(reserve_grow(h->b_payload2, 64));
  VO_CHECK(zalloc(h, h->b_state, 3 * sizeof(BaState)));
  VO_CHECK(reserve_grow(h->b_dbg, 64 * 8));
  VO_CHECK(zalloc(h, h->b_cnt, 4096));
  D.counters = h->b_cnt.as<unsigned int>();
  D.fused = h->collectives ? 0 : 1;
  D.div_np1 = (unsigned)((0x100000000ull + (unsigned)(6 * h->nf)) / (unsigned)(6 * h->nf + 1));
  D.dbg = h->b_dbg.as<unsigned long long>();
  if (h->arenas) {  // everything this build uploads: one copy; everything that starts at zero: one memset
    const size_t done = std::min(h->pre_copied, h->up_used);  // the edge arrays are already on their way (ba_fill_problem)
    if (h->up_used > done)
      VO_HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(h->b_up_arena.p) + done, h->up_pin.data() + done, h->up_used - done,
                                  hipMemcpyHostToDevice, h->stream));
    if (h->zero_used) VO_HIP_CHECK(hipMemsetAsync(h->b_zero_arena.p, 0, h->zero_used, h->stream));
  }
  D.e_cam = h->b_ecam.as<int>(), D.e_pt = h->b_ept.as<int>();
  D.e_obs = h->b_eobs.as<double>(), D.e_is = h->b_eis.as<double>();
  D.e_active = h->b_eact.as<uint8_t>();
  D.pt_start = h->b_ptstart.as<int>(), D.local_pts = h->b_local.as<int>();
  D.cam_slot = h->b_camslot.as<int>(), D.slot_cam = h->b_slotcam.as<int>();
  D.cam_start = h->b_camstart.as<int>(), D.cam_edges = h->b_camedges.as<int>();
  D.pt_in = h->b_ptin.as<uint8_t>(), D.cam_in = h->b_camin.as<uint8_t>();
  D.Xc[0] = h->b_xc0.as<double>(), D.Xc[1] = h->b_xc1.as<double>();
  D.Xp[0] = h->b_xp0.as<double>(), D.Xp[1] = h->b_xp1.as<double>();
  D.PC[0] = h->b_pc0.as<double>(), D.PC[1] = h->b_pc1.as<double>();
  D.scale_c = h->b_sc.as<double>(), D.scale_p = h->b_sp.as<double>();
  D.hinv = h->b_hinv.as<double>(), D.gl2 = h->b_gl2.as<double>(), D.dl = h->b_dl.as<double>();
  D.Wt[0] = h->b_wt.as<double>(), D.Wt[1] = h->b_wt1.as<double>();
  D.hll[0] = h->b_hll0.as<double>(), D.hll[1] = h->b_hll1.as<double>();
  D.slab_pt[0] = h->b_spt.as<double>(), D.slab_pt[1] = h->b_spt1.as<double>();
  D.slab_gemm = h->b_sgemm.as<double>(), D.slab_cam = h->b_scam.as<double>();
  D.payload = h->ext_payload ? h->ext_payload : h->b_payload.as<double>();
  D.zc = h->b_zc.as<double>();
  D.slab_bs = h->b_sbs.as<double>();
  D.payload2 = h->ext_payload2 ? h->ext_payload2 : h->b_payload2.as<double>();
  D.st = h->b_state.as<BaState>();
  D.hist = D.st + 1;
  const int n = D.large ? 6 : 6 * h->nf;
  h->solve_lds = ((size_t)(n + 1) * (n + 1) + 5 * n + (size_t)h->nf * 21 + kSolveRed + (size_t)h->nf * 27 + 8) * 8;
  if (!D.large && h->solve_lds > 64 * 1024)
    VO_HIP_CHECK(hipFuncSetAttribute((const void *)k_ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->solve_lds));
  h->gemm_lds = (size_t)kGemmLdsDoubles * 8;
  if (h->gemm_lds > 64 * 1024)
    VO_HIP_CHECK(hipFuncSetAttribute((const void *)k_ba_gemm, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->gemm_lds));
  if (h->collectives && h->allreduce) {
    // Handshake: the collective schedule of the LM loop (how many all-reduces per iteration, of how many doubles) follows
    // from the options and the problem, and ranks that disagree would wait for each other forever.  One tiny all-reduce
    // before any solve: every rank contributes (c, c^2, 1, g) -- c = its 16-bit protocol word (protocol bits, shard count,
    // five bits of a hash of the problem sizes), g = twenty more bits of that hash (64-bit arithmetic).  All words are
    // equal iff n sum(c^2) == sum(c)^2, exact in doubles up to 2^10 ranks (with the former 20-bit word it was
    // exact up to ~90); sum(g) == n g catches a different problem in all but one case in 2^20; sum(1) is the number of
    // ranks that answered.
    const unsigned long long hash = (unsigned long long)h->n_edges * 31ull + (unsigned long long)h->n_pts * 7ull + (unsigned long long)h->nf;
    const unsigned proto = (1u + (h->seg_mode ? 1u : 0u) + (D.large ? 2u : 0u)) | ((unsigned)(h->n_shards & 0xff) << 3) |
                           ((unsigned)(hash & 31ull) << 11);
    const double c = (double)proto, g = (double)((hash >> 5) & 0xfffffull);
    double hs[4] = {c, c * c, 1.0, g};
    VO_CHECK(reserve_grow(h->b_merge, 64));
    VO_HIP_CHECK(hipMemcpyAsync(h->b_merge.p, hs, sizeof(hs), hipMemcpyHostToDevice, h->stream));
    const int rc = h->allreduce(h->allreduce_user, h->b_merge.as<double>(), 4, (void *)h->stream);
    if (rc != 0) {
      vo::set_error("BA all-reduce callback failed with status %d (handshake)", rc);
      return VO_ERR_HIP;
    }
    VO_HIP_CHECK(hipMemcpyAsync(hs, h->b_merge.p, sizeof(hs), hipMemcpyDeviceToHost, h->stream));
    VO_HIP_CHECK(hipStreamSynchronize(h->stream));
    const double nr = hs[2];
    if (nr != (double)h->n_shards || nr * hs[1] != hs[0] * hs[0] || hs[3] != nr * g) {
      vo::set_error("sharded BA: the ranks disagree on the collective protocol (options / shard count / problem): %g ranks answered, "
                    "%d configured, protocol word %u here (vo_ba_set_option and vo_ba_set_shard must be the same on every rank)",
                    nr, h->n_shards, proto);
      return VO_ERR_INVALID;
    }
  }
  h->built = true;
  return VO_OK;
}

int lm_begin(vo_ba *h, double hm, double hs, int max_it, const uint8_t *active_caller, bool keep_device_mask,
             uint8_t *classify0_out = nullptr) {
  VO_CHECK(build_device(h));
  h->state_cached = false;  // a solve is about to move the state
  BaDev &D = h->D;
  hipStream_t st = h->stream;
  int set_active = 0;
  if (!keep_device_mask) {
    if (!active_caller) {
      set_active = 1;
    } else {
      std::vector<uint8_t> act(std::max(1, h->n_edges), 1);
      for (int e = 0; e < h->n_edges; e++) act[e] = active_caller[h->perm[e]] ? 1 : 0;
      VO_HIP_CHECK(hipMemcpyAsync(D.e_active, act.data(), h->n_edges, hipMemcpyHostToDevice, st));
      VO_HIP_CHECK(hipStreamSynchronize(st));  // act is a stack-lifetime buffer
    }
  }
  if (++D.epoch > 255) {  // the membership stamps are bytes: clear them when the epoch wraps
    const int nmax = std::max(std::max(h->n_edges, h->n_pts), h->n_cams);
    hipLaunchKernelGGL(k_ba_clear, dim3((nmax + 255) / 256), dim3(256), 0, st, D, 0, (uint8_t *)nullptr);
    D.epoch = 1;
  }
  hipLaunchKernelGGL(k_ba_setup, dim3((std::max(1, h->n_edges) + 255) / 256), dim3(256), 0, st, D, set_active,
                     classify0_out, max_it, h->archive_slot, hm, hs);
  hipLaunchKernelGGL(k_ba_lin0, dim3(D.n_pblocks), dim3(256), 0, st, D);  // first linearisation of the solve
  VO_HIP_CHECK(hipGetLastError());
  h->lm_max_it = max_it;
  return VO_OK;
}

// large-system path, first half of an LM iteration: this shard's part of the reduced system (and of the
// camera blocks / cost) into the buffer a multi-GPU driver all-reduces
int launch_linearize_large(vo_ba *h) {
  BaDev &D = h->D;
  hipStream_t st = h->stream;
  hipLaunchKernelGGL(k_ba_hinv_large, dim3((D.n_local + 255) / 256), dim3(256), 0, st, D);
  hipLaunchKernelGGL(k_ba_cams_large, dim3(std::max(1, h->nf * D.n_cchunks)), dim3(kCamChunk), 0, st, D);
  hipLaunchKernelGGL(k_ba_zero_large, dim3(D.n_ltiles + 8), dim3(256), 0, st, D);
  if (D.n_pairs > 0) {  // n_pairs: a multiple of 32
    if (g_opt_pairs_kernel.load(std::memory_order_relaxed) == 0)
      hipLaunchKernelGGL(k_ba_pairs_lds, dim3(D.n_pairs / 4), dim3(256), 0, st, D);
    else
      hipLaunchKernelGGL(k_ba_pairs, dim3(D.n_pairs / 4), dim3(256), 0, st, D);
  }
  hipLaunchKernelGGL(k_ba_partials_large, dim3(std::max(1, (h->nf * 27 + 255) / 256)), dim3(256), 0, st, D);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}
// second half: damped system, factorisation, step, candidate poses (k_ba_backsub follows)
int shard_allreduce(vo_ba *h, double *buf, size_t n);
int launch_step_large(vo_ba *h) {
  BaDev &D = h->D;
  hipStream_t st = h->stream;
  hipLaunchKernelGGL(k_ba_prestep_large, dim3(1), dim3(256), 0, st, D);
  hipLaunchKernelGGL(k_ba_assemble_large, dim3(D.n_ltiles + 8), dim3(256), 0, st, D);
  if (h->seg_mode) {
    // Per-rank segment factorisation (csrc/chol.hip, split plans): this rank's segments are eliminated into the separator
    // block; the separator block -- the only part of the reduced system that is a sum over the ranks -- is all-reduced and
    // solved by every rank; the segments' unknowns follow by back-substitution and the step is gathered by a second,
    // small all-reduce (zeros outside the own columns).  DESIGN.md section 6.
    constexpr int T = vo::kCholPanel;
    double *pk = h->b_segpack.as<double>();
    const int2 *tiles = h->b_segtiles.as<int2>();
    const size_t n_sep = (size_t)h->n_seg_tiles * T * T + (size_t)(D.ld - D.seg_row0);
    vo::chol_split_phase(D.Sd, D.ld, D.chol_fail, st, h->seg_plan[0], 1, h->seg_c0);
    hipLaunchKernelGGL(k_ba_pack_seg, dim3(h->n_seg_tiles + 4), dim3(256), 0, st, D, tiles, h->n_seg_tiles, pk, 0, 1);
    VO_CHECK(shard_allreduce(h, pk, n_sep));
    hipLaunchKernelGGL(k_ba_pack_seg, dim3(h->n_seg_tiles + 4), dim3(256), 0, st, D, tiles, h->n_seg_tiles, pk, 1, 1);
    vo::chol_split_phase(D.Sd, D.ld, D.chol_fail, st, h->seg_plan[1], 2, h->seg_c0);
    vo::chol_split_phase(D.Sd, D.ld, D.chol_fail, st, h->seg_plan[2], 3, h->seg_c0);
    hipLaunchKernelGGL(k_ba_pack_seg, dim3(16), dim3(256), 0, st, D, tiles, 0, pk, 0, 2);
    VO_CHECK(shard_allreduce(h, pk, (size_t)D.ld + 1));
    hipLaunchKernelGGL(k_ba_pack_seg, dim3(16), dim3(256), 0, st, D, tiles, 0, pk, 1, 2);
  } else {
    vo::chol_factor_solve(D.Sd, D.ld, D.chol_fail, st, h->chol_plan);
  }
  hipLaunchKernelGGL(k_ba_poststep_large, dim3(1), dim3(256), 0, st, D);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int launch_linearize(vo_ba *h) {
  BaDev &D = h->D;
  hipStream_t st = h->stream;
  if (D.large) return launch_linearize_large(h);
  const int tdim = D.Mpad / 16, tiles = tdim * (tdim + 1) / 2;
  // Schur product tiles and the camera blocks in one launch (independent roles)
  hipLaunchKernelGGL(k_ba_gemm, dim3(tiles * D.ksplit + h->nf * D.n_cchunks), dim3(kGemmThreads), h->gemm_lds, st, D);
  if (!D.fused) {
    const int np = (D.Mpad * D.Mpad + h->nf * 27 + 1) * 4;
    hipLaunchKernelGGL(k_ba_reduce, dim3((np + 255) / 256), dim3(256), 0, st, D);
  }
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}
int launch_step(vo_ba *h) {
  BaDev &D = h->D;
  hipStream_t st = h->stream;
  if (D.large)
    VO_CHECK(launch_step_large(h));
  else
    hipLaunchKernelGGL(k_ba_solve, dim3(1), dim3(kSolveThreads), h->solve_lds, st, D);
  hipLaunchKernelGGL(k_ba_backsub, dim3(D.n_pblocks), dim3(256), 0, st, D);
  if (h->collectives) hipLaunchKernelGGL(k_ba_reduce2, dim3(1), dim3(64), 0, st, D);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}
int launch_update(vo_ba *h) {
  if (h->D.fused) return VO_OK;  // done by the last block of k_ba_backsub
  hipLaunchKernelGGL(k_ba_update, dim3(1), dim3(64), 0, h->stream, h->D);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int lm_end(vo_ba *h, vo_lm_summary *sum) {
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  BaState s;
  VO_HIP_CHECK(hipMemcpy(&s, h->D.st, sizeof(s), hipMemcpyDeviceToHost));
  if (sum) {
    sum->iterations = s.iter;
    sum->accepted = s.accepted;
    sum->termination = s.termination;
    sum->reserved = 0;
    sum->initial_cost = s.initial_cost;
    sum->final_cost = s.x_cost;
    sum->final_radius = s.radius;
  }
  return VO_OK;
}

void drop_graphs(vo_ba *h) {
  for (auto &g : h->graphs) (void)hipGraphExecDestroy(g.second);
  h->graphs.clear();
}

// one collective of the sharded LM loop: the caller's all-reduce (RCCL over xGMI in production) on the handle's stream
int shard_allreduce(vo_ba *h, double *buf, size_t n) {
  const int rc = h->allreduce(h->allreduce_user, buf, n, (void *)h->stream);
  if (rc != 0) {
    vo::set_error("BA all-reduce callback failed with status %d", rc);
    return VO_ERR_HIP;
  }
  return VO_OK;
}

int run_lm_eager(vo_ba *h, int max_it) {
  const bool sharded = h->collectives;
  double *p1 = nullptr, *p2 = nullptr;
  size_t n1 = 0, n2 = 0;
  if (sharded) {  // exactly two collectives per LM iteration; every control decision is a
                  // function of all-reduced values, so the shards stay in lock-step without a host sync
    VO_CHECK(vo_ba_reduced_system(h, &p1, &n1));
    VO_CHECK(vo_ba_reduced_cost(h, &p2, &n2));
  }
  const bool packed = sharded && h->D.large && h->n_pack_tiles > 0;
  const size_t n_pack = packed ? (size_t)h->n_pack_tiles * vo::kCholPanel * vo::kCholPanel + h->D.ld + (size_t)h->nf * 27 + 1 + h->n_shards : 0;
  for (int it = 0; it < max_it; it++) {
    VO_CHECK(launch_linearize(h));
    if (packed) {  // only the tiles that exist, the right-hand side and the extras travel (k_ba_pack_large)
      const dim3 grid(h->n_pack_tiles + 8);
      hipLaunchKernelGGL(k_ba_pack_large, grid, dim3(256), 0, h->stream, h->D, h->b_packtiles.as<int2>(), h->n_pack_tiles, h->b_pack.as<double>(), 0);
      VO_CHECK(shard_allreduce(h, h->b_pack.as<double>(), n_pack));
      hipLaunchKernelGGL(k_ba_pack_large, grid, dim3(256), 0, h->stream, h->D, h->b_packtiles.as<int2>(), h->n_pack_tiles, h->b_pack.as<double>(), 1);
      VO_HIP_CHECK(hipGetLastError());
    } else if (sharded && h->seg_mode) {
      // segment solve: only the camera-block extras (per key-frame 27 sums, the cost, the gradient maxima) are summed
      // here; the separator block and the step follow inside launch_step_large
      VO_CHECK(shard_allreduce(h, h->D.Sd + large_ext_off(h->D.ld), (size_t)h->nf * 27 + 1 + h->n_shards));
    } else if (sharded) {
      VO_CHECK(shard_allreduce(h, p1, n1));
    }
    VO_CHECK(launch_step(h));
    if (sharded) VO_CHECK(shard_allreduce(h, p2, n2));
    VO_CHECK(launch_update(h));
  }
  return VO_OK;
}

// The iteration sequence has fixed launch parameters (everything that varies lives in BaState), so
// on a library-owned stream it is captured once into a hipGraph and replayed: one host call per
// solve instead of 4 launches per iteration.
int run_lm(vo_ba *h, int max_it) {
  // Measured on MI355X / ROCm 7.2: replaying a 30-45 node graph costs ~100-200 us of host time before
  // the first node starts, while eager launches (~4 us each) stay ahead of ~20 us kernels.  Graph
  // replay is therefore opt-in (vo_set_option(VO_OPT_BA_GRAPH, 1)), for hosts whose launch path is the bottleneck.
  const bool use_graph = g_opt_ba_graph.load(std::memory_order_relaxed) != 0;
  if (!use_graph || !h->own_stream || max_it < 1 || h->collectives) return run_lm_eager(h, max_it);
  auto it = h->graphs.find(max_it);
  if (it == h->graphs.end()) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return run_lm_eager(h, max_it);
    const int rc = run_lm_eager(h, max_it);
    if (hipStreamEndCapture(h->stream, &graph) != hipSuccess || rc != VO_OK || !graph) {
      (void)hipGetLastError();
      if (graph) (void)hipGraphDestroy(graph);
      return run_lm_eager(h, max_it);
    }
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      return run_lm_eager(h, max_it);
    }
    (void)hipGraphDestroy(graph);
    it = h->graphs.emplace(max_it, exec).first;
  }
  VO_HIP_CHECK(hipGraphLaunch(it->second, h->stream));
  return VO_OK;
}

int current_index(vo_ba *h, int *cur) {
  BaState s;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  VO_HIP_CHECK(hipMemcpy(&s, h->D.st, sizeof(s), hipMemcpyDeviceToHost));
  *cur = s.cur;
  return VO_OK;
}

}  // namespace

extern "C" {

// The kernel is a chain of ~20 dependent LM iterations whose fixed part (6 x 6 solve, exp / log, reductions) every
// wavefront of a workgroup repeats, and it needs all 256 registers (one wavefront per SIMD).  A single frame is
// fastest with four wavefronts sharing its observations; a batch is fastest with ONE wavefront per frame, so that a
// CU works on four frames at once instead of four times on one (1024 frames x 1000 observations: 1.45 -> see DESIGN).
static inline int pose_block_width(int n_problems) {
  const int forced = g_opt_pose_block.load(std::memory_order_relaxed);  // vo_set_option(VO_OPT_POSE_BLOCK, ...)
  if (forced == 64 || forced == 128 || forced == 256) return forced;
  return n_problems >= 512 ? 64 : 256;
}

int vo_pose_only_solve_dev(int n_problems, const int32_t *dev_offsets, int max_obs, const double *dev_points,
                           const double *dev_obs, const double *dev_inv_sigma, const double *dev_cam5,
                           double *dev_poses, uint8_t *dev_outlier, int32_t *dev_n_inliers,
                           vo_lm_summary *dev_summaries, void *hip_stream) {
  (void)max_obs;
  if (n_problems < 0 || (n_problems > 0 && (!dev_offsets || !dev_poses || !dev_outlier || !dev_n_inliers || !dev_cam5)))
    return VO_ERR_INVALID;
  if (n_problems == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  const int bw = pose_block_width(n_problems);
  hipLaunchKernelGGL(bw == 64 ? k_pose_only<true> : k_pose_only<false>, dim3(n_problems), dim3(bw), 0, (hipStream_t)hip_stream,
                     dev_offsets, dev_points, dev_obs, dev_inv_sigma, dev_cam5, dev_poses, dev_outlier, dev_n_inliers, dev_summaries, 0);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_pose_only_solve_ranges_dev(int n_problems, const int32_t *dev_ranges, const double *dev_points, const double *dev_obs,
                                  const double *dev_inv_sigma, const double *dev_cam5, double *dev_poses,
                                  uint8_t *dev_outlier, int32_t *dev_n_inliers, vo_lm_summary *dev_summaries,
                                  void *hip_stream) {
  if (n_problems < 0 || (n_problems > 0 && (!dev_ranges || !dev_poses || !dev_outlier || !dev_n_inliers || !dev_cam5)))
    return VO_ERR_INVALID;
  if (n_problems == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  const int bw = pose_block_width(n_problems);
  hipLaunchKernelGGL(bw == 64 ? k_pose_only<true> : k_pose_only<false>, dim3(n_problems), dim3(bw), 0, (hipStream_t)hip_stream,
                     dev_ranges, dev_points, dev_obs, dev_inv_sigma, dev_cam5, dev_poses, dev_outlier, dev_n_inliers, dev_summaries, 1);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_pose_only_solve(int n_problems, const int32_t *offsets, const double *points, const double *obs,
                       const double *inv_sigma, const double cam[5], double *poses, uint8_t *outlier,
                       int32_t *n_inliers, vo_lm_summary *summaries) {
  if (n_problems < 0 || (n_problems > 0 && (!offsets || !poses || !n_inliers || !cam))) return VO_ERR_INVALID;
  if (n_problems == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  const int total = offsets[n_problems];
  if (total > 0 && (!points || !obs || !inv_sigma || !outlier)) return VO_ERR_INVALID;
  // One staging block each way (a tracking thread calls this once or twice per frame: eleven small
  // copies cost more than the kernel's first LM iterations).  Per host thread, grow-only.
  auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
  const size_t o_pts = 0, o_obs = o_pts + (size_t)total * 24, o_is = o_obs + (size_t)total * 24,
               o_cam = o_is + (size_t)total * 8, o_pose = o_cam + 40, o_off = o_pose + (size_t)n_problems * 48,
               in_bytes = up8(o_off + (size_t)(n_problems + 1) * 4);
  const size_t r_pose = 0, r_sum = r_pose + (size_t)n_problems * 48,
               r_inl = r_sum + (size_t)n_problems * 2 * sizeof(vo_lm_summary), r_out = up8(r_inl + (size_t)n_problems * 4),
               out_bytes = up8(r_out + (size_t)std::max(total, 1));
  hipStream_t st = vo::thread_stream();  // the calling thread's own stream: never queues behind another thread's solve
  thread_local vo::PinnedBuf pinned;
  thread_local vo::ScratchBuf d_in, d_out;
  VO_CHECK(pinned.reserve(std::max(in_bytes, out_bytes)));
  uint8_t *stage = pinned.data();
  if (total > 0) {
    memcpy(&stage[o_pts], points, (size_t)total * 24);
    memcpy(&stage[o_obs], obs, (size_t)total * 24);
    memcpy(&stage[o_is], inv_sigma, (size_t)total * 8);
  }
  memcpy(&stage[o_cam], cam, 40);
  memcpy(&stage[o_pose], poses, (size_t)n_problems * 48);
  memcpy(&stage[o_off], offsets, (size_t)(n_problems + 1) * 4);
  VO_CHECK(d_in.reserve(in_bytes));
  VO_CHECK(d_out.reserve(out_bytes));
  VO_HIP_CHECK(hipMemcpyAsync(d_in.p, stage, in_bytes, hipMemcpyHostToDevice, st));
  uint8_t *di = d_in.as<uint8_t>(), *dout = d_out.as<uint8_t>();
  // the kernel updates poses in place: give it the output block's copy
  VO_HIP_CHECK(hipMemcpyAsync(dout + r_pose, di + o_pose, (size_t)n_problems * 48, hipMemcpyDeviceToDevice, st));
  VO_HIP_CHECK(hipMemsetAsync(dout + r_sum, 0, (size_t)n_problems * 2 * sizeof(vo_lm_summary), st));
  VO_CHECK(vo_pose_only_solve_dev(n_problems, reinterpret_cast<const int32_t *>(di + o_off), 0,
                                  reinterpret_cast<const double *>(di + o_pts), reinterpret_cast<const double *>(di + o_obs),
                                  reinterpret_cast<const double *>(di + o_is), reinterpret_cast<const double *>(di + o_cam),
                                  reinterpret_cast<double *>(dout + r_pose), dout + r_out,
                                  reinterpret_cast<int32_t *>(dout + r_inl),
                                  reinterpret_cast<vo_lm_summary *>(dout + r_sum), st));
  // the inputs have left the staging block once the kernel has run: it takes the results back
  if (hipMemcpyAsync(stage, dout, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    vo::set_error("pose-only kernel failed: %s", hipGetErrorString(hipGetLastError()));
    return VO_ERR_HIP;
  }
  memcpy(poses, &stage[r_pose], (size_t)n_problems * 48);
  if (total > 0) memcpy(outlier, &stage[r_out], total);
  memcpy(n_inliers, &stage[r_inl], (size_t)n_problems * 4);
  if (summaries) memcpy(summaries, &stage[r_sum], (size_t)n_problems * 2 * sizeof(vo_lm_summary));
  return VO_OK;
}

int vo_sim3_solve(int n_problems, const int32_t *offsets, const double *cam_match, const double *pix_curr,
                  const double *inv_sigma_curr, const double *cam_curr, const double *pix_match,
                  const double *inv_sigma_match, const double camera[4], int fix_scale, double *poses, double *scales,
                  uint8_t *outlier, int32_t *n_inliers, vo_lm_summary *summaries) {
  if (n_problems < 0 || (n_problems > 0 && (!offsets || !poses || !scales || !n_inliers || !camera))) return VO_ERR_INVALID;
  if (n_problems == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  const int total = offsets[n_problems];
  if (total > 0 && (!cam_match || !pix_curr || !inv_sigma_curr || !cam_curr || !pix_match || !inv_sigma_match || !outlier))
    return VO_ERR_INVALID;
  thread_local vo::ScratchBuf d_off, d_pm, d_pc, d_isc, d_Pc, d_pxm, d_ism, d_cam, d_pose, d_sc, d_out, d_inl, d_sum;
  int rc = VO_OK;
  auto fail = [&](int r) { return r; };
  hipStream_t st = vo::thread_stream();
  auto upload = [&](vo::DevBuf &b, const void *src, size_t bytes) { return vo::upload(b, src, bytes, st, "vo_sim3_solve"); };
  if ((rc = upload(d_off, offsets, (size_t)(n_problems + 1) * 4)) != VO_OK) return fail(rc);
  if ((rc = upload(d_pm, cam_match, (size_t)total * 24)) != VO_OK) return fail(rc);
  if ((rc = upload(d_pc, pix_curr, (size_t)total * 16)) != VO_OK) return fail(rc);
  if ((rc = upload(d_isc, inv_sigma_curr, (size_t)total * 8)) != VO_OK) return fail(rc);
  if ((rc = upload(d_Pc, cam_curr, (size_t)total * 24)) != VO_OK) return fail(rc);
  if ((rc = upload(d_pxm, pix_match, (size_t)total * 16)) != VO_OK) return fail(rc);
  if ((rc = upload(d_ism, inv_sigma_match, (size_t)total * 8)) != VO_OK) return fail(rc);
  if ((rc = upload(d_cam, camera, 32)) != VO_OK) return fail(rc);
  if ((rc = upload(d_pose, poses, (size_t)n_problems * 48)) != VO_OK) return fail(rc);
  if ((rc = upload(d_sc, scales, (size_t)n_problems * 8)) != VO_OK) return fail(rc);
  if ((rc = d_out.reserve(std::max(64, total))) != VO_OK) return fail(rc);
  if ((rc = d_inl.reserve((size_t)n_problems * 4)) != VO_OK) return fail(rc);
  if ((rc = d_sum.reserve((size_t)n_problems * 2 * sizeof(vo_lm_summary))) != VO_OK) return fail(rc);
  if (fix_scale)
    hipLaunchKernelGGL(k_sim3<6>, dim3(n_problems), dim3(256), 0, st, d_off.as<int>(), d_pm.as<double>(),
                       d_pc.as<double>(), d_isc.as<double>(), d_Pc.as<double>(), d_pxm.as<double>(), d_ism.as<double>(),
                       d_cam.as<double>(), d_pose.as<double>(), d_sc.as<double>(), d_out.as<uint8_t>(), d_inl.as<int>(),
                       d_sum.as<vo_lm_summary>());
  else
    hipLaunchKernelGGL(k_sim3<7>, dim3(n_problems), dim3(256), 0, st, d_off.as<int>(), d_pm.as<double>(),
                       d_pc.as<double>(), d_isc.as<double>(), d_Pc.as<double>(), d_pxm.as<double>(), d_ism.as<double>(),
                       d_cam.as<double>(), d_pose.as<double>(), d_sc.as<double>(), d_out.as<uint8_t>(), d_inl.as<int>(),
                       d_sum.as<vo_lm_summary>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(poses, d_pose.p, (size_t)n_problems * 48, st, "vo_sim3_solve"));
  VO_CHECK(vo::copy_d2h(scales, d_sc.p, (size_t)n_problems * 8, st, "vo_sim3_solve"));
  if (total > 0) VO_CHECK(vo::copy_d2h(outlier, d_out.p, total, st, "vo_sim3_solve"));
  VO_CHECK(vo::copy_d2h(n_inliers, d_inl.p, (size_t)n_problems * 4, st, "vo_sim3_solve"));
  if (summaries)
    VO_CHECK(vo::copy_d2h(summaries, d_sum.p, (size_t)n_problems * 2 * sizeof(vo_lm_summary), st, "vo_sim3_solve"));
  return vo::stream_sync(st, "vo_sim3_solve");
}

// the host part of a problem: validation, stable grouping of the edges by point (the order Ceres' Schur eliminator walks its
// chunks) by a counting sort, free-camera slots
static int ba_check_args(int n_cams, const double *poses, const uint8_t *cam_fixed, int n_points, const double *points, int n_edges,
                         const int32_t *edge_cam, const int32_t *edge_point, const double *edge_obs, const double *edge_inv_sigma,
                         const double *cam, const char *fn) {
  if (n_cams < 1 || n_points < 0 || n_edges < 0 || !poses || !cam_fixed || !cam || (n_points > 0 && !points) ||
      (n_edges > 0 && (!edge_cam || !edge_point || !edge_obs || !edge_inv_sigma))) {
    vo::set_error("%s: invalid argument", fn);
    return VO_ERR_INVALID;
  }
  return VO_OK;
}
// Two passes over the caller's edges: (1) range check + edges per point (into a scratch vector: a rejected problem leaves the
// handle as it was), (2) the stable scatter into point order -- straight into the page-locked upload block when the problem is
// LDS-sized (the offsets are the ones build_device's first four uploads compute) -- which also counts the edges per free camera.
static int ba_fill_problem(vo_ba *h, int n_cams, const double *poses, const uint8_t *cam_fixed, int n_points, const double *points,
                           int n_edges, const int32_t *edge_cam, const int32_t *edge_point, const double *edge_obs,
                           const double *edge_inv_sigma, const double cam[5], const char *fn) {
  std::vector<int> &cnt = h->fill_tmp;
  cnt.assign((size_t)n_points + 1, 0);
  {
    int *c1 = cnt.data() + 1;
    unsigned bad = 0;
    for (int e = 0; e < n_edges; e++) {
      const unsigned pc = (unsigned)edge_cam[e], pp = (unsigned)edge_point[e];
      if (pc >= (unsigned)n_cams || pp >= (unsigned)n_points) {
        bad = 1;
        vo::set_error("%s: edge %d references camera %d / point %d out of range", fn, e, edge_cam[e], edge_point[e]);
        break;
      }
      c1[pp]++;
    }
    if (bad) return VO_ERR_INVALID;
  }
  h->n_cams = n_cams, h->n_pts = n_points, h->n_edges = n_edges, h->nf = 0;
  h->poses.assign(poses, poses + 6 * (size_t)n_cams);
  h->points.assign(points, points + 3 * (size_t)n_points);
  h->cam_fixed.assign(cam_fixed, cam_fixed + n_cams);
  memcpy(h->cam, cam, sizeof(h->cam));
  h->cam_slot.assign(n_cams, -1);
  h->slot_cam.clear();
  for (int c = 0; c < n_cams; c++)
    if (!cam_fixed[c]) {
      h->cam_slot[c] = h->nf++;
      h->slot_cam.push_back(c);
    }
  h->perm.resize(n_edges);
  bool placed = false;
  h->pre_copied = 0;
  size_t edge_bytes = 0;
  if (6 * h->nf + 1 <= kMaxN && n_edges > 0 && h->stream && h->up_pin.reserve(up_pin_bytes(n_edges, n_points, n_cams)) == VO_OK &&
      reserve_grow(h->b_up_arena, h->up_pin.bytes) == VO_OK) {
    // the staging offsets of build_device's uploads of e_cam, e_pt, e_obs, e_is (in that order, from offset 0)
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o0 = 0, o1 = al(o0 + std::max<size_t>((size_t)n_edges * 4, 64)), o2 = al(o1 + std::max<size_t>((size_t)n_edges * 4, 64)),
                 o3 = al(o2 + std::max<size_t>((size_t)n_edges * 24, 64));
    if (o3 + (size_t)n_edges * 8 <= h->up_pin.bytes) {
      uint8_t *pb = h->up_pin.data();
      h->e_cam.bind(reinterpret_cast<int *>(pb + o0), n_edges), h->e_pt.bind(reinterpret_cast<int *>(pb + o1), n_edges);
      h->e_obs.bind(reinterpret_cast<double *>(pb + o2), 3 * (size_t)n_edges), h->e_is.bind(reinterpret_cast<double *>(pb + o3), n_edges);
      placed = true;
      edge_bytes = o3 + (size_t)n_edges * 8;
    }
  }
  if (!placed) h->e_cam.resize(n_edges), h->e_pt.resize(n_edges), h->e_obs.resize(3 * (size_t)n_edges), h->e_is.resize(n_edges);
  h->pt_start.resize((size_t)n_points + 1);
  {
    int run = 0;
    for (int j = 0; j < n_points; j++) {  // cnt[j + 1] = edges of point j -> cnt[j] = next free position of point j
      const int c = cnt[j + 1];
      h->pt_start[j] = run, cnt[j] = run, run += c;
    }
    h->pt_start[n_points] = run;
  }
  h->cam_count.assign((size_t)h->nf + 1, 0);
  int *ec = h->e_cam.data(), *ep = h->e_pt.data(), *pm = h->perm.data(), *fl = cnt.data(), *cc = h->cam_count.data();
  const int *slot = h->cam_slot.data();
  double *eo = h->e_obs.data(), *ei = h->e_is.data();
  for (int e = 0; e < n_edges; e++) {  // counting sort: stable, caller order within a point
    const int pp = edge_point[e], pc = edge_cam[e];
    const int sidx = fl[pp]++;
    pm[sidx] = e, ec[sidx] = pc, ep[sidx] = pp, ei[sidx] = edge_inv_sigma[e];
    eo[3 * (size_t)sidx] = edge_obs[3 * (size_t)e], eo[3 * (size_t)sidx + 1] = edge_obs[3 * (size_t)e + 1];
    eo[3 * (size_t)sidx + 2] = edge_obs[3 * (size_t)e + 2];
    const int sl = slot[pc];
    cc[sl < 0 ? h->nf : sl]++;
  }
  // The edge arrays (1.1 of the 1.4 MB a config-3 build uploads) start their trip now: the copy runs while build_device does
  // the rest of the host work (camera lists, small tables) instead of behind it.
  if (placed && edge_bytes > 0 &&
      hipMemcpyAsync(h->b_up_arena.p, h->up_pin.p, edge_bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess)
    h->pre_copied = edge_bytes;
  return VO_OK;
}

int vo_ba_create(vo_ba **out, int n_cams, const double *poses, const uint8_t *cam_fixed, int n_points,
                 const double *points, int n_edges, const int32_t *edge_cam, const int32_t *edge_point,
                 const double *edge_obs, const double *edge_inv_sigma, const double cam[5]) {
  if (!out) {
    vo::set_error("vo_ba_create: invalid argument");
    return VO_ERR_INVALID;
  }
  VO_CHECK(ba_check_args(n_cams, poses, cam_fixed, n_points, points, n_edges, edge_cam, edge_point, edge_obs, edge_inv_sigma, cam, "vo_ba_create"));
  VO_CHECK(vo::ensure_device());
  vo_ba *h = new vo_ba();
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    vo::set_error("hipStreamCreate failed");
    return VO_ERR_HIP;
  }
  h->own_stream = true;
  const int frc = ba_fill_problem(h, n_cams, poses, cam_fixed, n_points, points, n_edges, edge_cam, edge_point, edge_obs, edge_inv_sigma, cam, "vo_ba_create");
  if (frc != VO_OK) {
    vo_ba_destroy(h);
    return frc;
  }
  *out = h;
  return VO_OK;
}

// A NEW problem in an existing handle (localMapping.cpp:38 solves a different local window per key-frame): stream, device
// buffers (grow-only, with headroom), page-locked staging and options stay; nothing is allocated or freed unless the new
// problem is larger than every one before it.  Shard, all-reduce callback and options are kept.
int vo_ba_reset(vo_ba *h, int n_cams, const double *poses, const uint8_t *cam_fixed, int n_points, const double *points,
                int n_edges, const int32_t *edge_cam, const int32_t *edge_point, const double *edge_obs,
                const double *edge_inv_sigma, const double cam[5]) {
  if (!h) {
    vo::set_error("vo_ba_reset: null handle");
    return VO_ERR_INVALID;
  }
  VO_CHECK(ba_check_args(n_cams, poses, cam_fixed, n_points, points, n_edges, edge_cam, edge_point, edge_obs, edge_inv_sigma, cam, "vo_ba_reset"));
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));  // nothing of the previous problem is in flight
  drop_graphs(h);
  // (a problem that fails the range check leaves the handle's previous problem in place, still built)
  VO_CHECK(ba_fill_problem(h, n_cams, poses, cam_fixed, n_points, points, n_edges, edge_cam, edge_point, edge_obs, edge_inv_sigma, cam, "vo_ba_reset"));
  vo::chol_plan_destroy(h->chol_plan);
  h->chol_plan = nullptr;
  for (auto *&sp : h->seg_plan) vo::chol_plan_destroy(sp), sp = nullptr;
  h->built = false, h->state_cached = false;
  // caller-owned reduce buffers were sized for the previous problem (vo_ba_set_reduce_buffers): they must be set again
  h->ext_payload = nullptr, h->ext_payload2 = nullptr;
  h->D = BaDev{};
  h->lm_max_it = 0, h->archive_slot = -1, h->lba_second = false;
  h->n_pack_tiles = 0, h->pt_owner.clear(), h->collectives = false, h->seg_mode = false, h->seg_c0 = 0, h->n_seg_tiles = 0;
  h->order_parts = 1, h->order_cyclic = 0, h->order_sep = 0, h->order_depth = 0, h->order_tiles = 0;
  return VO_OK;
}

void vo_ba_destroy(vo_ba *h) {
  if (!h) return;
  (void)hipStreamSynchronize(h->stream);
  drop_graphs(h);
  for (vo::DevBuf *b : {&h->b_ecam, &h->b_ept, &h->b_eobs, &h->b_eis, &h->b_eact, &h->b_ptstart, &h->b_local,
                        &h->b_camslot, &h->b_slotcam, &h->b_camstart, &h->b_camedges, &h->b_ptin, &h->b_camin,
                        &h->b_xc0, &h->b_xc1, &h->b_xp0, &h->b_xp1, &h->b_pc0, &h->b_pc1, &h->b_sc, &h->b_sp, &h->b_hinv, &h->b_gl2,
                        &h->b_dl, &h->b_wt, &h->b_wt1, &h->b_hll0, &h->b_hll1, &h->b_spt1, &h->b_sgemm, &h->b_scam, &h->b_spt, &h->b_payload, &h->b_zc,
                        &h->b_sbs, &h->b_payload2, &h->b_state, &h->b_out, &h->b_dbg, &h->b_cnt, &h->b_we0, &h->b_we1,
                        &h->b_glsc0, &h->b_glsc1, &h->b_Sd, &h->b_scv, &h->b_ddv, &h->b_gppv, &h->b_cholfail,
                        &h->b_pairstart, &h->b_paircc, &h->b_paire, &h->b_merge, &h->b_packtiles, &h->b_pack, &h->b_ltiles,
                        &h->b_segtiles, &h->b_segpack, &h->b_up_arena, &h->b_zero_arena})
    b->release();
  vo::chol_plan_destroy(h->chol_plan);
  for (auto *sp : h->seg_plan) vo::chol_plan_destroy(sp);
  if (h->pin.p) (void)hipHostFree(h->pin.p);
  if (h->up_pin.p) (void)hipHostFree(h->up_pin.p);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int vo_ba_set_stream(vo_ba *h, void *s) {
  if (!h) return VO_ERR_INVALID;
  (void)hipStreamSynchronize(h->stream);
  drop_graphs(h);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  h->own_stream = false;
  h->stream = (hipStream_t)s;
  return VO_OK;
}

int vo_ba_set_shard(vo_ba *h, int shard, int n_shards) {
  if (!h || n_shards < 1 || shard < 0 || shard >= n_shards || h->built) {
    vo::set_error("vo_ba_set_shard: invalid argument or called after the first solve");
    return VO_ERR_INVALID;
  }
  h->shard = shard, h->n_shards = n_shards;
  return VO_OK;
}

int vo_ba_set_option(vo_ba *h, int option, int value) {
  if (!h || h->built) {
    vo::set_error("vo_ba_set_option: null handle or called after the first use of the handle");
    return VO_ERR_INVALID;
  }
  switch (option) {
    case VO_BA_OPT_SEGMENTS: h->opt_segments = value != 0; return VO_OK;
    case VO_BA_OPT_COLLECTIVES_AT_ONE_RANK: h->opt_collectives_1 = value != 0; return VO_OK;
    case VO_BA_OPT_ORDER_PARTS: h->opt_order_parts = value; return VO_OK;
    default: vo::set_error("vo_ba_set_option: unknown option %d", option); return VO_ERR_INVALID;
  }
}

int vo_set_option(int option, int value) {
  switch (option) {
    case VO_OPT_BA_GRAPH: g_opt_ba_graph.store(value != 0); return VO_OK;
    case VO_OPT_POSE_BLOCK:
      if (value != 0 && value != 64 && value != 128 && value != 256) {
        vo::set_error("vo_set_option(VO_OPT_POSE_BLOCK): 0 (automatic), 64, 128 or 256");
        return VO_ERR_INVALID;
      }
      g_opt_pose_block.store(value);
      return VO_OK;
    case VO_OPT_BA_PAIRS_KERNEL:
      if (value != 0 && value != 1) {
        vo::set_error("vo_set_option(VO_OPT_BA_PAIRS_KERNEL): 0 (blocks staged through LDS) or 1 (lane = couple, register loads)");
        return VO_ERR_INVALID;
      }
      g_opt_pairs_kernel.store(value);
      return VO_OK;
    case VO_OPT_HAMMING_KERNEL:
      if (value != 0 && value != 1) {
        vo::set_error("vo_set_option(VO_OPT_HAMMING_KERNEL): 0 (matrix cores) or 1 (VALU)");
        return VO_ERR_INVALID;
      }
      vo::set_hamming_kernel(value);
      return VO_OK;
    default: vo::set_error("vo_set_option: unknown option %d", option); return VO_ERR_INVALID;
  }
}

int vo_ba_n_free_cams(const vo_ba *h) { return h ? h->nf : 0; }

int vo_ba_debug_order(vo_ba *h, int out[8]) {
  if (!h || !out) return VO_ERR_INVALID;
  VO_CHECK(build_device(h));
  out[0] = h->order_parts, out[1] = h->order_cyclic, out[2] = h->order_sep, out[3] = h->order_depth, out[4] = h->order_tiles;
  out[5] = h->D.large ? h->D.ld / vo::kCholPanel : 0;
  out[6] = 0;
  out[7] = h->seg_mode ? h->seg_c0 : 0;  // per-rank segment factorisation: first separator tile column (0: replicated)
  vo::chol_plan_info(h->chol_plan, nullptr, nullptr, &out[6]);
  return VO_OK;
}

int vo_ba_set_state(vo_ba *h, const double *poses, const double *points) {
  if (!h) return VO_ERR_INVALID;
  VO_CHECK(build_device(h));
  h->state_cached = false;
  int cur;
  VO_CHECK(current_index(h, &cur));
  if (poses) VO_HIP_CHECK(hipMemcpy(h->D.Xc[cur], poses, (size_t)h->n_cams * 48, hipMemcpyHostToDevice));
  if (points && h->n_pts) VO_HIP_CHECK(hipMemcpy(h->D.Xp[cur], points, (size_t)h->n_pts * 24, hipMemcpyHostToDevice));
  return VO_OK;
}

// Sharded handle with an all-reduce callback: every shard holds the up-to-date coordinates of ITS points only
// (and classifies ITS edges only).  One more collective -- outside the LM loop -- leaves the full point array
// (and the full erase mask) on every rank: owners contribute their values, everybody else zeros.
static int merge_shards(vo_ba *h, int cur, uint8_t *erase_sorted /*n_edges, in/out, or NULL*/) {
  if (h->n_shards <= 1 || !h->allreduce || !h->built) return VO_OK;
  const size_t np3 = (size_t)h->n_pts * 3, n = np3 + (erase_sorted ? (size_t)h->n_edges : 0);
  if (n == 0) return VO_OK;
  std::vector<double> buf(n, 0.0);
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (np3) VO_HIP_CHECK(hipMemcpy(buf.data(), h->D.Xp[cur], np3 * 8, hipMemcpyDeviceToHost));
  for (int j = 0; j < h->n_pts; j++)
    if (h->pt_owner[j] != h->shard) buf[3 * j] = buf[3 * j + 1] = buf[3 * j + 2] = 0.0;
  if (erase_sorted)
    for (int e = 0; e < h->n_edges; e++)
      buf[np3 + e] = (h->pt_owner[h->e_pt[e]] == h->shard && erase_sorted[e]) ? 1.0 : 0.0;
  VO_CHECK(h->b_merge.reserve(n * 8));
  VO_HIP_CHECK(hipMemcpyAsync(h->b_merge.p, buf.data(), n * 8, hipMemcpyHostToDevice, h->stream));
  VO_CHECK(shard_allreduce(h, h->b_merge.as<double>(), n));
  VO_HIP_CHECK(hipMemcpyAsync(buf.data(), h->b_merge.p, n * 8, hipMemcpyDeviceToHost, h->stream));
  if (np3) VO_HIP_CHECK(hipMemcpyAsync(h->D.Xp[cur], h->b_merge.p, np3 * 8, hipMemcpyDeviceToDevice, h->stream));
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (erase_sorted)
    for (int e = 0; e < h->n_edges; e++) erase_sorted[e] = buf[np3 + e] != 0.0;
  return VO_OK;
}

int vo_ba_get_state(vo_ba *h, double *poses, double *points) {
  if (!h) return VO_ERR_INVALID;
  if (h->built && h->state_cached) {  // the state vo_ba_local_ba_finish brought back with its results
    if (poses) memcpy(poses, h->pin.data() + h->cache_xc_off, (size_t)h->n_cams * 48);
    if (points && h->n_pts) memcpy(points, h->pin.data() + h->cache_xp_off, (size_t)h->n_pts * 24);
    return VO_OK;
  }
  VO_CHECK(build_device(h));
  int cur;
  VO_CHECK(current_index(h, &cur));
  VO_CHECK(merge_shards(h, cur, nullptr));
  if (poses) VO_HIP_CHECK(hipMemcpy(poses, h->D.Xc[cur], (size_t)h->n_cams * 48, hipMemcpyDeviceToHost));
  if (points && h->n_pts) VO_HIP_CHECK(hipMemcpy(points, h->D.Xp[cur], (size_t)h->n_pts * 24, hipMemcpyDeviceToHost));
  return VO_OK;
}

int vo_ba_lm_begin(vo_ba *h, double hm, double hs, int max_it, const uint8_t *edge_active) {
  if (!h || max_it < 0) return VO_ERR_INVALID;
  return lm_begin(h, hm, hs, max_it, edge_active, false);
}
// (a handle in segment mode sums three things per iteration in its own order; the split-phase interface, whose caller sums the
//  whole reduced system between linearize and step, is for handles without a callback)
static int reject_segment_mode(const vo_ba *h, const char *fn) {
  if (h->seg_mode) {
    vo::set_error("%s: the handle runs the per-rank segment factorisation through its all-reduce callback; drive it with vo_ba_solve "
                  "(VO_BA_OPT_SEGMENTS selected this form)", fn);
    return VO_ERR_INVALID;
  }
  return VO_OK;
}
int vo_ba_linearize(vo_ba *h) {
  if (!h || !h->built) return VO_ERR_INVALID;
  VO_CHECK(reject_segment_mode(h, "vo_ba_linearize"));
  return launch_linearize(h);
}
int vo_ba_step(vo_ba *h) {
  if (!h || !h->built) return VO_ERR_INVALID;
  VO_CHECK(reject_segment_mode(h, "vo_ba_step"));
  return launch_step(h);
}
int vo_ba_update(vo_ba *h) { return h && h->built ? launch_update(h) : VO_ERR_INVALID; }
int vo_ba_lm_end(vo_ba *h, vo_lm_summary *s) { return h && h->built ? lm_end(h, s) : VO_ERR_INVALID; }

int vo_ba_set_allreduce(vo_ba *h, vo_allreduce_fn fn, void *user) {
  if (!h) return VO_ERR_INVALID;
  h->allreduce = fn, h->allreduce_user = user;
  return VO_OK;
}

int vo_ba_set_reduce_buffers(vo_ba *h, double *dev_system, double *dev_cost) {
  if (!h) return VO_ERR_INVALID;
  drop_graphs(h);
  h->ext_payload = dev_system, h->ext_payload2 = dev_cost;
  if (h->built) {
    h->D.payload = dev_system ? dev_system : h->b_payload.as<double>();
    h->D.payload2 = dev_cost ? dev_cost : h->b_payload2.as<double>();
    if (h->D.large) {
      h->D.Sd = dev_system ? dev_system : h->b_Sd.as<double>();  // the Cholesky storage is the payload
      // the caller's all-reduce sums the WHOLE storage, and the tiles outside the factorisation plan are never written:
      // they must be zeros, not whatever the caller's allocation held
      if (dev_system) VO_HIP_CHECK(hipMemsetAsync(dev_system, 0, (size_t)(h->D.ld + vo::kCholPanel) * h->D.ld * 8, h->stream));
    }
  }
  return VO_OK;
}

int vo_ba_reduced_system(vo_ba *h, double **p, size_t *n) {
  if (!h || !p || !n) return VO_ERR_INVALID;
  VO_CHECK(build_device(h));
  if (h->D.large) {  // matrix, right-hand side row, solution row, extras: the whole Cholesky storage
    *p = h->D.Sd;
    *n = (size_t)(h->D.ld + vo::kCholPanel) * h->D.ld;
    return VO_OK;
  }
  *p = h->D.payload;
  *n = (size_t)h->D.Mpad * h->D.Mpad + (size_t)h->nf * 27 + 1 + h->n_shards;
  return VO_OK;
}
int vo_ba_reduced_cost(vo_ba *h, double **p, size_t *n) {
  if (!h || !p || !n) return VO_ERR_INVALID;
  VO_CHECK(build_device(h));
  *p = h->D.payload2;
  *n = 6;
  return VO_OK;
}

// A sharded handle (vo_ba_set_shard, n_shards > 1) holds partial sums only: without the two all-reduces per LM
// iteration the camera step would silently be solved from one shard's partial system.
static int reject_unreduced_shards(const vo_ba *h, const char *fn) {
  if (h->n_shards > 1 && !h->allreduce) {
    vo::set_error("%s: handle is shard %d of %d and has no all-reduce callback (vo_ba_set_allreduce); drive it through "
                  "vo_ba_lm_begin / linearize / step / update with the caller's all-reduce in between", fn, h->shard, h->n_shards);
    return VO_ERR_INVALID;
  }
  return VO_OK;
}

int vo_ba_solve(vo_ba *h, double hm, double hs, int max_it, const uint8_t *edge_active, vo_lm_summary *sum) {
  if (!h || max_it < 0) return VO_ERR_INVALID;
  VO_CHECK(reject_unreduced_shards(h, "vo_ba_solve"));
  VO_CHECK(lm_begin(h, hm, hs, max_it, edge_active, false));
  VO_CHECK(run_lm(h, max_it));
  return lm_end(h, sum);
}

static void summary_from_state(const BaState &s, vo_lm_summary *sum) {
  sum->iterations = s.iter;
  sum->accepted = s.accepted;
  sum->termination = s.termination;
  sum->reserved = 0;
  sum->initial_cost = s.initial_cost;
  sum->final_cost = s.x_cost;
  sum->final_radius = s.radius;
}

int vo_ba_local_ba_enqueue(vo_ba *h, const volatile unsigned char *stop) {
  if (!h) return VO_ERR_INVALID;
  VO_CHECK(reject_unreduced_shards(h, "vo_ba_local_ba"));
  h->lba_second = false;
  if (stop && *stop) return VO_ERR_STOPPED;  // :594-595 (no write-back, Q-B8)
  // The whole schedule is queued without a host synchronisation in between; the stop flag is
  // polled where the reference polls it (:612), which here is at enqueue time of problem 2.
  h->archive_slot = -1;
  VO_CHECK(lm_begin(h, (double)sqrtf(5.991f), (double)sqrtf(7.815f), 5, nullptr, false));
  VO_CHECK(run_lm(h, 5));
  uint8_t *out = h->b_out.as<uint8_t>();
  const dim3 eg((std::max(1, h->n_edges) + 255) / 256);
  if (!(stop && *stop)) {  // :612
    h->archive_slot = 0;  // problem 1's final state is archived by problem 2's setup kernel, which also
                          // classifies the edges at problem 1's solution (:618-689) before marking
    VO_CHECK(lm_begin(h, 0.0, 0.0, 10, nullptr, true, out));
    h->archive_slot = -1;
    VO_CHECK(run_lm(h, 10));
    h->lba_second = true;
  }
  if (!h->lba_second) hipLaunchKernelGGL(k_ba_clear, eg, dim3(256), 0, h->stream, h->D, 0, out);  // no outliers known yet
  hipLaunchKernelGGL(k_ba_classify, eg, dim3(256), 0, h->stream, h->D, 1, out);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_ba_local_ba_finish(vo_ba *h, uint8_t *edge_erase, vo_lm_summary *sums) {
  if (!h || !edge_erase || !h->built) return VO_ERR_INVALID;
  // both results land in one page-locked block: a copy into pageable memory costs ~20 us apiece
  const size_t st_off = ((size_t)std::max(1, h->n_edges) + 15) & ~(size_t)15;
  // ... and, for an unsharded handle, both ping-pong copies of the poses and points ride along (a few hundred KB): the caller's
  // vo_ba_get_state then needs no round trip of its own (which of the two is current is known only after the wait)
  const bool cache = h->n_shards <= 1 && !h->D.large;
  const size_t xc_bytes = (size_t)h->n_cams * 48, xp_bytes = (size_t)h->n_pts * 24;
  const size_t xc_off = (st_off + 3 * sizeof(BaState) + 63) & ~(size_t)63, xp_off = (xc_off + 2 * xc_bytes + 63) & ~(size_t)63;
  VO_CHECK(h->pin.reserve(cache ? xp_off + 2 * xp_bytes + 64 : st_off + 3 * sizeof(BaState)));
  uint8_t *tmp = h->pin.data();
  BaState *st = reinterpret_cast<BaState *>(tmp + st_off);
  VO_HIP_CHECK(hipMemcpyAsync(tmp, h->b_out.p, h->n_edges, hipMemcpyDeviceToHost, h->stream));
  VO_HIP_CHECK(hipMemcpyAsync(st, h->D.st, 3 * sizeof(BaState), hipMemcpyDeviceToHost, h->stream));
  if (cache)
    for (int k = 0; k < 2; k++) {
      VO_HIP_CHECK(hipMemcpyAsync(tmp + xc_off + k * xc_bytes, h->D.Xc[k], xc_bytes, hipMemcpyDeviceToHost, h->stream));
      if (xp_bytes) VO_HIP_CHECK(hipMemcpyAsync(tmp + xp_off + k * xp_bytes, h->D.Xp[k], xp_bytes, hipMemcpyDeviceToHost, h->stream));
    }
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  h->state_cached = cache;
  h->cache_xc_off = xc_off + (size_t)st[0].cur * xc_bytes, h->cache_xp_off = xp_off + (size_t)st[0].cur * xp_bytes;
  VO_CHECK(merge_shards(h, st[0].cur, tmp));  // no-op for an unsharded handle
  for (int s = 0; s < h->n_edges; s++) edge_erase[h->perm[s]] = tmp[s];
  if (sums) {
    if (h->lba_second) {
      summary_from_state(st[1], &sums[0]);
      summary_from_state(st[0], &sums[1]);
    } else {
      summary_from_state(st[0], &sums[0]);
      memset(&sums[1], 0, sizeof(vo_lm_summary));
    }
  }
  return VO_OK;
}

int vo_ba_local_ba(vo_ba *h, const volatile unsigned char *stop, uint8_t *edge_erase, vo_lm_summary *sums) {
  if (!h || !edge_erase) return VO_ERR_INVALID;
  for (int e = 0; e < h->n_edges; e++) edge_erase[e] = 0;
  const int rc = vo_ba_local_ba_enqueue(h, stop);
  if (rc != VO_OK) return rc;
  return vo_ba_local_ba_finish(h, edge_erase, sums);
}

int vo_ba_classify(vo_ba *h, int final_pass) {
  if (!h) return VO_ERR_INVALID;
  VO_CHECK(build_device(h));
  uint8_t *out = h->b_out.as<uint8_t>();
  hipLaunchKernelGGL(k_ba_classify, dim3((std::max(1, h->n_edges) + 255) / 256), dim3(256), 0, h->stream, h->D,
                     final_pass ? 1 : 0, out);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_ba_lm_begin_inliers(vo_ba *h, double hm, double hs, int max_it) {
  if (!h || max_it < 0) return VO_ERR_INVALID;
  return lm_begin(h, hm, hs, max_it, nullptr, true);
}

int vo_ba_get_edge_outliers(vo_ba *h, uint8_t *edge_erase) {
  if (!h || !edge_erase) return VO_ERR_INVALID;
  VO_CHECK(build_device(h));
  std::vector<uint8_t> tmp(std::max(1, h->n_edges));
  VO_HIP_CHECK(hipMemcpyAsync(tmp.data(), h->b_out.p, h->n_edges, hipMemcpyDeviceToHost, h->stream));
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int s = 0; s < h->n_edges; s++) edge_erase[h->perm[s]] = tmp[s];
  return VO_OK;
}

int vo_ba_debug_stamps(vo_ba *h, unsigned long long *out /*64*/) {
  if (!h || !h->built) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  VO_HIP_CHECK(hipMemcpy(out, h->b_dbg.p, 512, hipMemcpyDeviceToHost));
  return VO_OK;
}

int vo_ba_debug_schur(vo_ba *h, double hm, double hs, double point_damping, const uint8_t *edge_active, double *S,
                      double *b, double *cost) {
  // undamped reduced system: run the linearisation kernels with a huge radius (D -> clamp/1e300 ~ 0)
  // and no Jacobi scaling effect removed on the host.
  if (!h || !S || !b || !cost) return VO_ERR_INVALID;
  VO_CHECK(build_device(h));
  if (h->D.large) {
    vo::set_error("vo_ba_debug_schur: only for the LDS-sized reduced systems");
    return VO_ERR_CAPACITY;
  }
  (void)point_damping;
  VO_CHECK(lm_begin(h, hm, hs, 1, edge_active, false));
  BaState s;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  VO_HIP_CHECK(hipMemcpy(&s, h->D.st, sizeof(s), hipMemcpyDeviceToHost));
  s.radius = 1e300;
  VO_HIP_CHECK(hipMemcpy(h->D.st, &s, sizeof(s), hipMemcpyHostToDevice));
  const int fused_saved = h->D.fused;
  h->D.fused = 0;  // materialise the complete payload with k_ba_reduce
  const int lrc = launch_linearize(h);
  h->D.fused = fused_saved;
  VO_CHECK(lrc);
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  const int n = 6 * h->nf, M = h->D.Mpad;
  std::vector<double> pay((size_t)M * M + (size_t)h->nf * 27 + 1 + h->n_shards);
  VO_HIP_CHECK(hipMemcpy(pay.data(), h->D.payload, pay.size() * 8, hipMemcpyDeviceToHost));
  const double *HP = pay.data() + (size_t)M * M;
  for (int r = 0; r < n; r++) {
    for (int c = 0; c < n; c++) {
      double v = -pay[(size_t)std::min(r, c) * M + std::max(r, c)];  // only upper tiles are produced
      if (r / 6 == c / 6) {
        const int slot = r / 6, a = std::min(r % 6, c % 6), bb = std::max(r % 6, c % 6);
        int t = 0;
        for (int q = 0; q < a; q++) t += 6 - q;
        v += HP[slot * 27 + t + (bb - a)];
      }
      S[(size_t)r * n + c] = v;
    }
    b[r] = HP[(r / 6) * 27 + 21 + r % 6] - pay[(size_t)r * M + n];
  }
  *cost = pay[(size_t)M * M + (size_t)h->nf * 27];
  return VO_OK;
}

int vo_se3_exp(const double xi[6], double R[9], double t[3]) {
  if (!xi || !R || !t) return VO_ERR_INVALID;
  const Se3 T = se3_exp(xi);
  const double *q = T.q;
  const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
  const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
  t[0] = T.t[0], t[1] = T.t[1], t[2] = T.t[2];
  return VO_OK;
}

int vo_se3_log(const double R[9], const double t[3], double xi[6]) {
  if (!xi || !R || !t) return VO_ERR_INVALID;
  // rotation matrix -> unit quaternion (Eigen's Quaternion(Matrix3) branch structure)
  Se3 T;
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    T.q[0] = 0.5 * s;
    s = 0.5 / s;
    T.q[1] = (R[7] - R[5]) * s, T.q[2] = (R[2] - R[6]) * s, T.q[3] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    double q[4];
    q[1 + i] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * s;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    for (int a = 0; a < 4; a++) T.q[a] = q[a];
  }
  quat_normalize(T.q);
  T.t[0] = t[0], T.t[1] = t[1], T.t[2] = t[2];
  se3_log(T, xi);
  return VO_OK;
}

}  // extern "C"
