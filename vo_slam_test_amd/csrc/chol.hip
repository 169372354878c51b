// chol.hip -- dense SPD factor + solve for the reduced camera system of a global bundle adjustment and for the
// loop-closure pose graph (6 N unknowns, N up to ~680 key-frames) on gfx950, FP64.
//
// What the reference asks of its solver: Ceres DENSE_SCHUR's Cholesky of the reduced camera matrix
// (src/optimizer_ceres.cpp:248, :600, :695) and SPARSE_NORMAL_CHOLESKY of the pose graph (:1252-1258).
//
// ONE launch, persistent workgroups, tile dataflow.  The matrix is cut into 64 x 64 tiles; tile (i, j) -- and the
// tile row that carries the right-hand side, which rides through the factorisation and comes out as L^-1 b -- is a
// TASK: its owner keeps the tile in MFMA accumulators, subtracts L(i,k) L(j,k)^T for k < j as those tiles become
// available (v_mfma_f64_16x16x4_f64 from LDS-staged operands), then finishes it (64 x 64 Cholesky on the diagonal,
// X L(j,j)^T = T below it) and publishes it.  Tasks are handed out by a ticket counter in column-major order, so
// everything a task waits for has an earlier ticket and is either finished or running: no deadlock, whatever the
// number of resident workgroups.  The trailing matrix is never re-read and re-written panel after panel (the round-1
// kernels did 47 x 4 dependent launches of 16-25 us each: 3.8 ms of launch latency for 9 GFLOP); here every tile is
// read once and written once, and the only serial chain left is diag(j) -> L(j+1, j) -> diag(j+1).
// The backward substitution L^T x = y runs in the same launch: one workgroup walks the diagonal from the bottom and
// takes the three nearest tiles of every column itself, the other workgroups deliver the far partial products.
//
// Hand-off between workgroups (MI355X: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed):
// every handed-off double is stored and loaded with agent-scope relaxed atomics (sc1: write-through / L1 bypass),
// the producer drains its stores (s_waitcnt vmcnt(0)), barriers, and one lane raises the tile's flag; consumers
// poll the flag with agent-scope loads.  Every poll loop is bounded: on expiry the kernel raises `fail` and all
// workgroups leave (a hang would cost the GPU box).
#include "vo_common.h"

#include <algorithm>
#include <mutex>
#include <vector>

namespace {

using namespace vo;

constexpr int NB = vo::kCholPanel;  // 64
#ifndef VO_CHOL_NEWTON
#define VO_CHOL_NEWTON 2
#endif
constexpr int kNewton = VO_CHOL_NEWTON;  // v_rsq_f64 is good to ~2^-26: one step gives ~1e-15, two steps full precision
constexpr int LP = NB + 1;          // LDS pitch of a staged tile (conflict-free column and row walks)
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int ld_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ double bcast_lane(double v, int src_lane) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src_lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src_lane);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

#ifdef VO_CHOL_STAMPS  // developer build: s_memrealtime stamps (100 MHz) of the critical tasks -> workspace tail
#define CSTAMP(slot, k) if (threadIdx.x == 0) C.stamps[(slot) * 16 + (k)] = wall_clock64()
#else
#define CSTAMP(slot, k)
#endif

struct CholCtx {
  double *A;
  int ld, m;          // m = ld / 64 tile columns; tile row m = the right-hand-side rows
  int *fail, *ticket, *done, *ready, *xready, *pcount, *invready;
  double *partial;    // [m][m][64] far partial products of the backward substitution
  double *linv;       // [m][64][64] inverses of the diagonal tiles (backward substitution by products, not by 64 pivots)
  int spin_limit;
  unsigned long long *stamps;  // [2 m][16] (VO_CHOL_STAMPS builds)
  // The plan (vo::chol_plan_create): which 64 x 64 tiles of L exist -- the structure of a reduced camera system is its
  // covisibility graph, and under a nested-dissection order of the key-frames whole tile columns are independent of
  // each other -- and the task list in ticket order.  A dense plan lists every tile.
  const int4 *tasks;            // (type, i, j, aux): 0 tile (i, j) of the factorisation (i == m: right-hand-side row),
                                // 1 inverse of diagonal tile j, 2 the backward chain, 3 far link (i, j), aux = its rank
  const unsigned long long *rowmask;  // [m + 1]: bit k of rowmask[i] = tile (i, k) of L exists (row m: the rhs row, all ones)
  const int2 *colinfo;          // [m]: (far links of column j, tile (j + 1, j) exists)
  int n_tasks, n_factor;        // tasks in all, factorisation tasks (what `done` counts)
};

// workgroup-wide wait for a flag (bounded).  Returns false when the kernel is being abandoned.
__device__ __forceinline__ bool wait_flag(const CholCtx &C, const int *flag, int want, int *s_state) {
  if (threadIdx.x == 0) {
    int ok = 1;
    for (int n = 0; ld_flag(flag) < want; n++) {
      if (n > C.spin_limit || ld_flag(C.fail) != 0) {
        if (n > C.spin_limit) atomicMax(C.fail, 2);  // dependency never arrived: give up loudly instead of hanging
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    *s_state = ok;
  }
  __syncthreads();
  const bool ok = *s_state != 0;
  __syncthreads();
  return ok;
}

// Tiles travel as 16-byte buffer loads / stores with the sc1 bit (agent scope: write-through on the store side,
// L1 bypass on the load side) -- the 8-byte agent-scope atomics the language offers move a 32 KB tile in 4 us, these
// in about a third of that.  A thread owns two adjacent doubles of rows r, r + 8, ...
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 1 << 4;  // cache-policy bit of the raw buffer intrinsics on gfx94x / gfx950
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const CholCtx &C) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)C.A, 0, (int)(((long long)(C.ld + NB) * C.ld * 8) & 0x7fffffff), 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t linv_rsrc(const CholCtx &C) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)C.linv, 0, (int)((long long)C.m * NB * NB * 8), 0x00020000);
}
__device__ __forceinline__ void load_tile_rs(const __amdgpu_buffer_rsrc_t rs, int pitch, int R0, int C0, double (*T)[LP]) {
  const int tid = threadIdx.x, c = 2 * (tid & 31), rr = tid >> 5;
  u32x4 v[8];
#pragma unroll
  for (int q = 0; q < 8; q++)
    v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long long)(R0 + rr + 8 * q) * pitch + C0 + c) * 8), 0, kSc1);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    T[rr + 8 * q][c] = __longlong_as_double(((unsigned long long)v[q].y << 32) | v[q].x);
    T[rr + 8 * q][c + 1] = __longlong_as_double(((unsigned long long)v[q].w << 32) | v[q].z);
  }
}
__device__ __forceinline__ void load_tile(const CholCtx &C, int R0, int C0, double (*T)[LP], bool /*coherent*/) {
  const int tid = threadIdx.x, c = 2 * (tid & 31), rr = tid >> 5;
  const __amdgpu_buffer_rsrc_t rs = tile_rsrc(C);
  u32x4 v[8];
#pragma unroll
  for (int q = 0; q < 8; q++)
    v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long long)(R0 + rr + 8 * q) * C.ld + C0 + c) * 8), 0, kSc1);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    T[rr + 8 * q][c] = __longlong_as_double(((unsigned long long)v[q].y << 32) | v[q].x);
    T[rr + 8 * q][c + 1] = __longlong_as_double(((unsigned long long)v[q].w << 32) | v[q].z);
  }
}
__device__ __forceinline__ void store_tile(const CholCtx &C, int R0, int C0, const double (*T)[LP], bool lower_only) {
  const int tid = threadIdx.x, c = 2 * (tid & 31), rr = tid >> 5;
  const __amdgpu_buffer_rsrc_t rs = tile_rsrc(C);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int r = rr + 8 * q;
    if (lower_only && c > r) continue;  // (the pair (r, c), (r, c + 1) with c == r also carries one entry above the diagonal: a zero)
    const unsigned long long lo = __double_as_longlong(T[r][c]), hi = __double_as_longlong(T[r][c + 1]);
    const u32x4 v = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(((long long)(R0 + r) * C.ld + C0 + c) * 8), 0, kSc1);
  }
}

// 64 x 64 Cholesky of the LDS tile T (lower triangle in, factor out), whole workgroup, four panels of 16 columns:
//  (1) the first wavefront factors the panel -- lane = row, the row's 16 panel entries in registers, the pivot row's
//      entries broadcast with v_readlane (no LDS round trip on the pivot chain, which is the critical path of the
//      whole factorisation); the reciprocal square root is a single-precision estimate refined by three Newton
//      steps in double (full precision at a fraction of v_sqrt_f64 + v_div_f64's latency);
//  (2) all four wavefronts apply the rank-16 update to the trailing tiles on the matrix cores.
// (A lane = row version with the whole 64-entry row in registers needs > 512 registers once it is part of this
// kernel: 7 KB of scratch per lane and 15 ms per factorisation.)
template <class PanelDone, class UpdateDone>
__device__ __forceinline__ bool tile_chol(double (*T)[LP], double *rdiag /*[NB]*/, double *colbuf /*[NB]*/, PanelDone &&panel_done,
                                          UpdateDone &&update_done) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q4 = lane >> 4;
  bool bad = false;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int c0 = 16 * b;
    if (wave == 0) {
      double p[16];
#pragma unroll
      for (int c = 0; c < 16; c++) p[c] = T[lane][c0 + c];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const double d = bcast_lane(p[j], c0 + j);
        if (!(d > 0.0) || !(d < 1e300)) bad = true;  // uniform
        // 1 / sqrt(d): hardware estimate + Newton steps (e = 1 - d r^2, r += r e / 2): the pivot chain is the critical
        // path of the whole factorisation, every dependent FP64 operation on it costs ~47 x 64 x 18 cycles
        double r = __builtin_amdgcn_rsq(d);
#pragma unroll
        for (int nr = 0; nr < kNewton; nr++) r = __builtin_fma(0.5 * r, __builtin_fma(-(d * r), r, 1.0), r);
        const double a = lane == c0 + j ? d * r : p[j] * r;  // L[lane][c0 + j] (meaningful for lane >= c0 + j)
        p[j] = a;
        if (lane == c0 + j) rdiag[c0 + j] = r;
        if (j < 15) {
          // the next pivot only needs column j + 1 of this update: its factor comes by v_readlane (no LDS round trip
          // on the pivot chain); the other columns take theirs from an LDS broadcast, off the chain
          p[j + 1] -= a * bcast_lane(a, c0 + j + 1);
          colbuf[lane] = a;
#pragma unroll
          for (int c = j + 2; c < 16; c++) p[c] -= a * colbuf[c0 + c];  // one wavefront issues its LDS operations in order
        }
      }
#pragma unroll
      for (int c = 0; c < 16; c++) T[lane][c0 + c] = (lane >= c0 + c) ? p[c] : 0.0;
    }
    __syncthreads();
    panel_done(b);  // columns c0 .. c0 + 15 are final: the owner ships them while the trailing update runs
    // trailing tiles (rt >= ct > b): T -= P P^T with P = the panel columns just written
    const int nt = (3 - b) * (4 - b) / 2;
    for (int t = wave; t < nt; t += 4) {
      int rt = b + 1, u = t;
      while (u > rt - (b + 1)) u -= rt - b, rt++;
      const int ct = b + 1 + u;
      const int R = 16 * rt, Cc = 16 * ct;
      double4_t acc = {T[R + q4][Cc + i16], T[R + q4 + 4][Cc + i16], T[R + q4 + 8][Cc + i16], T[R + q4 + 12][Cc + i16]};
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[R + i16][c0 + 4 * s2 + q4], T[Cc + i16][c0 + 4 * s2 + q4], acc, 0, 0, 0);
      T[R + q4][Cc + i16] = acc[0], T[R + q4 + 4][Cc + i16] = acc[1], T[R + q4 + 8][Cc + i16] = acc[2], T[R + q4 + 12][Cc + i16] = acc[3];
    }
    update_done(b);  // (contains the workgroup barrier that closes the panel step)
  }
  return !__syncthreads_or(bad ? 1 : 0);
}

// X L^T = W for the 64 rows of W (in place), L = lower-triangular LDS tile, rdiag = 1 / diag(L).  Rows are
// independent: every wavefront owns 16 of them and needs no workgroup barrier.  Column blocks of 16: the part of a
// block that depends on earlier blocks is an MFMA product, the 16 x 16 triangle a per-row substitution.
__device__ __forceinline__ void tile_trsm_block(double (*W)[LP], const double (*L)[LP], const double *rdiag, int b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, q4 = lane >> 4;
  const int r0 = 16 * wave, c0 = 16 * b;
  if (b > 0) {
    double4_t acc = {W[r0 + q4][c0 + i16], W[r0 + q4 + 4][c0 + i16], W[r0 + q4 + 8][c0 + i16], W[r0 + q4 + 12][c0 + i16]};
    for (int s = 0; s < 4 * b; s++)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[r0 + i16][4 * s + q4], L[c0 + i16][4 * s + q4], acc, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    W[r0 + q4][c0 + i16] = acc[0], W[r0 + q4 + 4][c0 + i16] = acc[1], W[r0 + q4 + 8][c0 + i16] = acc[2], W[r0 + q4 + 12][c0 + i16] = acc[3];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 16) {
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = W[r0 + lane][c0 + c];
#pragma unroll
    for (int c = 0; c < 16; c++) {
      double v = x[c];
#pragma unroll
      for (int q = 0; q < c; q++) v -= x[q] * L[c0 + c][c0 + q];
      x[c] = v * rdiag[c0 + c];
    }
#pragma unroll
    for (int c = 0; c < 16; c++) W[r0 + lane][c0 + c] = x[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void k_chol_tiles(CholCtx C) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double(*Pr)[LP] = reinterpret_cast<double(*)[LP]>(lds);
  double(*Pc)[LP] = reinterpret_cast<double(*)[LP]>(lds + NB * LP);
  double *col = lds + 2 * NB * LP, *rdiag = col + NB, *xv = rdiag + NB;  // xv[4][NB]: backward-substitution vectors
  __shared__ int s_ticket, s_state;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q4 = lane >> 4;
  const int m = C.m, ld = C.ld;
  const int nF = C.n_factor;
  for (;;) {
    if (tid == 0) s_ticket = atomicAdd(C.ticket, 1);
    __syncthreads();
    const int t = s_ticket;
    __syncthreads();
    if (t >= C.n_tasks) return;
    if (tid == 0) s_state = ld_flag(C.fail);
    __syncthreads();
    if (s_state != 0) return;  // abandoned (not positive definite, or a dependency timed out)
    __syncthreads();
    // Ticket order (the plan's): column by column over the tiles that exist (i = j .. m), and behind column j's tasks the
    // inverse of diagonal tile j - 2, which is complete by then (a task that polls for a long time costs the chain
    // memory bandwidth on the flag lines); the last two inverses follow the last column; then the backward chain and its
    // far links.  Everything a task waits for has an earlier ticket.
    const int4 task = C.tasks[t];
    const int inv_j = task.x == 1 ? task.z : -1;
    if (task.x == 0) {
      // ---------------------------------------------------------------- factorisation task (i, j)
      const int j = task.z;
      const int i = task.y;                                  // j <= i <= m  (i == m: right-hand-side rows)
      const int R0 = i < m ? NB * i : ld, C0 = NB * j;
      const int qr = (wave >> 1) * 32, qc = (wave & 1) * 32;
      double4_t acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const double *p = C.A + (long long)(R0 + qr + 16 * a + q4) * ld + C0 + qc + 16 * b + i16;
          acc[a][b] = double4_t{p[0], p[4LL * ld], p[8LL * ld], p[12LL * ld]};
        }
      bool alive = true;
      const int stamp_slot = i == j ? 2 * j : (i == j + 1 ? 2 * j + 1 : -1);
      // only the tile columns k in which both L(i, k) and L(j, k) exist contribute (and only those tiles are ever
      // published: a tile that does not exist must not be waited for)
      unsigned long long kmask = C.rowmask[i] & C.rowmask[j] & ((1ull << j) - 1ull);
      while (kmask != 0ull && alive) {
        const int k = (int)__builtin_ctzll(kmask);
        kmask &= kmask - 1ull;
        if (stamp_slot >= 0 && k == j - 1) CSTAMP(stamp_slot, 0);
        alive = wait_flag(C, C.ready + i * m + k, 1, &s_state);
        if (alive && i != j) alive = wait_flag(C, C.ready + j * m + k, 1, &s_state);
        if (!alive) break;
        if (stamp_slot >= 0 && k == j - 1) CSTAMP(stamp_slot, 1);
        load_tile(C, R0, NB * k, Pr, true);
        if (i != j) load_tile(C, NB * j, NB * k, Pc, true);
        __syncthreads();
        double(*Pb)[LP] = i != j ? Pc : Pr;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++) {
            const int r0 = qr + 16 * a, c0 = qc + 16 * b;
#pragma unroll
            for (int s = 0; s < NB / 4; s++)
              acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pr[r0 + i16][4 * s + q4], Pb[c0 + i16][4 * s + q4], acc[a][b], 0, 0, 0);
          }
        __syncthreads();
      }
      if (!alive) return;
      if (stamp_slot >= 0) CSTAMP(stamp_slot, 2);
      // accumulators -> LDS working tile
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int g = 0; g < 4; g++) Pr[qr + 16 * a + q4 + 4 * g][qc + 16 * b + i16] = acc[a][b][g];
      __syncthreads();
      if (i == j) {
        // The diagonal tile is published PANEL BY PANEL (flag = panels shipped): the tile below it starts its
        // triangular solve on the first 16 columns while the pivots of the next panel are still being computed.
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(C);
        double *rd_out = C.partial + (long long)m * m * NB + NB * j;
        const bool ok = tile_chol(
            Pr, rdiag, col,
            [&](int b) {  // panel b: rows 16 b .. 63 x 16 columns = 2 doubles per thread and row group
              const int c = 16 * b + 2 * (tid & 7), rr = tid >> 3;  // 8 threads per row, 32 rows per pass
#pragma unroll
              for (int q = 0; q < 2; q++) {
                const int r = rr + 32 * q;
                if (r < 16 * b) continue;
                const unsigned long long lo = __double_as_longlong(Pr[r][c]), hi = __double_as_longlong(Pr[r][c + 1]);
                const u32x4 v = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(((long long)(R0 + r) * ld + C0 + c) * 8), 0, kSc1);
              }
              if (tid < 16) st_sc1(rd_out + 16 * b + tid, rdiag[16 * b + tid]);
            },
            [&](int b) {  // (flagging a panel one pivot phase later, when its stores have long landed, was measured: the
                          // tile below falls behind by as much and the column period grows from 22.9 to 25.4 us)
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __syncthreads();
              if (tid == 0) st_flag(C.ready + i * m + j, b + 1);
            });
        if (!ok && tid == 0) atomicMax(C.fail, 1);  // not positive definite
        if (stamp_slot >= 0) CSTAMP(stamp_slot, 3);
        if (stamp_slot >= 0) CSTAMP(stamp_slot, 6);
        if (tid == 0) atomicAdd(C.done, 1);
        continue;
      } else {
        // X L(j,j)^T = T, panel by panel as the diagonal tile's owner ships them
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(C);
        const double *rd_in = C.partial + (long long)m * m * NB + NB * j;
        for (int b = 0; b < 4; b++) {
          if (!wait_flag(C, C.ready + j * m + j, b + 1, &s_state)) return;
          if (stamp_slot >= 0 && b == 3) CSTAMP(stamp_slot, 3);
          {  // panel b of L(j,j): rows 16 b .. 63, columns 16 b .. 16 b + 15
            const int c = 16 * b + 2 * (tid & 7), rr = tid >> 3;
            u32x4 v[2];
#pragma unroll
            for (int q = 0; q < 2; q++)
              v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long long)(NB * j + rr + 32 * q) * ld + C0 + c) * 8), 0, kSc1);
#pragma unroll
            for (int q = 0; q < 2; q++) {
              Pc[rr + 32 * q][c] = __longlong_as_double(((unsigned long long)v[q].y << 32) | v[q].x);
              Pc[rr + 32 * q][c + 1] = __longlong_as_double(((unsigned long long)v[q].w << 32) | v[q].z);
            }
            if (tid < 16) rdiag[16 * b + tid] = ld_sc1(rd_in + 16 * b + tid);
          }
          __syncthreads();
          if (stamp_slot >= 0 && b == 3) CSTAMP(stamp_slot, 4);
          tile_trsm_block(Pr, Pc, rdiag, b);
        }
        __syncthreads();
        if (stamp_slot >= 0) CSTAMP(stamp_slot, 5);
        store_tile(C, R0, C0, Pr, false);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(C.ready + i * m + j, 1);
      if (stamp_slot >= 0) CSTAMP(stamp_slot, 6);
      if (tid == 0) atomicAdd(C.done, 1);
      continue;
    }
    const double *y = C.A + (long long)ld * ld;          // L^-1 b after the factorisation
    double *xsol = C.A + (long long)(ld + 1) * ld;       // solution row
    const double *rd_all = C.partial + (long long)m * m * NB;
    double *Sacc = C.partial;                             // [m][NB]: running sum_{i >= j + 2} L(i,j)^T x_i of column j
    double *red = xv;                                     // [4][NB] partial dot products of the four wavefronts
    // L(i,j)^T x for the staged tile T and the vector in `vec`: thread (g, c) takes rows r = g, g + 4, ...;
    // returns the complete sum in threads 0..63 (fixed order)
    auto tile_matvec_t = [&](double (*T)[LP], const double *vec) {
      const int c = tid & 63, g = tid >> 6;
      double acc = 0;
#pragma unroll
      for (int r = 0; r < NB / 4; r++) acc += T[g + 4 * r][c] * vec[g + 4 * r];
      red[g * NB + c] = acc;
      __syncthreads();
      const double tot = (red[c] + red[NB + c]) + (red[2 * NB + c] + red[3 * NB + c]);
      __syncthreads();
      return tot;
    };
    if (inv_j >= 0) {
      // ---------------------------------------------------------------- X = L(j,j)^-1 for the backward substitution.
      // One wavefront, lane = column of X, L's entries broadcast from LDS: 2016 multiply-adds per lane (~10 us) -- on
      // nobody's critical path except for the last tile, and it turns the 64 sequential pivots of every backward
      // step into one tile product.
      const int j = inv_j;
      if (!wait_flag(C, C.ready + j * m + j, 4, &s_state)) return;
      load_tile(C, NB * j, NB * j, Pr, true);
      if (tid < NB) rdiag[tid] = ld_sc1(rd_all + NB * j + tid);
      __syncthreads();
      if (wave == 0) {
        // X is built row by row in the second LDS tile (a column in 64 registers would cost the whole kernel its second
        // workgroup per CU): X[r][c] = (delta_rc - sum_{k<r} L[r][k] X[k][c]) / L[r][r], lane = c
        for (int r = 0; r < NB; r++) {
          double s0 = lane == r ? 1.0 : 0.0, s1 = 0, s2 = 0, s3 = 0;
          int k = 0;
          for (; k + 4 <= r; k += 4) {
            s0 -= Pr[r][k] * Pc[k][lane];
            s1 -= Pr[r][k + 1] * Pc[k + 1][lane];
            s2 -= Pr[r][k + 2] * Pc[k + 2][lane];
            s3 -= Pr[r][k + 3] * Pc[k + 3][lane];
          }
          for (; k < r; k++) s0 -= Pr[r][k] * Pc[k][lane];
          Pc[r][lane] = ((s0 + s1) + (s2 + s3)) * rdiag[r];
        }
        double *out = C.linv + (long long)j * NB * NB;
        for (int r = 0; r < NB; r++) st_sc1(out + r * NB + lane, Pc[r][lane]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(C.invready, 1);  // a counter: the chain waits for all of them at once
      }
      continue;
    }
    if (task.x == 2) {
      // ---------------------------------------------------------------- backward substitution, the chain:
      // x_j = L(j,j)^-T (y_j - L(j+1,j)^T x_{j+1} - S_j), S_j delivered by the other workgroups.  Per column: two
      // tile products (no pivot loop); x_j is flagged for the far links one step later, when its write-through
      // store has landed behind the next column's tile loads, so the chain itself never waits for a store.
      CSTAMP(0, 8);
      if (!wait_flag(C, C.done, nF, &s_state)) return;  // every tile of L (and y = L^-1 b) is published
      if (!wait_flag(C, C.invready, m, &s_state)) return;  // and every inverse
      const __amdgpu_buffer_rsrc_t rsi = linv_rsrc(C);
      double *svec = rdiag;  // right-hand side of the column (rdiag is free here)
      for (int j = m - 1; j >= 0; j--) {
        if (j == 0) CSTAMP(0, 9);
        const int2 ci = C.colinfo[j];
        const bool near = ci.y != 0;  // tile (j + 1, j) exists
        if (near) load_tile(C, NB * (j + 1), NB * j, Pr, true);
        load_tile_rs(rsi, NB, NB * j, 0, Pc);
        double yj = 0;
        if (tid < NB) yj = ld_sc1(y + NB * j + tid);
        // (all vector-memory operations issued so far have completed once the tiles are in LDS: x_{j+1} is visible)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (j + 1 < m && tid == 0) st_flag(C.xready + j + 1, 1);
        const int links = ci.x;
        if (links > 0 && !wait_flag(C, C.pcount + j, links, &s_state)) return;
        __syncthreads();
        double sj = yj;
        if (tid < NB && links > 0) sj -= ld_sc1(Sacc + NB * j + tid);
        if (near) {
          const double nv = tile_matvec_t(Pr, col);  // col = x_{j+1}
          sj -= nv;
        }
        if (tid < NB) svec[tid] = sj;
        __syncthreads();
        const double xj = tile_matvec_t(Pc, svec);  // X^T s: X is lower triangular, zeros above
        if (tid < NB) {
          col[tid] = xj;
          st_sc1(xsol + NB * j + tid, xj);
        }
        __syncthreads();
        if (j == 0) CSTAMP(0, 10);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(C.xready + 0, 1);
      continue;
    }
    {
      // ---------------------------------------------------------------- S_j += L(i,j)^T x_i for i >= j + 2, one link
      // of column j's chain per task: links run over the existing tiles i = m-1, m-2, ... (fixed order: deterministic sums)
      const int i = task.y, j = task.z;   // ticket order: i descending, then j descending (closest to the chain first)
      if (!wait_flag(C, C.xready + i, 1, &s_state)) return;
      load_tile(C, NB * i, NB * j, Pr, true);
      if (tid < NB) col[tid] = ld_sc1(xsol + NB * i + tid);
      __syncthreads();
      const double pv = tile_matvec_t(Pr, col);
      const int link = task.w;  // existing far tiles of column j below this one
      if (link > 0 && !wait_flag(C, C.pcount + j, link, &s_state)) return;
      if (tid < NB) st_sc1(Sacc + NB * j + tid, (link > 0 ? ld_sc1(Sacc + NB * j + tid) : 0.0) + pv);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(C.pcount + j, link + 1);
    }
  }
}

int chol_ws_ints(int m) { return 16 + (m + 1) * m + 3 * m; }

}  // namespace

// ---- plans -----------------------------------------------------------------------------------------------------
struct vo::CholPlan {
  int m = 0, n_tasks = 0, n_factor = 0, n_tiles = 0, depth = 0;
  vo::DevBuf dev;  // [tasks int4][rowmask u64 (m + 1)][colinfo int2 (m)]
  const int4 *tasks = nullptr;
  const unsigned long long *rowmask = nullptr;
  const int2 *colinfo = nullptr;
};

// Symbolic tile Cholesky: `pattern[i]` (bit k: tile (i, k), k <= i, of the lower triangle of A may be non-zero; NULL =
// dense) -> the tiles of L (fill included), the length of the longest chain of dependent tile columns, the number of
// tiles.
void vo::chol_symbolic(int m, const unsigned long long *pattern, unsigned long long *lmask /*[m]*/, int *depth, int *n_tiles) {
  for (int i = 0; i < m; i++) {
    const unsigned long long lower = (i == 63 ? ~0ull : ((1ull << (i + 1)) - 1ull));
    lmask[i] = (pattern ? pattern[i] : ~0ull) & lower;
    lmask[i] |= 1ull << i;
  }
  for (int j = 0; j < m; j++)  // every pair of rows i1 > i2 > j with tiles in column j creates tile (i1, i2)
    for (int i2 = j + 1; i2 < m; i2++) {
      if (!((lmask[i2] >> j) & 1ull)) continue;
      for (int i1 = i2 + 1; i1 < m; i1++)
        if ((lmask[i1] >> j) & 1ull) lmask[i1] |= 1ull << i2;
    }
  int nt = 0, dmax = 0;
  std::vector<int> d((size_t)m, 1);
  for (int j = 0; j < m; j++) {  // column j's diagonal tile needs every column k < j in which row j has a tile
    for (int k = 0; k < j; k++)
      if ((lmask[j] >> k) & 1ull) d[j] = std::max(d[j], d[k] + 1);
    dmax = std::max(dmax, d[j]);
    nt += __builtin_popcountll(lmask[j]);
  }
  if (depth) *depth = dmax;
  if (n_tiles) *n_tiles = nt;
}

vo::CholPlan *vo::chol_plan_create(int m, const unsigned long long *pattern) {
  if (m < 1 || m > 64) return nullptr;
  std::vector<unsigned long long> lm((size_t)m + 1);
  int depth = 0, nt = 0;
  vo::chol_symbolic(m, pattern, lm.data(), &depth, &nt);
  lm[m] = m == 64 ? ~0ull : ((1ull << m) - 1ull);  // the right-hand-side row
  auto has = [&](int i, int j) { return ((lm[i] >> j) & 1ull) != 0; };
  std::vector<int4> tasks;
  int n_factor = 0;
  // Ticket order (see the kernel): tile columns by their level in the dependency graph, so that the first columns of
  // independent parts of the matrix are handed out together.  Everything column j waits for lies in columns of a lower
  // level: tiles (i, k) and (j, k) both exist only if row j has a tile in column k, i.e. level(k) < level(j).
  std::vector<int> level((size_t)m, 1), seq((size_t)m);
  for (int j = 0; j < m; j++) {
    for (int k = 0; k < j; k++)
      if (has(j, k)) level[j] = std::max(level[j], level[k] + 1);
    seq[j] = j;
  }
  std::stable_sort(seq.begin(), seq.end(), [&](int a, int b) { return level[a] < level[b]; });
  for (int p = 0; p < m; p++) {
    const int j = seq[p];
    for (int i = j; i <= m; i++)
      if (has(i, j)) tasks.push_back(make_int4(0, i, j, 0)), n_factor++;
    if (p >= 2) tasks.push_back(make_int4(1, 0, seq[p - 2], 0));
  }
  for (int p = std::max(0, m - 2); p < m; p++) tasks.push_back(make_int4(1, 0, seq[p], 0));
  tasks.push_back(make_int4(2, 0, 0, 0));
  std::vector<int2> colinfo((size_t)m);
  for (int j = 0; j < m; j++) {
    int far = 0;
    for (int i = j + 2; i < m; i++) far += has(i, j) ? 1 : 0;
    colinfo[j] = make_int2(far, j + 1 < m && has(j + 1, j) ? 1 : 0);
  }
  for (int i = m - 1; i >= 2; i--)      // far links: i descending, then j descending (closest to the chain first);
    for (int j = i - 2; j >= 0; j--) {  // aux = the existing far tiles of column j below this one (its turn in the sum)
      if (!has(i, j)) continue;
      int below = 0;
      for (int r = i + 1; r < m; r++) below += has(r, j) ? 1 : 0;
      tasks.push_back(make_int4(3, i, j, below));
    }
  vo::CholPlan *P = new vo::CholPlan();
  P->m = m, P->n_tasks = (int)tasks.size(), P->n_factor = n_factor, P->n_tiles = nt, P->depth = depth;
  const size_t o_mask = tasks.size() * sizeof(int4), o_col = o_mask + (size_t)(m + 1) * 8, total = o_col + (size_t)m * sizeof(int2);
  std::vector<uint8_t> img(total);
  memcpy(img.data(), tasks.data(), o_mask);
  memcpy(img.data() + o_mask, lm.data(), (size_t)(m + 1) * 8);
  memcpy(img.data() + o_col, colinfo.data(), (size_t)m * sizeof(int2));
  if (P->dev.reserve(total) != VO_OK || hipMemcpy(P->dev.p, img.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
    P->dev.release();
    delete P;
    return nullptr;
  }
  P->tasks = reinterpret_cast<const int4 *>(P->dev.p);
  P->rowmask = reinterpret_cast<const unsigned long long *>(reinterpret_cast<uint8_t *>(P->dev.p) + o_mask);
  P->colinfo = reinterpret_cast<const int2 *>(reinterpret_cast<uint8_t *>(P->dev.p) + o_col);
  return P;
}

void vo::chol_plan_destroy(vo::CholPlan *p) {
  if (!p) return;
  p->dev.release();
  delete p;
}
void vo::chol_plan_info(const vo::CholPlan *p, int *n_tiles, int *depth) {
  if (n_tiles) *n_tiles = p ? p->n_tiles : 0;
  if (depth) *depth = p ? p->depth : 0;
}

namespace {
// dense plans, one per matrix size ever used by this process (never freed: a few KB each)
const vo::CholPlan *dense_plan(int m) {
  static std::mutex mu;
  static const vo::CholPlan *cache[65] = {nullptr};
  std::lock_guard<std::mutex> lock(mu);
  if (!cache[m]) cache[m] = vo::chol_plan_create(m, nullptr);
  return cache[m];
}


}  // namespace

size_t vo::chol_workspace_bytes(int ld) {
  const int m = ld / NB;
  const size_t ints = ((size_t)chol_ws_ints(m) * 4 + 255) & ~(size_t)255;
  return ints + ((size_t)m * m * NB + (size_t)m * NB + (size_t)m * NB * NB) * 8 + (size_t)2 * m * 16 * 8;
}

void vo::chol_factor_solve(double *A, int ld, void *workspace, hipStream_t st, const vo::CholPlan *plan) {
  const int m = ld / NB;
  if (!plan || plan->m != m) plan = dense_plan(m);
  if (!plan) return;  // (allocation failure: the caller's fail flag stays clear and the solution row untouched -- reported by the next HIP call)
  int *wsI = reinterpret_cast<int *>(workspace);
  const size_t ints = ((size_t)chol_ws_ints(m) * 4 + 255) & ~(size_t)255;
  // everything but the fail flag (word 0, owned by the caller) starts at zero
  (void)hipMemsetAsync(wsI + 1, 0, ints - 4, st);
  CholCtx C;
  C.A = A, C.ld = ld, C.m = m;
  C.fail = wsI, C.ticket = wsI + 1, C.done = wsI + 2, C.ready = wsI + 16, C.xready = C.ready + (m + 1) * m, C.pcount = C.xready + m, C.invready = C.pcount + m;
  C.partial = reinterpret_cast<double *>(reinterpret_cast<uint8_t *>(workspace) + ints);
  C.linv = C.partial + (size_t)m * m * NB + (size_t)m * NB;
  C.stamps = reinterpret_cast<unsigned long long *>(C.linv + (size_t)m * NB * NB);
  C.spin_limit = 4000000;  // ~ a second of polling: far beyond any healthy wait (a factorisation lasts ~1 ms)
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const size_t lds = (size_t)(2 * NB * LP + 2 * NB + 4 * NB) * 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_chol_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  C.tasks = plan->tasks, C.rowmask = plan->rowmask, C.colinfo = plan->colinfo;
  C.n_tasks = plan->n_tasks, C.n_factor = plan->n_factor;
  const int grid = std::max(1, std::min(plan->n_tasks, 2 * n_cu));
  hipLaunchKernelGGL(k_chol_tiles, dim3(grid), dim3(256), lds, st, C);
}
