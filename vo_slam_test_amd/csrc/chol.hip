// chol.hip -- dense SPD factor + solve for the reduced camera system of a global bundle adjustment and for the
// loop-closure pose graph (6 N unknowns, N up to ~680 key-frames) on gfx950, FP64.
//
// What the reference asks of its solver: Ceres DENSE_SCHUR's Cholesky of the reduced camera matrix
// (src/optimizer_ceres.cpp:248, :600, :695) and SPARSE_NORMAL_CHOLESKY of the pose graph (:1252-1258).
//
// TWO launches (factorisation, backward substitution), persistent workgroups, tile dataflow, driven by a PLAN
// (vo::chol_plan_create): the 64 x 64 tiles of L that exist -- the structure of a reduced camera system is its
// covisibility graph; under a nested-dissection order of the key-frames whole tile columns are independent of each
// other and are factored concurrently -- and the task tables in ticket order.  A dense plan lists every tile.
//
// Factorisation (k_chol_tiles).  Tile (i, j) -- and the tile row that carries the right-hand side, which rides through
// the factorisation and comes out as L^-1 b -- is a TASK: its owner keeps the tile in MFMA accumulators, subtracts
// L(i,k) L(j,k)^T for the k < j in which both tiles exist as those become available (v_mfma_f64_16x16x4_f64 from
// LDS-staged operands), then finishes it (64 x 64 Cholesky on the diagonal, X L(j,j)^T = T below it) and publishes
// it.  Tasks are handed out by a ticket counter, columns in the order of their level in the dependency graph, so
// everything a task waits for has an earlier ticket and is either finished or running: no deadlock, whatever the
// number of resident workgroups.  Every tile is read once and written once; the serial chain that remains is
// diag(j) -> L(j+1, j) -> diag(j+1), and it is what the kernel is built around:
//   * the owner of L(j+1, j) IS the owner of diag(j+1): that tile's last update comes from LDS, sixteen columns at a
//     time as the triangular solve finishes them, not from memory (store + flag + load);
//   * the diagonal tile is published panel by panel, the tile below solves against panel b while panel b + 1's pivots
//     are being computed, and asks for the next panel early when it has fallen behind;
//   * in the pivot loop nothing waits for LDS (readlane for the next two columns, LDS factors one pivot late), the
//     reciprocal square root is the hardware estimate + one third-order step;
//   * stores leave from the wavefronts that idle during the pivots.
// Measured (BASELINE config 4, 47 tile columns, order chosen by ba.hip: 3 segments + separators, 26 dependent columns):
// column period 22.9 -> 15.5 us, factorisation 1.08 -> 0.46 ms.
//
// Backward substitution L^T x = y (k_chol_back): chains of columns joined by their sub-diagonal tiles walk down with
// the two nearest tiles of every column and the inverse of its diagonal tile (a type-1 task of the first launch) in
// registers; the segments of a nested-dissection order are separate chains that run concurrently; all other tiles are
// far links delivered by the other workgroups as running sums.  0.18 -> 0.07 ms.
//
// Hand-off between workgroups (MI355X: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed):
// every handed-off double is stored and loaded with agent-scope relaxed atomics (sc1: write-through / L1 bypass),
// the producer drains its stores (s_waitcnt vmcnt(0)), barriers, and one lane raises the tile's flag; consumers
// poll the flag with agent-scope loads.  Every poll loop is bounded: on expiry the kernel raises `fail` and all
// workgroups leave (a hang would cost the GPU box).
#include "vo_common.h"

#include <algorithm>
#include <mutex>
#include <numeric>
#include <vector>

namespace {

using namespace vo;

constexpr int NB = vo::kCholPanel;  // 64
// decided in rounds 3-4 (8-wide pivot sub-panels, flagging the sub-diagonal tile with a later panel, four
// polls in flight and plain Newton steps were each measured and lost)
constexpr int kPivotWidth = 16;   // pivot sub-panel = the 16-column panel
constexpr int kSubflagPanel = 0;  // the panel step behind which the sub-diagonal tile's stores are drained and flagged
constexpr int kNewton = 3;  // v_rsq_f64 is good to ~2^-26: 1, 2 = Newton steps (~1e-15, full precision); 3 = one third-order step (full)
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
  int *fail, *ticket, *ticket2, *done, *ready, *xready, *pcount, *invready;  // (ticket2: k_chol_back's)
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
  const int2 *colinfo;          // [m]: (far links of column j; bits 0, 1: its chain takes the tiles (j + 1, j), (j + 2, j) itself)
  int n_tasks, n_factor;        // tasks in all, factorisation tasks (what `done` counts)
  int n_front;                  // tasks of the first launch (types 0 and 1); the rest is k_chol_back's
};

// workgroup-wide wait for a flag (bounded).  Returns false when the kernel is being abandoned.  A poll is a trip to
// memory (~1 us: the loads bypass the caches); ONE poll at a time (four in flight, a quarter of a trip apart, saw a flag
// ~0.4 us sooner and cost the chain more in flag traffic: 470 -> 523 us, round 3); `seen` (optional) receives the value
// read, which may be ahead of `want` (panel counters: no need to ask again).
__device__ __forceinline__ bool wait_flag(const CholCtx &C, const int *flag, int want, int *s_state, int *seen = nullptr) {
  if (threadIdx.x == 0) {
    int ok = 1;
    int v0 = ld_flag(flag);
    for (int n = 0; v0 < want; n++) {
      if (n > C.spin_limit || ld_flag(C.fail) != 0) {
        if (n > C.spin_limit) atomicMax(C.fail, 2);  // dependency never arrived: give up loudly instead of hanging
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
      v0 = ld_flag(flag);
    }
    s_state[0] = ok;
    s_state[1] = v0;
  }
  __syncthreads();
  const bool ok = s_state[0] != 0;
  if (seen) *seen = s_state[1];
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
// 64 x 64 Cholesky of the LDS tile T (lower triangle in, factor out), whole workgroup, four panels of 16 columns:
//  (1) the first wavefront factors the panel -- lane = row, the row's 16 panel entries in registers, the pivot row's
//      entries broadcast with v_readlane (no LDS round trip on the pivot chain, which is the critical path of the
//      whole factorisation); the reciprocal square root is the hardware estimate refined by one third-order
//      step in double (full precision at a fraction of v_sqrt_f64 + v_div_f64's latency);
//  (2) all four wavefronts apply the rank-16 update to the trailing tiles on the matrix cores.
// (A lane = row version with the whole 64-entry row in registers needs > 512 registers once it is part of this
// kernel: 7 KB of scratch per lane and 15 ms per factorisation.)
// `side(b)`: what the other three wavefronts do while the first one is on panel b's pivots (they would wait at the barrier).
template <class PanelDone, class UpdateDone, class Side>
__device__ __forceinline__ bool tile_chol(double (*T)[LP], double *rdiag /*[NB]*/, double *colbuf /*[2][NB]*/, PanelDone &&panel_done,
                                          UpdateDone &&update_done, Side &&side) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q4 = lane >> 4;
  bool bad = false;
  // Pivot sub-panels of PW columns inside the 16-column panels (publication stays per panel).  A pivot updates the
  // PW - 1 - j later columns of its sub-panel itself and leaves everything to the right to an MFMA update by all four
  // wavefronts.  PW = 8 (28 instead of 120 in-loop updates per 16 columns, one more barrier + rank-8 update) was
  // measured: 11.2 against 10.4 us per diagonal tile -- the pivot loop is bound by the latency of its dependent chain
  // (~300 cycles per pivot), not by the updates issued next to it; so was forming the next diagonal ahead of the column
  // (d' = a' - x^2 / d through v_rcp_f64: 2.28 against 2.05 us per 16 pivots).  PW = 16 is the product.
  constexpr int PW = kPivotWidth, NSP = 16 / PW;
#pragma unroll
  for (int sp = 0; sp < 4 * NSP; sp++) {
    const int c0 = PW * sp, b = sp / NSP;
    const bool last_of_panel = (sp % NSP) == NSP - 1;
    if (wave == 0) {
      // One wavefront issues in order, so everything between two pivots costs the chain its issue slots and every
      // wait its latency.  The update of pivot j therefore reaches the columns in three ways: columns j + 1 and j + 2
      // (needed by the next two pivots) take their factor by v_readlane right away; the columns from j + 3 on take it
      // from an LDS broadcast ONE PIVOT LATER (double-buffered column, reads issued ahead of the pivot arithmetic), so
      // no pivot waits for an LDS round trip.
      // (LDS accesses of the loop: one base register each + immediate offsets -- addresses kept uniform are computed on
      // the scalar unit, spilled and moved back for every access)
      typedef __attribute__((address_space(3))) double lds_double;
      unsigned cb_a = (unsigned)(size_t)colbuf, cbl_a = cb_a + 8u * (unsigned)lane;
      asm volatile("" : "+v"(cb_a), "+v"(cbl_a));
      const lds_double *cb = (const lds_double *)(size_t)cb_a;  // factors: uniform address, broadcast reads
      lds_double *cbl = (lds_double *)(size_t)cbl_a;            // this lane's entry of the column
      double p[PW], rv = 0.0;
#pragma unroll
      for (int c = 0; c < PW; c++) p[c] = T[lane][c0 + c];
#pragma unroll
      for (int j = 0; j < PW; j++) {
        double f[PW];  // factors of pivot j - 1 for the columns j + 2 .. PW - 1
        if (j >= 1) {
#pragma unroll
          for (int c = j + 2; c < PW; c++) f[c] = cb[((j - 1) & 1) * NB + c0 + c];
        }
        const double d = bcast_lane(p[j], c0 + j);
        if (!(d > 0.0) || !(d < 1e300)) bad = true;  // uniform
        // 1 / sqrt(d): hardware estimate (~2^-26) + one third-order step, r (1 + e/2 + 3 e^2/8) with e = 1 - d r^2 -- four
        // dependent operations to full precision (two Newton steps are six); the pivot chain is the critical path of the
        // whole factorisation
        double r = __builtin_amdgcn_rsq(d);
        if (kNewton == 3) {
          const double e = __builtin_fma(-(d * r), r, 1.0);
          r = __builtin_fma(r * e, __builtin_fma(0.375, e, 0.5), r);
        } else {
#pragma unroll
          for (int nr = 0; nr < kNewton; nr++) r = __builtin_fma(0.5 * r, __builtin_fma(-(d * r), r, 1.0), r);
        }
        const double a = p[j] * r;  // L[lane][c0 + j] (meaningful for lane >= c0 + j; the pivot lane holds d: d r = sqrt(d))
        p[j] = a;
        rv = lane == c0 + j ? r : rv;
        if (j < PW - 1) p[j + 1] -= a * bcast_lane(a, c0 + j + 1);
        if (j < PW - 2) p[j + 2] -= a * bcast_lane(a, c0 + j + 2);
        if (j < PW - 3) cbl[(j & 1) * NB] = a;
        if (j >= 1) {
#pragma unroll
          for (int c = j + 2; c < PW; c++) p[c] -= p[j - 1] * f[c];
        }
      }
      if (lane >= c0 && lane < c0 + PW) rdiag[lane] = rv;
#pragma unroll
      for (int c = 0; c < PW; c++) T[lane][c0 + c] = (lane >= c0 + c) ? p[c] : 0.0;
    } else if (sp % NSP == 0) {
      side(b);
    }
    __syncthreads();
    if (last_of_panel) panel_done(b);  // columns 16 b .. 16 b + 15 are final: the owner ships them while the trailing update runs
    // trailing tiles: T -= P P^T with P = the PW columns just written.  After the last sub-panel of a panel: the 16 x 16
    // tiles rt >= ct > b; after an earlier one also the panel's own tile column ct = b, of which only the columns to the
    // right of the sub-panel change (the B operand of the others is zero: they hold finished columns of L).
    const int cb0 = last_of_panel ? b + 1 : b;
    const int nrow = 4 - cb0, nt = nrow * (nrow + 1) / 2;
    for (int t = wave; t < nt; t += 4) {
      int rt = cb0, u = t;
      while (u > rt - cb0) u -= rt - cb0 + 1, rt++;
      const int ct = cb0 + u;
      const int R = 16 * rt, Cc = 16 * ct;
      const bool own = ct == b;  // (only when !last_of_panel)
      double4_t acc = {T[R + q4][Cc + i16], T[R + q4 + 4][Cc + i16], T[R + q4 + 8][Cc + i16], T[R + q4 + 12][Cc + i16]};
#pragma unroll
      for (int s2 = 0; s2 < PW / 4; s2++) {
        double bop = T[Cc + i16][c0 + 4 * s2 + q4];
        if (own && Cc + i16 < c0 + PW) bop = 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[R + i16][c0 + 4 * s2 + q4], bop, acc, 0, 0, 0);
      }
      T[R + q4][Cc + i16] = acc[0], T[R + q4 + 4][Cc + i16] = acc[1], T[R + q4 + 8][Cc + i16] = acc[2], T[R + q4 + 12][Cc + i16] = acc[3];
    }
    if (last_of_panel) update_done(b);  // (contains the workgroup barrier that closes the panel step)
    else __syncthreads();
  }
  return !__syncthreads_or(bad ? 1 : 0);
}

// X L^T = W for the 64 rows of W (in place), L = lower-triangular LDS tile, rdiag = 1 / diag(L).  Rows are
// independent: every wavefront owns 16 of them and needs no workgroup barrier.  Column blocks of 16: the part of a
// block that depends on earlier blocks is an MFMA product, the 16 x 16 triangle a per-row substitution.  This runs on
// the factorisation's critical path (the tile below the diagonal closes a column), so nothing in it may wait for LDS
// in a dependent position: the product's operands are all requested before the first MFMA and accumulated in
// independent chains; the substitution fetches row c + 1 of the triangle while row c is being used, and a row's sum
// takes the newest unknown last (two dependent operations per column).
template <int B>
__device__ __forceinline__ void tile_trsm_block(double (*W)[LP], const double (*L)[LP], const double *rdiag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, q4 = lane >> 4;
  const int r0 = 16 * wave, c0 = 16 * B;
  if (B > 0) {
    constexpr int KS = 4 * B, NCH = B == 1 ? 2 : 4;
    double wa[KS > 0 ? KS : 1], lb[KS > 0 ? KS : 1];
#pragma unroll
    for (int s = 0; s < KS; s++) wa[s] = -W[r0 + i16][4 * s + q4], lb[s] = L[c0 + i16][4 * s + q4];
    double4_t acc[NCH];
    acc[0] = double4_t{W[r0 + q4][c0 + i16], W[r0 + q4 + 4][c0 + i16], W[r0 + q4 + 8][c0 + i16], W[r0 + q4 + 12][c0 + i16]};
#pragma unroll
    for (int h = 1; h < NCH; h++) acc[h] = double4_t{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; s++) acc[s % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[s], lb[s], acc[s % NCH], 0, 0, 0);
    double4_t tot = NCH == 2 ? acc[0] + acc[1] : (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    W[r0 + q4][c0 + i16] = tot[0], W[r0 + q4 + 4][c0 + i16] = tot[1], W[r0 + q4 + 8][c0 + i16] = tot[2], W[r0 + q4 + 12][c0 + i16] = tot[3];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 16) {
    double x[16], rd[16], lrow[2][16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = W[r0 + lane][c0 + c], rd[c] = rdiag[c0 + c];
    lrow[1][0] = L[c0 + 1][c0];
    x[0] *= rd[0];
#pragma unroll
    for (int c = 1; c < 16; c++) {
      if (c + 1 < 16) {
#pragma unroll
        for (int q = 0; q <= c; q++) lrow[(c + 1) & 1][q] = L[c0 + c + 1][c0 + q];
      }
      double v = x[c];
#pragma unroll
      for (int q = 0; q + 1 < c; q++) v -= x[q] * lrow[c & 1][q];
      v -= x[c - 1] * lrow[c & 1][c - 1];
      x[c] = v * rd[c];
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
  __shared__ int s_ticket, s_state[2], s_stored;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q4 = lane >> 4;
  const int m = C.m, ld = C.ld;
  if (tid == 0) s_stored = 0;
  for (;;) {
    if (tid == 0) s_ticket = atomicAdd(C.ticket, 1);
    __syncthreads();
    const int t = s_ticket;
    __syncthreads();
    if (t >= C.n_front) return;
    if (tid == 0) s_state[0] = ld_flag(C.fail);
    __syncthreads();
    if (s_state[0] != 0) return;  // abandoned (not positive definite, or a dependency timed out)
    __syncthreads();
    // Ticket order (the plan's): column by column over the tiles that exist (i = j .. m), and behind column j's tasks the
    // inverse of diagonal tile j - 2, which is complete by then (a task that polls for a long time costs the chain
    // memory bandwidth on the flag lines); the last two inverses follow the last column; then the backward chain and its
    // far links.  Everything a task waits for has an earlier ticket.
    const int4 task = C.tasks[t];
    const int inv_j = task.x == 1 ? task.z : -1;
    if (task.x == 0) {
      // ---------------------------------------------------------------- factorisation task (i, j)
      // task.w != 0 (i == j + 1): the owner of the tile below the diagonal carries on as the owner of the NEXT diagonal
      // tile (i, i).  That tile's last update is L(i,j) L(i,j)^T, and L(i,j) is what this workgroup has just produced:
      // it is applied from LDS, sixteen columns at a time as the triangular solve finishes them (in the time this
      // workgroup would otherwise spend waiting for the next panel of L(j,j)), instead of travelling to memory and back
      // (store + flag + load: ~2.5 us of every column's ~17).
      const int j = task.z;
      const int i = task.y;                                  // j <= i <= m  (i == m: right-hand-side rows)
      const bool fused = task.w == 1;
      const bool update_only = task.w == 2;  // split plans: a separator tile takes this rank's segments' updates and stays unfactored
      const int R0 = i < m ? NB * i : ld, C0 = NB * j;
      const int qr = (wave >> 1) * 32, qc = (wave & 1) * 32;
      double4_t acc[2][2], accd[2][2];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const double *p = C.A + (long long)(R0 + qr + 16 * a + q4) * ld + C0 + qc + 16 * b + i16;
          acc[a][b] = double4_t{p[0], p[4LL * ld], p[8LL * ld], p[12LL * ld]};
          accd[a][b] = double4_t{0, 0, 0, 0};
          if (fused) {
            const double *pd = C.A + (long long)(R0 + qr + 16 * a + q4) * ld + R0 + qc + 16 * b + i16;
            accd[a][b] = double4_t{pd[0], pd[4LL * ld], pd[8LL * ld], pd[12LL * ld]};
          }
        }
      bool alive = true;
      const int stamp_slot = i == j ? 2 * j : (i == j + 1 ? 2 * j + 1 : -1);
      // only the tile columns k in which both L(i, k) and L(j, k) exist contribute (and only those tiles are ever
      // published: a tile that does not exist must not be waited for); the next diagonal tile takes every L(i, k)
      const unsigned long long below_j = (1ull << j) - 1ull;
      const unsigned long long kmask = C.rowmask[i] & C.rowmask[j] & below_j;
      unsigned long long kall = fused ? (C.rowmask[i] & below_j) : kmask;
      while (kall != 0ull && alive) {
        const int k = (int)__builtin_ctzll(kall);
        kall &= kall - 1ull;
        const bool in_s = ((kmask >> k) & 1ull) != 0;
        if (stamp_slot >= 0 && k == j - 1) CSTAMP(stamp_slot, 0);
        alive = wait_flag(C, C.ready + i * m + k, 1, s_state);
        if (!alive) break;
        load_tile(C, R0, NB * k, Pr, true);
        __syncthreads();
        // (the last L(j, k) is the tile the previous column has just finished: what does not need it goes first)
        if (fused) {
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
              const int r0 = qr + 16 * a, c0 = qc + 16 * b;
#pragma unroll
              for (int s = 0; s < NB / 4; s++)
                accd[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pr[r0 + i16][4 * s + q4], Pr[c0 + i16][4 * s + q4], accd[a][b], 0, 0, 0);
            }
        }
        if (in_s) {
          if (i != j) {
            alive = wait_flag(C, C.ready + j * m + k, 1, s_state);
            if (!alive) break;
            if (stamp_slot >= 0 && k == j - 1) CSTAMP(stamp_slot, 1);
            load_tile(C, NB * j, NB * k, Pc, true);
            __syncthreads();
          }
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
        }
        __syncthreads();
      }
      if (!alive) return;
      if (update_only) {
        // A(i,j) - sum over this rank's segment columns: what the ranks' all-reduce turns into the Schur complement on the
        // separators (plain stores: the next reader is another launch, behind the collective)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++) {
            double *p = C.A + (long long)(R0 + qr + 16 * a + q4) * ld + C0 + qc + 16 * b + i16;
            p[0] = acc[a][b][0], p[4LL * ld] = acc[a][b][1], p[8LL * ld] = acc[a][b][2], p[12LL * ld] = acc[a][b][3];
          }
        if (tid == 0) atomicAdd(C.done, 1);
        continue;
      }
      if (stamp_slot >= 0) CSTAMP(stamp_slot, 2);
      // accumulators -> LDS working tile
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int g = 0; g < 4; g++) Pr[qr + 16 * a + q4 + 4 * g][qc + 16 * b + i16] = acc[a][b][g];
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(C);
      // A diagonal tile (working tile T, tile row d) is published PANEL BY PANEL (flag = panels shipped): the tile below
      // it starts its triangular solve on the first 16 columns while the pivots of the next panel are still being
      // computed.  `sub` (fused tasks): the flag of the sub-diagonal tile, whose stores are still in flight -- the three
      // wavefronts that issued them have nothing to do during the first panel's pivots: they drain, count in LDS, and
      // the last one raises the flag, so the store is on nobody's path.
      auto factor_diagonal = [&](double(*T)[LP], int d, int dslot, int *sub) {
        double *rd_out = C.partial + (long long)m * m * NB + NB * d;
        const bool ok = tile_chol(
            T, rdiag, xv,  // (xv: the backward substitution's vectors, free during the factorisation)
            [&](int b) {  // panel b: rows 16 b .. 63 x 16 columns = 2 doubles per thread and row group
              if (b == 1 && dslot >= 0) CSTAMP(dslot, 11);
              const int c = 16 * b + 2 * (tid & 7), rr = tid >> 3;  // 8 threads per row, 32 rows per pass
#pragma unroll
              for (int q = 0; q < 2; q++) {
                const int r = rr + 32 * q;
                if (r < 16 * b) continue;
                const unsigned long long lo = __double_as_longlong(T[r][c]), hi = __double_as_longlong(T[r][c + 1]);
                const u32x4 v = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(((long long)(NB * d + r) * ld + NB * d + c) * 8), 0, kSc1);
              }
              if (tid < 16) st_sc1(rd_out + 16 * b + tid, rdiag[16 * b + tid]);
            },
            [&](int b) {  // (flagging a panel one pivot phase later, when its stores have long landed, was measured: the
                          // tile below falls behind by as much and the column period grows from 22.9 to 25.4 us)
              if (b == 1 && dslot >= 0) CSTAMP(dslot, 12);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __syncthreads();
              if (tid == 0) st_flag(C.ready + d * m + d, b + 1);
              if (b == 0 && dslot >= 0) CSTAMP(dslot, 14);
              if (b == 1 && dslot >= 0) CSTAMP(dslot, 13);
            },
            [&](int b) {
              if (b != kSubflagPanel || !sub) return;
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wavefront's share of the sub-diagonal tile's stores)
              if (lane == 0 && atomicAdd(&s_stored, 1) == 2) {
                s_stored = 0;
                st_flag(sub, 1);
              }
            });
        if (!ok && tid == 0) atomicMax(C.fail, 1);  // not positive definite
        if (dslot >= 0) CSTAMP(dslot, 3);
        if (dslot >= 0) CSTAMP(dslot, 6);
        if (tid == 0) atomicAdd(C.done, 1);
      };
      if (i == j) {
        factor_diagonal(Pr, j, stamp_slot, nullptr);
        continue;
      }
      // X L(j,j)^T = T, panel by panel as the diagonal tile's owner ships them
      const double *rd_in = C.partial + (long long)m * m * NB + NB * j;
      // columns 16 b .. 16 b + 15 of X leave for memory as soon as every wavefront has solved its rows (callers: behind a
      // barrier), from wavefronts 1..3 only: the first one goes on to the next diagonal tile's pivots and must not have
      // stores of its own to wait for
      auto store_panel = [&](int b) {
        if (wave == 0) return;
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const int q = (wave - 1) * 64 + lane + 192 * t;
          if (q >= 512) break;
          const int r = q >> 3, cc = 16 * b + 2 * (q & 7);
          const unsigned long long lo = __double_as_longlong(Pr[r][cc]), hi = __double_as_longlong(Pr[r][cc + 1]);
          const u32x4 v = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(((long long)(R0 + r) * ld + C0 + cc) * 8), 0, kSc1);
        }
      };
      auto diag_rank16 = [&](int b) {  // next diagonal tile -= (columns 16 b .. 16 b + 15 of X) (same)^T
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int bq = 0; bq < 2; bq++) {
            const int r0 = qr + 16 * a, c0 = qc + 16 * bq;
#pragma unroll
            for (int s = 0; s < 4; s++)
              accd[a][bq] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pr[r0 + i16][16 * b + 4 * s + q4], Pr[c0 + i16][16 * b + 4 * s + q4], accd[a][bq], 0, 0, 0);
          }
      };
      int panels = 0;  // panels of L(j,j) known to be published
      bool have_next = false;
      u32x4 pv[2];
      double prd = 0;
      auto request_panel = [&](int b) {  // panel b of L(j,j): rows 16 b .. 63, columns 16 b .. 16 b + 15 -> registers
        const int c = 16 * b + 2 * (tid & 7), rr = tid >> 3;
#pragma unroll
        for (int q = 0; q < 2; q++)
          pv[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long long)(NB * j + rr + 32 * q) * ld + C0 + c) * 8), 0, kSc1);
        if (tid < 16) prd = ld_sc1(rd_in + 16 * b + tid);
      };
      for (int b = 0; b < 4; b++) {
        // (the barriers of the wait -- or the one in its place -- also close the solve of panel b - 1)
        if (panels >= b + 1) __syncthreads();
        else if (!wait_flag(C, C.ready + j * m + j, b + 1, s_state, &panels)) return;
        if (stamp_slot >= 0 && b == 3) CSTAMP(stamp_slot, 3);
        if (!have_next) request_panel(b);
        if (b > 0) store_panel(b - 1);
        if (fused && b > 0) diag_rank16(b - 1);  // (while the panel is on its way)
        {
          const int c = 16 * b + 2 * (tid & 7), rr = tid >> 3;
#pragma unroll
          for (int q = 0; q < 2; q++) {
            Pc[rr + 32 * q][c] = __longlong_as_double(((unsigned long long)pv[q].y << 32) | pv[q].x);
            Pc[rr + 32 * q][c + 1] = __longlong_as_double(((unsigned long long)pv[q].w << 32) | pv[q].z);
          }
          if (tid < 16) rdiag[16 * b + tid] = prd;
        }
        __syncthreads();
        // a workgroup that has fallen behind the diagonal tile's owner (the usual case: its last update could only start
        // when the previous column closed) asks for the next panel before it solves this one
        have_next = b < 3 && panels >= b + 2;
        if (have_next) request_panel(b + 1);
        if (stamp_slot >= 0 && b == 3) CSTAMP(stamp_slot, 4);
        if (b == 0) tile_trsm_block<0>(Pr, Pc, rdiag);
        else if (b == 1) tile_trsm_block<1>(Pr, Pc, rdiag);
        else if (b == 2) tile_trsm_block<2>(Pr, Pc, rdiag);
        else tile_trsm_block<3>(Pr, Pc, rdiag);
      }
      __syncthreads();
      if (stamp_slot >= 0) CSTAMP(stamp_slot, 5);
      store_panel(3);
      if (fused) {
        diag_rank16(3);
        // the next diagonal tile's working copy goes to the other LDS tile: Pr is the source of the sub-diagonal tile's store
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int g = 0; g < 4; g++) Pc[qr + 16 * a + q4 + 4 * g][qc + 16 * b + i16] = accd[a][b][g];
        __syncthreads();
        if (tid == 0) atomicAdd(C.done, 1);
        if (stamp_slot >= 0) CSTAMP(stamp_slot, 6);
        if (stamp_slot >= 0) CSTAMP(stamp_slot + 1, 2);
        factor_diagonal(Pc, i, stamp_slot >= 0 ? stamp_slot + 1 : -1, C.ready + i * m + j);
        continue;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(C.ready + i * m + j, 1);
      if (stamp_slot >= 0) CSTAMP(stamp_slot, 6);
      if (tid == 0) atomicAdd(C.done, 1);
      continue;
    }
    const double *rd_all = C.partial + (long long)m * m * NB;
    {
      // ---------------------------------------------------------------- X = L(j,j)^-1 for the backward substitution.
      // One wavefront, lane = column of X, L's entries broadcast from LDS: 2016 multiply-adds per lane (~10 us) -- on
      // nobody's critical path except for the last tile, and it turns the 64 sequential pivots of every backward
      // step into one tile product.
      const int j = inv_j;
      if (!wait_flag(C, C.ready + j * m + j, 4, s_state)) return;
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
    }
  }
}

// The backward substitution L^T x = y: its own launch behind the factorisation (own register allocation: the chain keeps
// four tiles in registers; as part of k_chol_tiles it pushed that kernel to 442 registers).  Tasks: the plan's types 2
// (the chain, first ticket) and 3 (far links).
__global__ __launch_bounds__(256, 2) void k_chol_back(CholCtx C) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double(*Pr)[LP] = reinterpret_cast<double(*)[LP]>(lds);
  double *col = lds + 2 * NB * LP, *rdiag = col + NB, *xv = rdiag + NB;
  __shared__ int s_ticket, s_state[2], s_pc;
  const int tid = threadIdx.x;
  const int m = C.m, ld = C.ld;
  (void)rdiag;
  for (;;) {
    if (tid == 0) s_ticket = atomicAdd(C.ticket2, 1);
    __syncthreads();
    const int t = s_ticket + C.n_front;
    __syncthreads();
    if (t >= C.n_tasks) return;
    if (tid == 0) s_state[0] = ld_flag(C.fail);
    __syncthreads();
    if (s_state[0] != 0) return;  // abandoned (not positive definite, or a dependency timed out)
    __syncthreads();
    const int4 task = C.tasks[t];
    const double *y = C.A + (long long)ld * ld;          // L^-1 b after the factorisation
    double *xsol = C.A + (long long)(ld + 1) * ld;       // solution row
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
    if (task.x == 2) {
      // ---------------------------------------------------------------- backward substitution, the chain:
      // x_j = L(j,j)^-T (y_j - sum_{i = j+1, j+2} L(i,j)^T x_i - S_j), S_j = the far links (i >= j + 3) delivered by the
      // other workgroups.  A step must not contain a dependent trip to memory (~1 us each, 47 steps): the two near
      // tiles, the inverse of the diagonal tile, y and the state of the far links of column j - 1 are requested during
      // step j and stay in REGISTERS -- a thread owns 8 rows x 2 columns of a tile, forms its part of T^T v there, and
      // the 8 parts of a column meet in LDS (fixed order).  x_i is flagged for the far links one step after it was
      // computed (its write-through store has landed by then), and a far link has two more steps to deliver (a third near
      // tile in registers would push the kernel past 256 registers: one workgroup per CU).
      if (task.y == m - 1) CSTAMP(0, 8);
      const __amdgpu_buffer_rsrc_t rsi = linv_rsrc(C), rst = tile_rsrc(C);
      const int c = 2 * (tid & 31), rr = tid >> 5;
      double *xs = Pr[0];         // [4][NB] ring of the last solutions x_j (slot j & 3)
      double *svec = xs + 4 * NB; // [NB] right-hand side of the column
      double *red8 = svec + NB;   // [8][NB] parts of a product
      u32x4 n1[8], n2[8], ia[8], ib[8];
      auto ld8 = [&](const __amdgpu_buffer_rsrc_t rs, int pitch, int R0, int C0, u32x4(&v)[8]) {
#pragma unroll
        for (int q = 0; q < 8; q++)
          v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long long)(R0 + rr + 8 * q) * pitch + C0 + c) * 8), 0, kSc1);
      };
      auto dot8 = [&](const u32x4(&v)[8], const double *vec, double &a0, double &a1) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const double x = vec[rr + 8 * q];
          a0 = __builtin_fma(__longlong_as_double(((unsigned long long)v[q].y << 32) | v[q].x), x, a0);
          a1 = __builtin_fma(__longlong_as_double(((unsigned long long)v[q].w << 32) | v[q].z), x, a1);
        }
      };
      auto issue_near = [&](int j, int mask) {
        if (mask & 1) ld8(rst, ld, NB * (j + 1), NB * j, n1);
        if (mask & 2) ld8(rst, ld, NB * (j + 2), NB * j, n2);
      };
      auto sum8 = [&]() {  // thread tid < NB: the complete product of its column
        double t = 0;
#pragma unroll
        for (int g = 0; g < 8; g++) t += red8[g * NB + tid];
        return t;
      };
      // state requested one step ahead: y_j, the far links' counter of column j, the tiles
      const int j_hi = task.y, j_lo = task.z;  // this chain's columns
      int2 ci = C.colinfo[j_hi];
      double yj = tid < NB ? ld_sc1(y + NB * j_hi + tid) : 0.0;
      if (tid == 0) s_pc = 0;  // (the first column's links, if any, are waited for)
      __syncthreads();
      ld8(rsi, NB, NB * j_hi, 0, ia);
      auto step = [&](int j, u32x4(&cur)[8], u32x4(&nxt)[8]) -> bool {
        if (j == 0) CSTAMP(0, 9);
#ifdef VO_CHOL_STAMPS
        const unsigned long long w0 = wall_clock64();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this column's tiles are here, and x_{j+1} has landed
#ifdef VO_CHOL_STAMPS
        if (tid == 0) C.stamps[1 * 16 + 8] += wall_clock64() - w0;  // time spent waiting for the requested tiles
#endif
        if (j < j_hi && tid == 0) st_flag(C.xready + j + 1, 1);
        const int links = ci.x, near = ci.y;
        // the far links: normally complete a step ago (pc was read then); S_j is requested first, ahead of the bulk loads
        const bool fast = links == 0 || s_pc >= links;  // (s_pc: written by thread 0 before the barrier that closed the last step)
        double sfar = 0;
        if (links > 0 && fast && tid < NB) sfar = ld_sc1(Sacc + NB * j + tid);
        int2 cn = ci;
        int pcn = 0;
        double yn = 0;
        if (j > j_lo) {
          cn = C.colinfo[j - 1];
          if (tid == 0 && cn.x > 0) pcn = ld_flag(C.pcount + j - 1);
          if (tid < NB) yn = ld_sc1(y + NB * (j - 1) + tid);
        }
        double a0 = 0, a1 = 0;
        if (near & 1) dot8(n1, xs + ((j + 1) & 3) * NB, a0, a1);
        if (near & 2) dot8(n2, xs + ((j + 2) & 3) * NB, a0, a1);
        if (j > j_lo) {
          issue_near(j - 1, cn.y);
          ld8(rsi, NB, NB * (j - 1), 0, nxt);
        }
        red8[rr * NB + c] = a0, red8[rr * NB + c + 1] = a1;
#ifdef VO_CHOL_STAMPS
        if (!fast && tid == 0) C.stamps[1 * 16 + 9] += 1;
#endif
        if (!fast) {  // (rare: a far link is late)
          if (!wait_flag(C, C.pcount + j, links, s_state)) return false;
          if (tid < NB) sfar = ld_sc1(Sacc + NB * j + tid);
        }
        __syncthreads();
        if (tid < NB) svec[tid] = (yj - sfar) - sum8();
        __syncthreads();
        double b0 = 0, b1 = 0;
        dot8(cur, svec, b0, b1);  // X^T s: X = L(j,j)^-1 is lower triangular, zeros above
        red8[rr * NB + c] = b0, red8[rr * NB + c + 1] = b1;
        __syncthreads();
        if (tid < NB) {
          const double xj = sum8();
          xs[(j & 3) * NB + tid] = xj;
          st_sc1(xsol + NB * j + tid, xj);
          if (tid == 0) s_pc = pcn;
        }
        __syncthreads();
        ci = cn, yj = yn;
        if (j == 0) CSTAMP(0, 10);
        return true;
      };
      bool ok = true;
      for (int j = j_hi; j >= j_lo && ok; j -= 2) {
        ok = step(j, ia, ib);
        if (ok && j > j_lo) ok = step(j - 1, ib, ia);
      }
      if (!ok) return;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(C.xready + j_lo, 1);
      continue;
    }
    {
      // ---------------------------------------------------------------- S_j += L(i,j)^T x_i for the tiles the chains do not take themselves, one link
      // of column j's chain per task: links run over the existing tiles i = m-1, m-2, ... (fixed order: deterministic sums)
      const int i = task.y, j = task.z;   // ticket order: i descending, then j descending (closest to the chain first)
      load_tile(C, NB * i, NB * j, Pr, true);  // the tile first: it is staged while x_i is still on its way
      if (!wait_flag(C, C.xready + i, 1, s_state)) return;
      if (tid < NB) col[tid] = ld_sc1(xsol + NB * i + tid);
      __syncthreads();
      const double pv = tile_matvec_t(Pr, col);
      const int link = task.w;  // existing far tiles of column j below this one
      if (link > 0 && !wait_flag(C, C.pcount + j, link, s_state)) return;
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
  int m = 0, n_tasks = 0, n_factor = 0, n_front = 0, n_tiles = 0, depth = 0, n_updates = 0;
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

// The general builder.  role[j] of tile column j: 0 = not part of this plan, 1 = factored (and, with `back`, solved) by it,
// 2 = update only (its tiles (i, j), i >= j, take the products over the columns in `kcols` and stay unfactored).  `kcols`:
// the columns whose tiles may be waited for / multiplied (bit k).  The whole matrix: role = 1 everywhere, kcols = all.
static vo::CholPlan *build_plan(int m, const unsigned long long *pattern, const std::vector<uint8_t> &role, unsigned long long kcols,
                                bool front, bool back) {
  if (m < 1 || m > 64) return nullptr;
  std::vector<unsigned long long> lm((size_t)m + 1);
  int depth = 0, nt = 0;
  vo::chol_symbolic(m, pattern, lm.data(), &depth, &nt);
  lm[m] = m == 64 ? ~0ull : ((1ull << m) - 1ull);  // the right-hand-side row
  auto has = [&](int i, int j) { return ((lm[i] >> j) & 1ull) != 0; };
  auto below = [](int j) { return (1ull << j) - 1ull; };
  std::vector<int4> tasks;
  int n_factor = 0;
  // Ticket order (see the kernel): tile columns by their level in the dependency graph, so that the first columns of
  // independent parts of the matrix are handed out together.  Everything column j waits for lies in columns of a lower
  // level: tiles (i, k) and (j, k) both exist only if row j has a tile in column k, i.e. level(k) < level(j).
  std::vector<int> level((size_t)m, 1), seq;
  for (int j = 0; j < m; j++) {
    if (role[j] != 1) continue;
    for (int k = 0; k < j; k++)
      if (role[k] == 1 && ((kcols >> k) & 1ull) && has(j, k)) level[j] = std::max(level[j], level[k] + 1);
    seq.push_back(j);
  }
  std::stable_sort(seq.begin(), seq.end(), [&](int a, int b) { return level[a] < level[b]; });
  const int ns = (int)seq.size();
  if (front) {
    for (int p = 0; p < ns; p++) {
      const int j = seq[p];
      for (int i = j; i <= m; i++) {
        if (!has(i, j)) continue;
        n_factor++;
        if (i == j && j > 0 && has(j, j - 1) && role[j - 1] == 1) continue;  // this diagonal tile belongs to the task of tile (j, j - 1)
        tasks.push_back(make_int4(0, i, j, (i == j + 1 && i < m && role[i] == 1) ? 1 : 0));
      }
      if (p >= 2) tasks.push_back(make_int4(1, 0, seq[p - 2], 0));
    }
    for (int p = std::max(0, ns - 2); p < ns; p++) tasks.push_back(make_int4(1, 0, seq[p], 0));
    // update-only tiles: behind every column they read
    for (int j = 0; j < m; j++) {
      if (role[j] != 2) continue;
      for (int i = j; i <= m; i++)
        if (has(i, j) && (lm[i] & lm[j] & kcols & below(j)) != 0ull) tasks.push_back(make_int4(0, i, j, 2)), n_factor++;
    }
  }
  const int n_front = (int)tasks.size();
  // Backward substitution.  A chain = a run of columns j_hi .. j_lo joined by their sub-diagonal tiles; it walks them
  // with the tiles (j + 1, j) and (j + 2, j) of its own columns in registers.  Where the sub-diagonal tile is missing
  // (the first column of a nested-dissection segment) a new chain starts: the segments' chains run concurrently once
  // the separators' unknowns are there.  Every other tile (i, j) is a far link: S_j += L(i,j)^T x_i by whichever
  // workgroup takes it, in the order i descending (a running sum: deterministic).
  std::vector<int> chain_top((size_t)m, -1);
  std::vector<int2> colinfo((size_t)m, make_int2(0, 0));
  if (back) {
    for (int j = m - 1; j >= 0; j--)
      if (role[j] == 1) chain_top[j] = (j == m - 1 || !has(j + 1, j) || role[j + 1] != 1) ? j : chain_top[j + 1];
    for (int j = m - 1; j >= 0; j--)
      if (role[j] == 1 && chain_top[j] == j) {
        int lo = j;
        while (lo > 0 && role[lo - 1] == 1 && chain_top[lo - 1] == j) lo--;
        tasks.push_back(make_int4(2, j, lo, 0));
      }
    auto is_near = [&](int i, int j) { return i - j <= 2 && i <= chain_top[j]; };
    for (int j = 0; j < m; j++) {
      if (role[j] != 1) continue;
      int far = 0, near = 0;
      for (int i = j + 1; i < m; i++) {
        if (!has(i, j)) continue;
        if (is_near(i, j)) near |= 1 << (i - j - 1);
        else far++;
      }
      colinfo[j] = make_int2(far, near);
    }
    for (int i = m - 1; i >= 1; i--)      // far links: i descending, then j descending (closest to the chain first);
      for (int j = i - 1; j >= 0; j--) {  // aux = the far links of column j with a larger i (its turn in the sum)
        if (role[j] != 1 || !has(i, j) || is_near(i, j)) continue;
        int before = 0;
        for (int r = i + 1; r < m; r++) before += (has(r, j) && !is_near(r, j)) ? 1 : 0;
        tasks.push_back(make_int4(3, i, j, before));
      }
  }
  std::vector<unsigned long long> rowmask((size_t)m + 1);
  for (int i = 0; i <= m; i++) rowmask[i] = lm[i] & kcols;
  int n_updates = 0;  // tile products L(i,k) L(j,k)^T of the factorisation (2 x 64^3 flop each)
  for (int i = 0; i < m; i++)
    for (int j = 0; j <= i; j++)
      if (role[j] != 0 && has(i, j)) n_updates += __builtin_popcountll(rowmask[i] & rowmask[j] & below(j));
  vo::CholPlan *P = new vo::CholPlan();
  P->n_updates = n_updates;
  P->m = m, P->n_tasks = (int)tasks.size(), P->n_factor = n_factor, P->n_front = n_front, P->n_tiles = nt, P->depth = depth;
  if (tasks.empty()) tasks.push_back(make_int4(-1, 0, 0, 0));  // (never handed out: n_tasks = 0)
  const size_t o_mask = tasks.size() * sizeof(int4), o_col = o_mask + (size_t)(m + 1) * 8, total = o_col + (size_t)m * sizeof(int2);
  std::vector<uint8_t> img(total);
  memcpy(img.data(), tasks.data(), o_mask);
  memcpy(img.data() + o_mask, rowmask.data(), (size_t)(m + 1) * 8);
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

vo::CholPlan *vo::chol_plan_create(int m, const unsigned long long *pattern) {
  if (m < 1 || m > 64) return nullptr;
  return build_plan(m, pattern, std::vector<uint8_t>((size_t)m, 1), ~0ull, true, true);
}

// Split plans of a nested-dissection order whose segments fill the tile columns [0, c0) and whose separators fill [c0, m):
// `own` = the segment columns this rank eliminates (bit j).
//   phase 1  this rank's segments: factor their columns (separator rows and the right-hand-side row included) and subtract
//            their products from the separator block, which stays unfactored (its sum over the ranks is the Schur complement)
//   phase 2  the separator block on its own: factor + backward substitution (the same on every rank)
//   phase 3  backward substitution of this rank's segment columns from the separators' solution
vo::CholPlan *vo::chol_plan_create_split(int m, const unsigned long long *pattern, int c0, unsigned long long own, int phase) {
  if (m < 1 || m > 64 || c0 < 1 || c0 >= m || phase < 1 || phase > 3) return nullptr;
  const unsigned long long seg = (1ull << c0) - 1ull, all = m == 64 ? ~0ull : ((1ull << m) - 1ull);
  own &= seg;
  std::vector<uint8_t> role((size_t)m, 0);
  for (int j = 0; j < m; j++) {
    const bool mine = ((own >> j) & 1ull) != 0;
    if (phase == 1) role[j] = mine ? 1 : (j >= c0 ? 2 : 0);
    else if (phase == 2) role[j] = j >= c0 ? 1 : 0;
    else role[j] = mine ? 1 : 0;
  }
  const unsigned long long kcols = phase == 2 ? (all & ~seg) : own;
  return build_plan(m, pattern, role, kcols, phase != 3, phase != 1);
}

void vo::chol_plan_destroy(vo::CholPlan *p) {
  if (!p) return;
  p->dev.release();
  delete p;
}
void vo::chol_plan_info(const vo::CholPlan *p, int *n_tiles, int *depth, int *n_updates) {
  if (n_tiles) *n_tiles = p ? p->n_tiles : 0;
  if (depth) *depth = p ? p->depth : 0;
  if (n_updates) *n_updates = p ? p->n_updates : 0;
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

// ---- block order of a large sparse system -------------------------------------------------------------------------
// A reduced camera matrix has a 6 x 6 block (c, c') wherever key-frames c and c' share a point; a pose graph's normal
// matrix one wherever an edge joins them.  Key-frames arrive in time order, so along a trajectory such a matrix is a
// band (a cyclic band once a loop is closed), and the tile Cholesky of a band is one chain of dependent tile columns
// however few tiles it has.  Cutting the band into P segments by separators as wide as the band -- segments first,
// separators last -- makes the segments' columns independent of each other: the chain shrinks to one segment plus the
// separators.  Candidates (P = 2..8, linear and cyclic cuts) are scored by the symbolic factorisation of their exact
// tile pattern, so a pattern that is not a band simply keeps the natural order; whichever order is used, the plan lists
// every tile the factor can touch (correctness never rests on the heuristic).  Segment lengths are multiples of 32
// blocks (3 tiles at 6 rows per block) so that segments do not share tiles.
static void tile_pattern(const std::vector<std::pair<int, int>> &pairs, const std::vector<int> &slot_of, int bs, int m,
                         std::vector<unsigned long long> &pat) {
  pat.assign((size_t)m, 0ull);
  for (const auto &pr : pairs) {
    const int a = slot_of[pr.first], b = slot_of[pr.second];
    const int hi = std::max(a, b), lo = std::min(a, b);
    for (int ti = bs * hi / vo::kCholPanel; ti <= (bs * hi + bs - 1) / vo::kCholPanel; ti++)
      for (int tj = bs * lo / vo::kCholPanel; tj <= (bs * lo + bs - 1) / vo::kCholPanel; tj++)
        pat[std::max(ti, tj)] |= 1ull << std::min(ti, tj);
  }
  for (int i = 0; i < m; i++) pat[i] |= 1ull << i;
  // every block's own (s, s) pair: a block that straddles a tile boundary (64 is not a multiple of 6: slots 10, 21, 42,
  // ...) also has entries in tile (k, k-1), whether or not it pairs with a slot-near neighbour
  for (size_t s = 0; s < slot_of.size(); s++) {
    const int a = slot_of[s];
    if (a < 0) continue;
    const int t0 = bs * a / vo::kCholPanel, t1 = (bs * a + bs - 1) / vo::kCholPanel;
    for (int ti = t0; ti <= t1 && ti < m; ti++)
      for (int tj = t0; tj <= ti; tj++) pat[ti] |= 1ull << tj;
  }
}

vo::CholOrder vo::chol_choose_order(int nf, int bs, const std::vector<std::pair<int, int>> &pairs, int m, int force_parts) {
  int w_lin = 0, w_cyc = 0;
  for (const auto &pr : pairs) {
    const int d = pr.first - pr.second;
    w_lin = std::max(w_lin, d), w_cyc = std::max(w_cyc, std::min(d, nf - d));
  }
  std::vector<unsigned long long> lmask((size_t)m);
  auto score = [&](vo::CholOrder &o) {
    tile_pattern(pairs, o.slot_of, bs, m, o.pattern);
    vo::chol_symbolic(m, o.pattern.data(), lmask.data(), &o.depth, &o.tiles);
  };
  vo::CholOrder best;
  best.slot_of.resize((size_t)nf);
  std::iota(best.slot_of.begin(), best.slot_of.end(), 0);
  score(best);
  if (force_parts == 1) return best;
  const int natural_depth = best.depth;
  bool have = false;
  vo::CholOrder pick = best;
  // Separators are eliminated in nested order -- every second one first (those are independent of each other: the
  // segments between them are gone), then every second one of the rest, ... -- and tried at their exact width and
  // rounded up to whole 32 key-frame units (separators that are eliminated concurrently must not share a tile either).
  for (int cyclic = 0; cyclic < 2; cyclic++) {
    const int w = cyclic ? w_cyc : w_lin;
    if (w < 1) continue;
    // Two families per (P, wide): (a) segments of `base` blocks with the last one taking the rest (base = the multiple of
    // 32 below / above the even share), (b) ALL segments of `base` blocks and the rest added to the separator that is
    // eliminated last -- then the separators start on a 32-block boundary too, and the ones that run concurrently do not
    // share a tile (config 4: 4 x 64 key-frames + separators 64 / 48 / 64 / 67: 23 dependent columns against 26).
    for (int P = 2; P <= 8; P++)
      for (int wide = 0; wide < 2; wide++)
        for (int variant = 0; variant < 6; variant++) {
          if (force_parts > 1 && P != force_parts) continue;
          const int n_sep = cyclic ? P : P - 1;
          std::vector<int> rank((size_t)n_sep), sep_len((size_t)n_sep), sep_order((size_t)n_sep);
          int sep_total = 0;
          for (int g = 0; g < n_sep; g++) {
            rank[g] = __builtin_ctz((unsigned)(g + 1));
            // the last separator of a group that runs concurrently may end anywhere: what follows depends on it
            sep_len[g] = wide && rank[g] == 0 ? (w + 31) / 32 * 32 : w;
            sep_total += sep_len[g];
            sep_order[g] = g;
          }
          std::stable_sort(sep_order.begin(), sep_order.end(), [&](int a, int b) { return rank[a] < rank[b]; });
          int in_segs = nf - sep_total;
          if (in_segs < 32 * P) continue;
          int base, last;
          if (variant < 2) {  // (a)
            base = (in_segs / P / 32 + variant) * 32;
            last = in_segs - base * (P - 1);
          } else {            // (b): base = the even share rounded down to 32, minus 0 .. 3 units
            base = (in_segs / P / 32 - (variant - 2)) * 32;
            last = base;
            if (base >= 32) {
              const int rest = in_segs - base * P;
              sep_len[sep_order[n_sep - 1]] += rest, sep_total += rest, in_segs -= rest;
            }
          }
          if (base < 32 || last < 1) continue;
          std::vector<int> sep_start((size_t)n_sep);
          int at = in_segs;
          for (int q = 0; q < n_sep; q++) sep_start[sep_order[q]] = at, at += sep_len[sep_order[q]];
          vo::CholOrder o;
          o.parts = P, o.cyclic = cyclic, o.sep = w;
          o.slot_of.assign((size_t)nf, -1);
          o.part_of.assign((size_t)nf, -1);
          o.seg_slots = in_segs;
          int next_seg = 0, pos = 0;
          for (int g = 0; g < P; g++) {
            const int len = g < P - 1 ? base : last;
            for (int q = 0; q < len; q++) o.part_of[next_seg] = g, o.slot_of[pos++] = next_seg++;
            if (g < n_sep)
              for (int q = 0; q < sep_len[g]; q++) o.slot_of[pos++] = sep_start[g] + q;
          }
          if (pos != nf) continue;
          score(o);
          if (!have || o.depth < pick.depth || (o.depth == pick.depth && o.tiles < pick.tiles)) pick = o, have = true;
        }
  }
  // a shorter chain pays for the extra fill only when it is clearly shorter
  if (have && (force_parts > 1 || pick.depth * 4 <= natural_depth * 3)) return pick;
  return best;
}


size_t vo::chol_workspace_bytes(int ld) {
  const int m = ld / NB;
  const size_t ints = ((size_t)chol_ws_ints(m) * 4 + 255) & ~(size_t)255;
  return ints + ((size_t)m * m * NB + (size_t)m * NB + (size_t)m * NB * NB) * 8 + (size_t)2 * m * 16 * 8;
}

namespace {
// which = 1: the factorisation launch, 2: the backward launch, 3: both.  xready_from < m: the unknowns of the tile columns
// [xready_from, m) are already in the solution row (split plans, phase 3).
void chol_launch(double *A, int ld, void *workspace, hipStream_t st, const vo::CholPlan *plan, int which, int xready_from) {
  const int m = ld / NB;
  int *wsI = reinterpret_cast<int *>(workspace);
  const size_t ints = ((size_t)chol_ws_ints(m) * 4 + 255) & ~(size_t)255;
  // everything but the fail flag (word 0, owned by the caller) starts at zero
  (void)hipMemsetAsync(wsI + 1, 0, ints - 4, st);
  CholCtx C;
  C.A = A, C.ld = ld, C.m = m;
  C.fail = wsI, C.ticket = wsI + 1, C.done = wsI + 2, C.ticket2 = wsI + 3, C.ready = wsI + 16, C.xready = C.ready + (m + 1) * m, C.pcount = C.xready + m, C.invready = C.pcount + m;
  if (xready_from < m) (void)hipMemsetD32Async((hipDeviceptr_t)(C.xready + xready_from), 1, (size_t)(m - xready_from), st);
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
    (void)hipFuncSetAttribute((const void *)k_chol_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  C.tasks = plan->tasks, C.rowmask = plan->rowmask, C.colinfo = plan->colinfo;
  C.n_tasks = plan->n_tasks, C.n_factor = plan->n_factor, C.n_front = plan->n_front;
  if ((which & 1) && plan->n_front > 0)
    hipLaunchKernelGGL(k_chol_tiles, dim3(std::max(1, std::min(plan->n_front, 2 * n_cu))), dim3(256), lds, st, C);
  if ((which & 2) && plan->n_tasks > plan->n_front)
    hipLaunchKernelGGL(k_chol_back, dim3(std::max(1, std::min(plan->n_tasks - plan->n_front, 2 * n_cu))), dim3(256), lds, st, C);
}
}  // namespace

void vo::chol_factor_solve(double *A, int ld, void *workspace, hipStream_t st, const vo::CholPlan *plan) {
  const int m = ld / NB;
  if (!plan || plan->m != m) plan = dense_plan(m);
  if (!plan) {  // no plan (allocation failure, m outside 1..64): raise the caller's fail flag -- never a silent no-op
    static const int abandoned = 2;
    (void)hipMemcpyAsync(workspace, &abandoned, 4, hipMemcpyHostToDevice, st);
    return;
  }
  chol_launch(A, ld, workspace, st, plan, 3, m);
}

// One phase of a split solve (plans of chol_plan_create_split; the caller sums the separator block over the ranks between
// phases 1 and 2 and the solution row after phase 3).  The inverses of the diagonal tiles and their reciprocal diagonals
// (phase 1) stay in the workspace for phase 3; every phase starts from fresh flags.
void vo::chol_split_phase(double *A, int ld, void *workspace, hipStream_t st, const vo::CholPlan *plan, int phase, int c0) {
  const int m = ld / NB;
  if (!plan || plan->m != m) {
    static const int abandoned = 2;
    (void)hipMemcpyAsync(workspace, &abandoned, 4, hipMemcpyHostToDevice, st);
    return;
  }
  chol_launch(A, ld, workspace, st, plan, phase == 1 ? 1 : (phase == 2 ? 3 : 2), phase == 3 ? c0 : m);
}
