// match.hip -- Hamming distance matrix on gfx950 and the guided greedy matchers that consume it.
// Replaces the arithmetic of myslam::Matcher (reference src/matcher.cpp): computeDistance
// (:1240-1256) becomes one all-pairs kernel; the sequential, order-dependent assignment logic of
// searchByProjection (:18-148, :274-353) is replayed on the host over the device-computed matrix
// so that match pairs stay identical to the reference's visiting order.
#include "vo_common.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <vector>

namespace {

using namespace vo;

// K6: D[i][j] = popcount(A_i xor B_j) over 256 bits.  Each thread keeps its B descriptors in registers (16-byte loads:
// a wavefront reads whole lines) and streams the 128-row A tile from LDS (all lanes read the same address: a broadcast,
// no bank conflict); results leave as one packed store per lane per row -- the kernel is bound by the 2 bytes/pair it
// writes and by the 16 vector instructions a distance costs.
constexpr int kHamRows = 128;
typedef uint32_t ham_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ham_u32x2 __attribute__((ext_vector_type(2)));

// NC columns per thread: a workgroup owns 256 NC columns x 128 rows.  NC = 4 (round 5): at <= 1024 columns a workgroup
// writes WHOLE rows, one after the other -- every 128-byte line of the matrix is filled by one workgroup within a few
// hundred cycles (rows are 2 nb bytes apart, not a multiple of the line: with two workgroups per row the lines at the
// row's start, middle and end were written in two parts at different times) -- and a row of A read from LDS serves four
// distances instead of two.
template <int NC>
__global__ __launch_bounds__(256) void k_hamming(const uint32_t *__restrict__ A, int na, long long a_stride,
                                                 const uint32_t *__restrict__ B, int nb, long long b_stride,
                                                 uint16_t *__restrict__ D, long long d_stride) {
  __shared__ __attribute__((aligned(16))) uint32_t a[kHamRows][8];
  const int tid = threadIdx.x;
  const long long p = blockIdx.z;
  A += p * a_stride * 8;
  B += p * b_stride * 8;
  D += p * d_stride;
  const int i0 = blockIdx.y * kHamRows;
  const int j0 = (blockIdx.x * 256 + tid) * NC;
  const bool wide = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;  // uniform
  for (int c = tid; c < kHamRows * 2; c += 256) {  // 16-byte chunk c of the tile: row c / 2, half c & 1
    const int r = c >> 1;
    ham_u32x4 v = {0u, 0u, 0u, 0u};
    if (i0 + r < na) {
      const uint32_t *src = A + (long long)(i0 + r) * 8 + 4 * (c & 1);
      if (wide)
        v = *reinterpret_cast<const ham_u32x4 *>(src);
      else
        v = ham_u32x4{src[0], src[1], src[2], src[3]};
    }
    *reinterpret_cast<ham_u32x4 *>(&a[r][4 * (c & 1)]) = v;
  }
  uint32_t b[NC][8];
  {
    // columns past the end are clamped, not predicated (a predicated load becomes a branch)
#pragma unroll
    for (int k = 0; k < NC; k++) {
      const uint32_t *s0 = B + (long long)min(j0 + k, nb - 1) * 8;
      if (wide) {
        const ham_u32x4 q0 = reinterpret_cast<const ham_u32x4 *>(s0)[0], q1 = reinterpret_cast<const ham_u32x4 *>(s0)[1];
#pragma unroll
        for (int w = 0; w < 4; w++) b[k][w] = q0[w], b[k][4 + w] = q1[w];
      } else {
#pragma unroll
        for (int w = 0; w < 8; w++) b[k][w] = s0[w];
      }
    }
  }
  __syncthreads();
  if (j0 >= nb) return;
  const bool vec_store = (nb % NC) == 0;  // whole groups, and every row start is 2 NC-byte aligned relative to D
  const int rows = min(kHamRows, na - i0);
  uint16_t *o = D + (long long)i0 * nb + j0;
  for (int r = 0; r < rows; r++, o += nb) {
    int d[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) d[k] = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const uint32_t av = a[r][w];
#pragma unroll
      for (int k = 0; k < NC; k++) d[k] += __popc(av ^ b[k][w]);
    }
    if (vec_store) {
      if (NC == 2) {
        *reinterpret_cast<uint32_t *>(o) = (uint32_t)d[0] | ((uint32_t)d[1] << 16);
      } else {
        *reinterpret_cast<ham_u32x2 *>(o) =
            ham_u32x2{(uint32_t)d[0] | ((uint32_t)d[1] << 16), (uint32_t)d[NC == 2 ? 0 : 2] | ((uint32_t)d[NC == 2 ? 1 : 3] << 16)};
      }
    } else {
#pragma unroll
      for (int k = 0; k < NC; k++)
        if (j0 + k < nb) o[k] = (uint16_t)d[k];
    }
  }
}

// K6 on the matrix cores (round 6).  The VALU form above costs 16 vector instructions per distance and is bound by their
// issue (262 M wave-instructions per 1024 pairs of 1000 x 1000), not by the 2 bytes per distance it writes.  A Hamming
// distance is an integer dot product: with p_i = |A_i|, q_j = |B_j| and a' the complemented bits of A_i,
//     ham(i, j) = p_i - q_j + sum_k (2 a'_ik) b_jk,
// so v_mfma_i32_32x32x32_i8 computes 32 x 32 distances in 8 K-steps of 32 bits on operands unpacked to bytes
// (b -> {0, 1}, a' -> {0, 2}) plus a ninth step whose operands carry (p_i in four byte-sized pieces | 1 1 1 1) and
// (1 1 1 1 | -q_j in four pieces): the accumulator holds the distance itself, exact in int32.  Unpacking 4 bits to the 4
// bytes of a dword is v_bfe + v_mul_u32_u24 (t * 0x00204081 puts bit i at bit 8 i, no carries) + v_and.  (The order of the
// bits along K is whatever the unpacking makes it -- the same for both operands, which is all a dot product needs.)
//   workgroup = 4 wavefronts = 128 rows x ALL columns, in chunks of 128 columns: every wavefront unpacks one 32-column
// tile of the chunk into LDS (the MFMA "A" operand, 9 x 16 bytes per lane), keeps its own 32 rows (the "B" operand) in
// registers, and runs 4 x 9 MFMAs per chunk; M = column, N = row, so a lane ends up with 4 consecutive columns of one row
// per accumulator group: packed to u16 they go through a wave-private LDS image and leave as 16-byte stores (see
// "Emission" in the kernel: whole 128-byte lines only, which is what decides the speed of this kernel).
// Per 1024 distances: 9 MFMAs (288 of the SIMD's cycles, a third of what the 2 KB of stores leave room for at HBM
// rate) and ~45 vector instructions instead of 256.  1024 pairs of 1000 x 1000: 0.41 ms against 0.505 (in bench.py's
// extract + match leg, interleaved on one box; a linear fill of the same 2.05 GB takes 0.36-0.38).
typedef int ham_i32x4 __attribute__((ext_vector_type(4)));
typedef int ham_i32x16 __attribute__((ext_vector_type(16)));
constexpr int kHmRows = 128;       // rows per workgroup (32 per wavefront)
constexpr int kHmChunkTiles = 4;   // 32-column tiles per chunk
constexpr int kHmSteps = 9;        // 8 K-steps of 32 descriptor bits + the popcount step
constexpr int kHmXBytes = kHmChunkTiles * kHmSteps * 64 * 16;
constexpr int kHmTBytes = 32 * 256;  // a wavefront's store image: 32 rows x one chunk of 128 columns

// the 16 bits [16 h, 16 h + 16) of q as 16 bytes: bit b -> byte b = (mask's byte) if set
__device__ __forceinline__ ham_i32x4 ham_unpack16(uint32_t q, uint32_t sh, uint32_t mul, uint32_t mask) {
  ham_i32x4 r;
#pragma unroll
  for (int n = 0; n < 4; n++) r[n] = (int)(__umul24(__builtin_amdgcn_ubfe(q, sh + 4 * n, 4), mul) & mask);
  return r;
}

__device__ __forceinline__ void ham_load_desc(const uint32_t *src, bool wide, uint32_t q[8]) {
  if (wide) {
    const ham_u32x4 q0 = reinterpret_cast<const ham_u32x4 *>(src)[0], q1 = reinterpret_cast<const ham_u32x4 *>(src)[1];
#pragma unroll
    for (int w = 0; w < 4; w++) q[w] = q0[w], q[4 + w] = q1[w];
  } else {
#pragma unroll
    for (int w = 0; w < 8; w++) q[w] = src[w];
  }
}

// halfword h of each of the descriptor's 8 dwords, two per register (sel: v_perm_b32 selector of the lane's half)
__device__ __forceinline__ void ham_compress(const uint32_t q[8], uint32_t sel, uint32_t out[4], uint32_t &pop) {
  pop = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    pop += __popc(q[2 * k]) + __popc(q[2 * k + 1]);
    out[k] = __builtin_amdgcn_perm(q[2 * k + 1], q[2 * k], sel);
  }
}

constexpr int kHmSuper = 8;  // chunks whose column tiles a wavefront fetches at once (1024 columns)

template <bool VEC_STORE>
__global__ __launch_bounds__(256, 2) void k_hamming_mfma(const uint32_t *__restrict__ A, int na, long long a_stride,
                                                         const uint32_t *__restrict__ B, int nb, long long b_stride,
                                                         uint16_t *__restrict__ D, long long d_stride) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kHmXBytes + 4 * kHmTBytes];
  ham_i32x4 *X = reinterpret_cast<ham_i32x4 *>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
  uint8_t *T = lds + kHmXBytes + wave * kHmTBytes;
  const long long p = blockIdx.z;
  A += p * a_stride * 8;
  B += p * b_stride * 8;
  D += p * d_stride;
  const bool wide = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;  // uniform
  const int i0 = blockIdx.x * kHmRows + wave * 32;
  const uint32_t sel = h ? 0x07060302u : 0x05040100u;

  // this wavefront's rows: 2 x complemented bits, and (p pieces | ones) in the ninth step (lanes 0-31 only)
  ham_i32x4 Y[kHmSteps];
  {
    uint32_t q[8], c4[4], pc;
    ham_load_desc(A + (long long)min(i0 + m, na - 1) * 8, wide, q);
    ham_compress(q, sel, c4, pc);
#pragma unroll
    for (int s = 0; s < 8; s++) Y[s] = ham_unpack16(~c4[s >> 1], 16 * (s & 1), 0x00408102u, 0x02020202u);
    const uint32_t pieces = (pc >> 2) | (((pc + 1) >> 2) << 8) | (((pc + 2) >> 2) << 16) | (((pc + 3) >> 2) << 24);
    Y[8] = h ? ham_i32x4{0, 0, 0, 0} : ham_i32x4{(int)pieces, 0x01010101, 0, 0};
  }

  // Emission (VEC_STORE): a row leaves in 128-byte windows on the 128-byte LINES of memory, 8 lanes x 16 bytes per row, 8 rows
  // per instruction.  The rows are 2 nb bytes apart (2000: not a multiple of the line), so row r's windows are shifted by
  // s_r = (address of the row) mod 128 against its columns; the store image T is a ring of 256 bytes per row (the two halves
  // of a chunk), from which window k = bytes [128 k - s_r, 128 k + 128 - s_r) of the row is read once half-chunk k is in.
  // The piece in front of a row's first line boundary shares its line with the previous row's tail: it is kept in registers
  // and stored in the last step, next to that tail.  Measured on the bare store pattern (tools/microbench/write_pattern.hip,
  // 2.05 GB): 256-byte pieces at fixed column offsets 0.56-0.60 ms, whole lines 0.48-0.51, whole lines with head and tail
  // written together 0.38 = a linear fill (a partial line whose other part arrives microseconds later is what costs).
  const int erow = lane >> 3, ep = lane & 7;
  uint8_t *eline[4];  // the line the row starts in
  int elo[4], ehi[4];  // the row's bytes are [elo, ehi) from there
  ham_u32x4 ehead[4];
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int r = i0 + 8 * it + erow;
    uint8_t *rowp = reinterpret_cast<uint8_t *>(D + (long long)min(r, na - 1) * nb);
    const int s = (int)(reinterpret_cast<uintptr_t>(rowp) & 127);
    eline[it] = rowp - s;
    elo[it] = s;
    ehi[it] = r < na ? s + 2 * nb : s;
    ehead[it] = ham_u32x4{0u, 0u, 0u, 0u};
  }
  // window k of the wavefront's 32 rows: T -> memory (4 instructions)
  auto emit = [&](int k, bool first, bool heads_now) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int rr = 8 * it + erow, e = 128 * k + 16 * ep;
      const ham_u32x4 v =
          *reinterpret_cast<const ham_u32x4 *>(T + rr * 256 + (((((e - elo[it]) >> 4) & 15) ^ (rr & 15)) << 4));
      const bool in = e >= elo[it] && e < ehi[it];
      if (first && elo[it] != 0) {
        ehead[it] = v;  // (k = 0: everything in front of the first line boundary)
      } else if (in) {
        *reinterpret_cast<ham_u32x4 *>(eline[it] + e) = v;
      }
      if (heads_now) {
        const int e0 = 16 * ep;
        if (elo[it] != 0 && e0 >= elo[it] && e0 < ehi[it]) *reinterpret_cast<ham_u32x4 *>(eline[it] + e0) = ehead[it];
      }
    }
  };

  int k_half = 0;  // half-chunks (64 columns) emitted so far
  for (int sc = 0; sc < nb; sc += 32 * kHmChunkTiles * kHmSuper) {
    // No load inside the chunk loop: vmcnt counts loads and stores in one queue, and a wavefront that waited for the next
    // chunk's descriptors waited for this chunk's stores as well (measured: compute 0.39 ms + stores 0.39 ms, not
    // overlapped).  The column tiles this wavefront unpacks in the next 8 chunks (tile 4 c + wave of chunk c) are fetched
    // here, all loads in flight at once, and kept as the lane's half of each dword: 4 registers + the popcount per tile.
    uint32_t Bc[kHmSuper][4], Bp[kHmSuper];
#pragma unroll
    for (int k = 0; k < kHmSuper; k++) {
      uint32_t q[8];
      ham_load_desc(B + (long long)min(sc + (k * kHmChunkTiles + wave) * 32 + m, nb - 1) * 8, wide, q);
      ham_compress(q, sel, Bc[k], Bp[k]);
    }
    const int n_chunks = min(kHmSuper, (nb - sc + 32 * kHmChunkTiles - 1) / (32 * kHmChunkTiles));
#pragma nounroll
    for (int c = 0; c < n_chunks; c++) {
      {  // column tile 4 c + wave -> X[wave]: bits as {0, 1}, (ones | -q pieces) in the ninth step
        ham_i32x4 *x = X + wave * kHmSteps * 64 + lane;
#pragma unroll
        for (int s = 0; s < 8; s++) x[s * 64] = ham_unpack16(Bc[0][s >> 1], 16 * (s & 1), 0x00204081u, 0x01010101u);
        const uint32_t pc = Bp[0];
        const uint32_t neg = ((0u - (pc >> 2)) & 255u) | (((0u - ((pc + 1) >> 2)) & 255u) << 8) |
                             (((0u - ((pc + 2) >> 2)) & 255u) << 16) | (((0u - ((pc + 3) >> 2)) & 255u) << 24);
        x[8 * 64] = h ? ham_i32x4{0, 0, 0, 0} : ham_i32x4{0x01010101, (int)neg, 0, 0};
#pragma unroll
        for (int k = 0; k + 1 < kHmSuper; k++) {  // the next chunk's tile moves to slot 0 (v_mov: the fast class)
#pragma unroll
          for (int w = 0; w < 4; w++) Bc[k][w] = Bc[k + 1][w];
          Bp[k] = Bp[k + 1];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: stores stay in flight
      if (i0 < na) {  // wave-uniform
        const int col0 = sc + c * 32 * kHmChunkTiles;
#pragma unroll
        for (int hc = 0; hc < 2; hc++) {
          if (col0 + hc * 64 < nb) {  // uniform
#pragma unroll
            for (int t = 2 * hc; t < 2 * hc + 2; t++) {
              const ham_i32x4 *x = X + t * kHmSteps * 64 + lane;
              ham_i32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
              for (int s = 0; s < kHmSteps; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x[s * 64], Y[s], acc, 0, 0, 0);
              // lane (m, h), registers 4 g .. 4 g + 3: row i0 + m, columns 32 t + 8 g + 4 h + 0..3 of the chunk; the 16-byte
              // pieces of a row are stored at (piece ^ row): the 256-byte pitch alone would put every row in the same banks
#pragma unroll
              for (int g = 0; g < 4; g++) {
                ham_u32x2 v = {(uint32_t)acc[4 * g] | ((uint32_t)acc[4 * g + 1] << 16),
                               (uint32_t)acc[4 * g + 2] | ((uint32_t)acc[4 * g + 3] << 16)};
                *reinterpret_cast<ham_u32x2 *>(T + m * 256 + (((4 * t + g) ^ (m & 15)) << 4) + h * 8) = v;
              }
            }
            // the image is this wavefront's own: LDS operations of a wavefront complete in order, no workgroup barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (VEC_STORE) {
              if (k_half == 0) emit(0, true, false); else emit(k_half, false, false);
            } else {
              for (int e = lane; e < 32 * 64; e += 64) {
                const int rr = e >> 6, cc = hc * 64 + (e & 63);
                if (i0 + rr < na && col0 + cc < nb)
                  D[(long long)(i0 + rr) * nb + col0 + cc] =
                      *reinterpret_cast<const uint16_t *>(T + rr * 256 + ((((cc >> 3) ^ (rr & 15)) << 4) | ((cc & 7) << 1)));
              }
            }
            k_half++;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // X is rewritten by the next chunk
    }
  }
  if (VEC_STORE && i0 < na) emit(k_half, false, true);  // the rows' last pieces, and the heads of the rows behind them
}


// vo_set_option(VO_OPT_HAMMING_KERNEL): 0 = the matrix-core form (default), 1 = the VALU form
std::atomic<int> g_hamming_kernel{0};

int launch_hamming(const uint8_t *a, int na, size_t as, const uint8_t *b, int nb, size_t bs, uint16_t *d,
                   size_t ds, int n_pairs, hipStream_t st) {
  if (na <= 0 || nb <= 0 || n_pairs <= 0) return VO_OK;
  if (g_hamming_kernel.load(std::memory_order_relaxed) == 0) {
    // 16-byte stores: D, the pair stride and every row start 16-byte aligned
    const bool al16 = (reinterpret_cast<uintptr_t>(d) & 15) == 0 && (ds & 7) == 0 && (nb & 7) == 0;
    dim3 grid((na + kHmRows - 1) / kHmRows, 1, n_pairs);
    if (al16)
      hipLaunchKernelGGL(k_hamming_mfma<true>, grid, dim3(256), 0, st, reinterpret_cast<const uint32_t *>(a), na,
                         (long long)as, reinterpret_cast<const uint32_t *>(b), nb, (long long)bs, d, (long long)ds);
    else
      hipLaunchKernelGGL(k_hamming_mfma<false>, grid, dim3(256), 0, st, reinterpret_cast<const uint32_t *>(a), na,
                         (long long)as, reinterpret_cast<const uint32_t *>(b), nb, (long long)bs, d, (long long)ds);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
  }
  // the 8-byte stores of the 4-column form need 8-byte aligned rows: D itself and the pair stride
  const bool al8 = (reinterpret_cast<uintptr_t>(d) & 7) == 0 && (ds & 3) == 0;
  if (al8) {
    dim3 grid((nb + 1023) / 1024, (na + kHamRows - 1) / kHamRows, n_pairs);
    hipLaunchKernelGGL(k_hamming<4>, grid, dim3(256), 0, st, reinterpret_cast<const uint32_t *>(a), na, (long long)as,
                       reinterpret_cast<const uint32_t *>(b), nb, (long long)bs, d, (long long)ds);
  } else {
    dim3 grid((nb + 511) / 512, (na + kHamRows - 1) / kHamRows, n_pairs);
    hipLaunchKernelGGL(k_hamming<2>, grid, dim3(256), 0, st, reinterpret_cast<const uint32_t *>(a), na, (long long)as,
                       reinterpret_cast<const uint32_t *>(b), nb, (long long)bs, d, (long long)ds);
  }
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

// MapPoint::computeDescriptor (mappoint.cpp:118-179): among the N descriptors observing one map point, the one
// whose MEDIAN Hamming distance to all N (itself included, sorted row entry int(0.5 * (N - 1))) is smallest;
// strict < keeps the first.  One workgroup per map point (a batch of points per launch): descriptors staged in
// LDS at a 36-byte pitch (conflict-free column reads), one wavefront per row -- lanes are columns, the row's k-th
// smallest distance by a 9-step bisection over ballot counts -- and an LDS atomicMin over (median << 16 | row).
constexpr int kMedianMax = 1024;  // observations per map point (36 KB of LDS)
__global__ __launch_bounds__(256) void k_median_desc(const uint32_t *desc, const int *offsets, int *best_idx) {
  extern __shared__ uint32_t md_lds[];  // [n][9]
  __shared__ unsigned s_best;
  const int set = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o = offsets[set], n = offsets[set + 1] - o;
  if (n <= 0) {  // `if (desp.empty()) return;` :136-137 -- nothing selected
    if (tid == 0) best_idx[set] = -1;
    return;
  }
  for (int i = tid; i < n * 8; i += 256) md_lds[(i >> 3) * 9 + (i & 7)] = desc[(long long)o * 8 + i];
  if (tid == 0) s_best = 256u << 16;  // bestMid = 256, bestIdx = 0  :159-160
  __syncthreads();
  const int k = (n - 1) / 2;  // int(0.5 * (N - 1))  :166
  constexpr int NC = kMedianMax / 64;
  for (int row = wave; row < n; row += 4) {
    uint32_t a[8];
#pragma unroll
    for (int w = 0; w < 8; w++) a[w] = md_lds[row * 9 + w];
    int d[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
      d[c] = 0x7fff;
      const int j = lane + 64 * c;
      if (64 * c < n && j < n) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) s += __popc(a[w] ^ md_lds[j * 9 + w]);
        d[c] = s;
      }
    }
    int lo = 0, hi = 256;  // smallest v with #(d <= v) >= k + 1
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < NC; c++)
        if (64 * c < n) cnt += __popcll(__builtin_amdgcn_ballot_w64(d[c] <= mid));
      if (cnt >= k + 1) hi = mid; else lo = mid + 1;
    }
    if (lane == 0) atomicMin(&s_best, ((unsigned)lo << 16) | (unsigned)row);
  }
  __syncthreads();
  if (tid == 0) best_idx[set] = (int)(s_best & 0xffffu);
}

// ---- BoW-node searches on the device: searchByBoW x 2 (matcher.cpp:449-559, :561-677) and searchForTriangulation
// (:867-1010).  The candidate set of a query (a feature of frame A) is the list of B's features in the same vocabulary
// node; the reference walks the nodes and their features in order and a claimed B feature is out for every later query.
// One wavefront replays that loop 64 queries at a time, lane = query: every lane walks its node list (a handful of
// entries; distances are computed on the fly, there is no all-pairs matrix), proposes its claim from the taken[] state
// of the previous steps, and a lane is re-proposed after the lanes in front of it have committed iff an earlier lane of
// the step claims a feature of its list -- the scheme of k_guided_replay (csrc/guided.hip).
constexpr int HISTO_LENGTH = 30;  // matcher.cpp:13
constexpr int kNodeBow0 = 0, kNodeBow1 = 1, kNodeTri = 2;

struct NodeArgs {
  int mode, nq, nA, nB, check_rot;
  float ratio, ex, ey;
  double F[9];
  float sf[16];
  const int4 *queries;          // (A feature, begin, end in bfeat, 0) in the reference's visiting order
  const uint32_t *bfeat;
  const uint4 *descA, *descB;   // 2 per feature
  const float *angA, *angB, *xA, *yA, *urA, *xB, *yB, *urB;
  const int *octB;
  const uint8_t *b_ok;          // B features that may be claimed at all (KF-KF: valid map point; triangulation: none yet)
  int4 *claims;                 // scratch [nq]: (A feature, B feature, rotation bin, 0) in claim order
  int *match, *n_matches;       // out: [nB] (mode 0: A index per B feature) or [nA] (B index per A feature)
};

// One workgroup (a single wavefront) per pair of frames: the arguments of pair p are args[p] (device memory), so that
// the ~10 searchForTriangulation calls of LocalMapping::createNewMapPoints (localMapping.cpp:187, one per neighbour
// key-frame) or the searchByBoW calls over the loop / relocalisation candidates are ONE launch.
__global__ __launch_bounds__(64) void k_node_replay(const NodeArgs *__restrict__ args) {
  // (a reference, not a copy: the 300-byte argument block with its dynamically indexed sf[] was copied to scratch memory;
  //  read in place its fields are scalar loads and sf[i] one global load)
  const NodeArgs &P = args[blockIdx.x];
  extern __shared__ __attribute__((aligned(16))) uint8_t nr_lds[];
  __shared__ int hist[32];
  const int lane = threadIdx.x;
  const int nBa = (P.nB + 15) & ~15;
  uint8_t *taken = nr_lds;                                  // [nB]
  int *tmpb = reinterpret_cast<int *>(nr_lds + nBa);         // [nB] first claiming lane of the round
  for (int i = lane; i < nBa; i += 64) taken[i] = 0, tmpb[i] = 64;
  if (lane < 32) hist[lane] = 0;
  const int nout = P.mode == kNodeBow0 ? P.nB : P.nA;
  for (int i = lane; i < nout; i += 64) P.match[i] = -1;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const float pdf = HISTO_LENGTH / 360.0f;
  int nclaims = 0;
  for (int gb = 0; gb < P.nq; gb += 64) {
    const bool live = gb + lane < P.nq;
    const int4 q = P.queries[min(gb + lane, P.nq - 1)];
    const int i1 = q.x, b0 = q.y, b1 = live ? q.z : q.y;
    const uint4 da = P.descA[2 * i1], db = P.descA[2 * i1 + 1];
    double l0 = 0, l1 = 0, l2 = 0;
    float den = 0;
    bool stereo1 = false;
    if (P.mode == kNodeTri) {  // epipolar line of feature 1 in image 2: l = F12^T p1 (checkEpipolarConstrain, :1306-1324)
      const double x1 = P.xA[i1], y1 = P.yA[i1];
      l0 = x1 * P.F[0] + y1 * P.F[3] + P.F[6], l1 = x1 * P.F[1] + y1 * P.F[4] + P.F[7], l2 = x1 * P.F[2] + y1 * P.F[5] + P.F[8];
      den = (float)(l0 * l0 + l1 * l1);
      stereo1 = P.urA[i1] >= 0;
    }
    const int len = b1 - b0;
    int lmax = len;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o));
    bool unresolved = len > 0;
    while (__builtin_amdgcn_ballot_w64(unresolved) != 0ull) {  // uniform
      int best1 = P.mode == kNodeTri ? 50 /*TH_LOW*/ : 256, best2 = 256, bidx = -1, bidx2 = -1;
      for (int t = 0; t < lmax; t++) {
        if (!(unresolved && t < len)) continue;
        const int i2 = (int)P.bfeat[b0 + t];
        if (taken[i2] || !P.b_ok[i2]) continue;  // :494 / :603 / :916-917
        const uint4 ea = P.descB[2 * i2], eb = P.descB[2 * i2 + 1];
        const int d = __popc(da.x ^ ea.x) + __popc(da.y ^ ea.y) + __popc(da.z ^ ea.z) + __popc(da.w ^ ea.w) +
                      __popc(db.x ^ eb.x) + __popc(db.y ^ eb.y) + __popc(db.z ^ eb.z) + __popc(db.w ^ eb.w);
        if (P.mode != kNodeTri) {
          if (d < best1) best2 = best1, bidx2 = bidx, best1 = d, bidx = i2;
          else if (d < best2) best2 = d, bidx2 = i2;
        } else {
          if (d > 50 || d > best1) continue;  // an equal later distance replaces the earlier one (:928)
          const float sigma = P.sf[min(max(P.octB[i2], 0), 15)];
          if (!stereo1 && !(P.urB[i2] >= 0)) {
            const float dx = P.ex - P.xB[i2], dy = P.ey - P.yB[i2];
            if (dx * dx + dy * dy < 100 * sigma) continue;  // too close to the epipole (:932-940)
          }
          if (den == 0) continue;
          const float num = (float)(l0 * P.xB[i2] + l1 * P.yB[i2] + l2);
          if (num * num / den < 3.84f * sigma * sigma) best1 = d, bidx = i2;
        }
      }
      bool accept = unresolved && bidx >= 0;
      if (accept && P.mode != kNodeTri) accept = best1 <= 50 && (float)best1 < P.ratio * (float)best2;  // :520 / :628
      if (accept) atomicMin(&tmpb[bidx], lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // stale: an earlier lane of the step claims the feature this lane chose, or the one its runner-up distance comes
      // from (the ratio test's operand); any other feature of its list leaves the decision as it is
      bool stale = false;
      if (unresolved && bidx >= 0) {
        stale = tmpb[bidx] < lane;
        if (P.mode != kNodeTri && bidx2 >= 0) stale |= tmpb[bidx2] < lane;
      }
      const unsigned long long sm = __builtin_amdgcn_ballot_w64(stale);
      const int first_stale = sm ? (int)__builtin_ctzll(sm) : 64;
      const bool fin = unresolved && lane < first_stale, claim = fin && accept;
      const unsigned long long cm = __builtin_amdgcn_ballot_w64(claim);
      if (claim) {
        taken[bidx] = 1;
        int bin = 0;
        if (P.check_rot) {  // :524-533 (cvRound), :633-641 / :958-966 (round)
          float r = P.angA[i1] - P.angB[bidx];
          if (r < 0) r += 360.0f;
          bin = P.mode == kNodeBow0 ? (int)rintf(r * pdf) : (int)roundf(r * pdf);
          if (bin == HISTO_LENGTH) bin = 0;
          atomicAdd(&hist[bin], 1);
        }
        const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
        P.claims[nclaims + below] = make_int4(i1, bidx, bin, 0);
      }
      nclaims += (int)__popcll(cm);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (accept) tmpb[bidx] = 64;
      unresolved = unresolved && !fin;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // computeThreeMax (:1258-1304) and the pruning of the other bins
  int i1m = -1, i2m = -1, i3m = -1;
  if (P.check_rot) {
    int m1 = 0, m2 = 0, m3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int sz = hist[i];
      if (sz > m1) m3 = m2, i3m = i2m, m2 = m1, i2m = i1m, m1 = sz, i1m = i;
      else if (sz > m2) m3 = m2, i3m = i2m, m2 = sz, i2m = i;
      else if (sz > m3) m3 = sz, i3m = i;
    }
    if (m2 < 0.1f * (float)m1) i2m = i3m = -1;
    else if (m3 < 0.1f * (float)m1) i3m = -1;
  }
  __threadfence_block();
  int kept = 0;
  for (int base = 0; base < nclaims; base += 64) {
    bool keep = false;
    if (base + lane < nclaims) {
      const int4 c = P.claims[base + lane];
      keep = !P.check_rot || c.z == i1m || c.z == i2m || c.z == i3m;
      if (keep) {
        if (P.mode == kNodeBow0) P.match[c.y] = c.x;
        else P.match[c.x] = c.y;
      }
    }
    kept += (int)__popcll(__builtin_amdgcn_ballot_w64(keep));
  }
  if (lane == 0) *P.n_matches = kept;
}

// walk two ascending node lists, calling f(ia, ib) for every common node id (:541-544 lower_bound walk)
template <class F>
void for_common_nodes(const vo_bow_view &a, const vo_bow_view &b, F &&f) {
  int ia = 0, ib = 0;
  while (ia < a.n_nodes && ib < b.n_nodes) {
    if (a.node_id[ia] == b.node_id[ib]) {
      f(ia, ib);
      ia++, ib++;
    } else if (a.node_id[ia] < b.node_id[ib]) {
      ia++;
    } else {
      ib++;
    }
  }
}

constexpr int kNodeMaxB = 16384;  // B features (LDS: 5 bytes each)

// One pair of a batched search (host pointers)
struct NodePair {
  const vo_frame_view *a;
  const uint8_t *a_flag;  // skip (triangulation: already has a map point) or valid (BoW) per A feature
  const vo_bow_view *an;
  const vo_frame_view *b;
  const uint8_t *b_flag;  // blocked (triangulation) or valid (KF-KF BoW) per B feature, or NULL
  const vo_bow_view *bn;
  const double *F;        // triangulation: F12 row-major
  float ex, ey;
  int32_t *match;
  int *n_matches;
};

// gather (the reference's pointer walk, flattened) for every pair, ONE staged upload, ONE launch (a workgroup per
// pair), ONE download.  Frames that appear in several pairs (the current key-frame of createNewMapPoints, the current
// frame of a relocalisation) are staged once.
int node_search_batch(int mode, int n_pairs, const NodePair *pairs, bool a_flag_is_skip, bool b_flag_is_blocked, float ratio,
                      int check_rot, const float *scale_factors) {
  size_t max_b = 0;
  for (int p = 0; p < n_pairs; p++) {
    const NodePair &Q = pairs[p];
    const int nout = mode == kNodeBow0 ? Q.b->n : Q.a->n;
    for (int i = 0; i < nout; i++) Q.match[i] = -1;
    *Q.n_matches = 0;
    if (Q.b->n > kNodeMaxB) {
      vo::set_error("BoW-node search: %d features in the second frame exceed %d", Q.b->n, kNodeMaxB);
      return VO_ERR_CAPACITY;
    }
    max_b = std::max<size_t>(max_b, (size_t)Q.b->n);
  }
  VO_CHECK(vo::ensure_device());
  // ---- host staging image: [args][per distinct frame: desc, angle, x, y, uright, octave][per pair: queries, bfeat, ok]
  std::vector<uint8_t> img;
  auto put = [&](const void *src, size_t bytes) {
    const size_t off = (img.size() + 15) & ~(size_t)15;
    img.resize(off + std::max<size_t>(bytes, 16));
    if (bytes && src) memcpy(img.data() + off, src, bytes);
    return off;
  };
  struct FrameOff { const vo_frame_view *v; size_t desc, ang, x, y, ur, oct; };
  std::vector<FrameOff> staged;
  auto stage_frame = [&](const vo_frame_view *v) {
    for (const FrameOff &f : staged)
      if (f.v == v) return f;
    FrameOff f{v, put(v->desc, (size_t)v->n * 32), put(v->angle, (size_t)v->n * 4), put(v->x, (size_t)v->n * 4),
               put(v->y, (size_t)v->n * 4), put(v->uright, (size_t)v->n * 4), put(v->octave, (size_t)v->n * 4)};
    staged.push_back(f);
    return f;
  };
  const size_t args_off = put(nullptr, (size_t)n_pairs * sizeof(NodeArgs));
  struct PairOff { size_t q, bf, ok, claims, match, nm; int nq, nout; FrameOff fa, fb; bool live; };
  std::vector<PairOff> po(n_pairs);
  size_t out_bytes = 0;  // device-only output / scratch area behind the image
  std::vector<int> queries;
  for (int p = 0; p < n_pairs; p++) {
    const NodePair &Q = pairs[p];
    PairOff &o = po[p];
    o.nout = mode == kNodeBow0 ? Q.b->n : Q.a->n;
    o.live = Q.a->n > 0 && Q.b->n > 0;
    o.nq = 0;
    if (!o.live) continue;
    queries.clear();
    for_common_nodes(*Q.an, *Q.bn, [&](int ia, int ib) {
      for (int t = Q.an->start[ia]; t < Q.an->start[ia + 1]; t++) {
        const int i1 = (int)Q.an->feat[t];
        if (a_flag_is_skip ? Q.a_flag[i1] != 0 : Q.a_flag[i1] == 0) continue;  // :488 / :597 / :905
        queries.push_back(i1), queries.push_back(Q.bn->start[ib]), queries.push_back(Q.bn->start[ib + 1]), queries.push_back(0);
      }
    });
    o.nq = (int)queries.size() / 4;
    if (o.nq == 0) {
      o.live = false;
      continue;
    }
    o.fa = stage_frame(Q.a), o.fb = stage_frame(Q.b);
    o.q = put(queries.data(), queries.size() * 4);
    const int nbf = Q.bn->n_nodes > 0 ? Q.bn->start[Q.bn->n_nodes] : 0;
    o.bf = put(Q.bn->feat, (size_t)std::max(nbf, 1) * 4);
    std::vector<uint8_t> bok(Q.b->n, 1);
    if (Q.b_flag)
      for (int i = 0; i < Q.b->n; i++) bok[i] = b_flag_is_blocked ? (Q.b_flag[i] ? 0 : 1) : (Q.b_flag[i] ? 1 : 0);
    o.ok = put(bok.data(), (size_t)Q.b->n);
    auto reserve_out = [&](size_t bytes) {
      const size_t off = (out_bytes + 15) & ~(size_t)15;
      out_bytes = off + std::max<size_t>(bytes, 16);
      return off;
    };
    o.claims = reserve_out((size_t)o.nq * 16), o.match = reserve_out((size_t)o.nout * 4), o.nm = reserve_out(16);
  }
  std::vector<int> live;
  for (int p = 0; p < n_pairs; p++)
    if (po[p].live) live.push_back(p);
  if (live.empty()) return VO_OK;
  const size_t img_bytes = (img.size() + 255) & ~(size_t)255;
  thread_local vo::ScratchBuf dbuf;
  thread_local vo::PinnedBuf stage;
  VO_CHECK(dbuf.reserve(img_bytes + out_bytes + 256));
  uint8_t *d = dbuf.as<uint8_t>(), *dout = d + img_bytes;
  // the argument blocks (device addresses are known now), one per LIVE pair, packed at the front of the args area
  NodeArgs *hargs = reinterpret_cast<NodeArgs *>(img.data() + args_off);
  for (size_t k = 0; k < live.size(); k++) {
    const NodePair &Q = pairs[live[k]];
    const PairOff &o = po[live[k]];
    NodeArgs P{};
    P.mode = mode, P.nq = o.nq, P.nA = Q.a->n, P.nB = Q.b->n, P.check_rot = check_rot, P.ratio = ratio, P.ex = Q.ex, P.ey = Q.ey;
    for (int i = 0; i < 9; i++) P.F[i] = Q.F ? Q.F[i] : 0.0;
    for (int i = 0; i < 16; i++) P.sf[i] = scale_factors ? scale_factors[std::min(i, 7)] : 1.f;
    P.queries = reinterpret_cast<const int4 *>(d + o.q), P.bfeat = reinterpret_cast<const uint32_t *>(d + o.bf);
    P.descA = reinterpret_cast<const uint4 *>(d + o.fa.desc), P.descB = reinterpret_cast<const uint4 *>(d + o.fb.desc);
    P.angA = reinterpret_cast<const float *>(d + o.fa.ang), P.angB = reinterpret_cast<const float *>(d + o.fb.ang);
    P.xA = reinterpret_cast<const float *>(d + o.fa.x), P.yA = reinterpret_cast<const float *>(d + o.fa.y);
    P.urA = reinterpret_cast<const float *>(d + o.fa.ur);
    P.xB = reinterpret_cast<const float *>(d + o.fb.x), P.yB = reinterpret_cast<const float *>(d + o.fb.y);
    P.urB = reinterpret_cast<const float *>(d + o.fb.ur), P.octB = reinterpret_cast<const int *>(d + o.fb.oct);
    P.b_ok = d + o.ok;
    P.claims = reinterpret_cast<int4 *>(dout + o.claims), P.match = reinterpret_cast<int *>(dout + o.match);
    P.n_matches = reinterpret_cast<int *>(dout + o.nm);
    hargs[k] = P;
  }
  hipStream_t st = vo::thread_stream();
  const char *what = "BoW-node search";
  VO_CHECK(stage.reserve(std::max(img.size(), out_bytes)));
  memcpy(stage.data(), img.data(), img.size());
  VO_CHECK(vo::copy_h2d(d, stage.data(), img.size(), st, what));
  const size_t lds = ((max_b + 15) & ~(size_t)15) * 5;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void *)k_node_replay, hipFuncAttributeMaxDynamicSharedMemorySize, kNodeMaxB * 5);
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(k_node_replay, dim3((unsigned)live.size()), dim3(64), lds, st, reinterpret_cast<const NodeArgs *>(d + args_off));
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(stage.data(), dout, out_bytes, st, what));
  VO_CHECK(vo::stream_sync(st, what));
  for (int p : live) {
    memcpy(pairs[p].match, stage.data() + po[p].match, (size_t)po[p].nout * 4);
    memcpy(pairs[p].n_matches, stage.data() + po[p].nm, 4);
  }
  return VO_OK;
}

}  // namespace

namespace vo {
void set_hamming_kernel(int v) { g_hamming_kernel.store(v, std::memory_order_relaxed); }
}  // namespace vo

extern "C" {

int vo_hamming_matrix_dev(const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *d, void *stream) {
  if ((na > 0 && !a) || (nb > 0 && !b) || (na > 0 && nb > 0 && !d) || na < 0 || nb < 0) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  return launch_hamming(a, na, 0, b, nb, 0, d, 0, 1, (hipStream_t)stream);
}

int vo_hamming_matrix_batch_dev(const uint8_t *a, int na, size_t as, const uint8_t *b, int nb, size_t bs,
                                uint16_t *d, size_t ds, int n_pairs, void *stream) {
  if (!a || !b || !d || na < 0 || nb < 0 || n_pairs < 0) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  return launch_hamming(a, na, as, b, nb, bs, d, ds, n_pairs, (hipStream_t)stream);
}

int vo_hamming_matrix(const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *d) {
  if (na < 0 || nb < 0) return VO_ERR_INVALID;
  if (na == 0 || nb == 0) return VO_OK;
  if (!a || !b || !d) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  // per host thread, grow-only: the stateless entry points do not allocate after the first call at a size
  // (never freed: a few MB per calling thread for the life of the process); the calling thread's own stream
  thread_local vo::ScratchBuf da, db, dd;
  hipStream_t st = vo::thread_stream();
  VO_CHECK(vo::upload(da, a, (size_t)na * 32, st, "vo_hamming_matrix"));
  VO_CHECK(vo::upload(db, b, (size_t)nb * 32, st, "vo_hamming_matrix"));
  VO_CHECK(dd.reserve((size_t)na * nb * 2));
  VO_CHECK(launch_hamming(da.as<uint8_t>(), na, 0, db.as<uint8_t>(), nb, 0, dd.as<uint16_t>(), 0, 1, st));
  VO_CHECK(vo::copy_d2h(d, dd.p, (size_t)na * nb * 2, st, "vo_hamming_matrix"));
  return vo::stream_sync(st, "vo_hamming_matrix");
}

int vo_median_descriptor(const uint8_t *desc, int n_sets, const int32_t *offsets, int32_t *best_idx) {
  if (n_sets < 0 || (n_sets > 0 && (!offsets || !best_idx))) return VO_ERR_INVALID;
  if (n_sets == 0) return VO_OK;
  const int total = offsets[n_sets];
  if (total < 0 || (total > 0 && !desc)) return VO_ERR_INVALID;
  int nmax = 0;
  for (int s = 0; s < n_sets; s++) {
    const int n = offsets[s + 1] - offsets[s];
    if (n < 0) return VO_ERR_INVALID;
    nmax = std::max(nmax, n);
  }
  if (nmax > kMedianMax) {
    vo::set_error("vo_median_descriptor: %d observations of one map point exceed %d", nmax, kMedianMax);
    return VO_ERR_CAPACITY;
  }
  VO_CHECK(vo::ensure_device());
  thread_local vo::ScratchBuf dd, doff, dbest;
  hipStream_t st = vo::thread_stream();
  VO_CHECK(vo::upload(dd, desc, (size_t)total * 32, st, "vo_median_descriptor"));
  VO_CHECK(vo::upload(doff, offsets, (size_t)(n_sets + 1) * 4, st, "vo_median_descriptor"));
  VO_CHECK(dbest.reserve((size_t)n_sets * 4));
  const size_t lds = (size_t)std::max(nmax, 1) * 36;
  hipLaunchKernelGGL(k_median_desc, dim3(n_sets), dim3(256), lds, st, dd.as<uint32_t>(), doff.as<int>(), dbest.as<int>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(best_idx, dbest.p, (size_t)n_sets * 4, st, "vo_median_descriptor"));
  return vo::stream_sync(st, "vo_median_descriptor");
}

}  // extern "C"

extern "C" {

int vo_match_bow(const vo_frame_view *a, const uint8_t *a_valid, const vo_bow_view *an, const vo_frame_view *b,
                 const uint8_t *b_valid, const vo_bow_view *bn, int mode, float ratio, int check_rot, int32_t *match,
                 int *n_matches) {
  if (!a || !b || !an || !bn || !a_valid || !match || !n_matches || (mode != 0 && mode != 1) || (mode == 1 && !b_valid))
    return VO_ERR_INVALID;
  const NodePair pr{a, a_valid, an, b, mode == 1 ? b_valid : nullptr, bn, nullptr, 0.f, 0.f, match, n_matches};
  return node_search_batch(mode == 0 ? kNodeBow0 : kNodeBow1, 1, &pr, false, false, ratio, check_rot, nullptr);
}

int vo_match_bow_batch(int n_pairs, const vo_frame_view *const *a, const uint8_t *const *a_valid, const vo_bow_view *const *an,
                       const vo_frame_view *const *b, const uint8_t *const *b_valid, const vo_bow_view *const *bn, int mode,
                       float ratio, int check_rot, int32_t *const *match, int *n_matches) {
  if (n_pairs < 0 || (mode != 0 && mode != 1) || (n_pairs > 0 && (!a || !a_valid || !an || !b || !bn || !match || !n_matches)) ||
      (mode == 1 && n_pairs > 0 && !b_valid))
    return VO_ERR_INVALID;
  std::vector<NodePair> pr((size_t)n_pairs);
  for (int p = 0; p < n_pairs; p++) {
    if (!a[p] || !a_valid[p] || !an[p] || !b[p] || !bn[p] || !match[p] || (mode == 1 && !b_valid[p])) return VO_ERR_INVALID;
    pr[p] = NodePair{a[p], a_valid[p], an[p], b[p], mode == 1 ? b_valid[p] : nullptr, bn[p], nullptr, 0.f, 0.f, match[p], &n_matches[p]};
  }
  if (n_pairs == 0) return VO_OK;
  return node_search_batch(mode == 0 ? kNodeBow0 : kNodeBow1, n_pairs, pr.data(), false, false, ratio, check_rot, nullptr);
}

int vo_match_triangulation(const vo_frame_view *a, const uint8_t *a_has, const vo_bow_view *an, const vo_frame_view *b,
                           const uint8_t *b_has, const vo_bow_view *bn, const double F[9], float ex, float ey,
                           const float *scale_factors, int check_rot, int32_t *match12, int *n_matches) {
  if (!a || !b || !an || !bn || !a_has || !b_has || !F || !scale_factors || !match12 || !n_matches) return VO_ERR_INVALID;
  const NodePair pr{a, a_has, an, b, b_has, bn, F, ex, ey, match12, n_matches};
  return node_search_batch(kNodeTri, 1, &pr, true, true, 0.f, check_rot, scale_factors);
}

int vo_match_triangulation_batch(int n_pairs, const vo_frame_view *a, const uint8_t *a_has, const vo_bow_view *an,
                                 const vo_frame_view *const *b, const uint8_t *const *b_has, const vo_bow_view *const *bn,
                                 const double *F /*[n_pairs][9]*/, const float *ex, const float *ey, const float *scale_factors,
                                 int check_rot, int32_t *const *match12, int *n_matches) {
  if (n_pairs < 0 || !a || !a_has || !an || !scale_factors || (n_pairs > 0 && (!b || !b_has || !bn || !F || !ex || !ey || !match12 || !n_matches)))
    return VO_ERR_INVALID;
  std::vector<NodePair> pr((size_t)n_pairs);
  for (int p = 0; p < n_pairs; p++) {
    if (!b[p] || !b_has[p] || !bn[p] || !match12[p]) return VO_ERR_INVALID;
    pr[p] = NodePair{a, a_has, an, b[p], b_has[p], bn[p], F + 9 * (size_t)p, ex[p], ey[p], match12[p], &n_matches[p]};
  }
  if (n_pairs == 0) return VO_OK;
  return node_search_batch(kNodeTri, n_pairs, pr.data(), true, true, 0.f, check_rot, scale_factors);
}

}  // extern "C" (kernels below)

namespace {

struct VocabDev {
  int n_nodes, depth;
  const int *child_start, *children, *word_id;
  const uint32_t *desc;  // 8 dwords per node
  const double *weight;
};

__global__ __launch_bounds__(256) void k_bow_transform(VocabDev V, int n, const uint32_t *feat, int levelsup, int *out_word,
                                                       double *out_weight, int *out_node) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t f[8];
#pragma unroll
  for (int k = 0; k < 8; k++) f[k] = feat[8LL * i + k];
  const int nid_level = V.depth - levelsup;
  int id = 0, level = 0, nid = 0;
  while (V.word_id[id] < 0 && V.child_start[id + 1] > V.child_start[id]) {
    level++;
    const int c0 = V.child_start[id], c1 = V.child_start[id + 1];
    int best = -1, bestd = 1 << 30;
    for (int c = c0; c < c1; c++) {
      const int ch = V.children[c];
      const uint32_t *d = V.desc + 8LL * ch;
      int dist = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) dist += __popc(f[k] ^ d[k]);
      if (dist < bestd) bestd = dist, best = ch;  // strict: the first minimum wins
    }
    id = best;
    if (level == nid_level) nid = id;
  }
  out_word[i] = V.word_id[id];
  out_weight[i] = V.weight[id];
  out_node[i] = nid;
}

}  // namespace

struct vo_vocab {
  VocabDev V{};
  vo::DevBuf b_cs, b_ch, b_wid, b_desc, b_w;
};

extern "C" {

int vo_vocab_create(vo_vocab **out, int n_nodes, int depth_L, const int32_t *child_start, const int32_t *children,
                    const uint8_t *node_desc, const double *node_weight, const int32_t *word_id) {
  if (!out || n_nodes < 1 || depth_L < 0 || !child_start || !node_desc || !node_weight || !word_id) return VO_ERR_INVALID;
  const int n_children = child_start[n_nodes];
  if (n_children < 0 || (n_children > 0 && !children)) return VO_ERR_INVALID;
  for (int i = 0; i < n_nodes; i++)
    if (child_start[i] > child_start[i + 1]) return VO_ERR_INVALID;
  for (int c = 0; c < n_children; c++)
    if (children[c] <= 0 || children[c] >= n_nodes) {  // a child is never the root: guarantees termination
      vo::set_error("vo_vocab_create: child %d out of range", children[c]);
      return VO_ERR_INVALID;
    }
  VO_CHECK(vo::ensure_device());
  vo_vocab *v = new vo_vocab();
  auto up = [](vo::DevBuf &b, const void *src, size_t bytes) -> int {
    VO_CHECK(b.reserve(std::max<size_t>(bytes, 64)));
    if (bytes) VO_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return VO_OK;
  };
  int rc;
  if ((rc = up(v->b_cs, child_start, (size_t)(n_nodes + 1) * 4)) != VO_OK || (rc = up(v->b_ch, children, (size_t)n_children * 4)) != VO_OK ||
      (rc = up(v->b_wid, word_id, (size_t)n_nodes * 4)) != VO_OK || (rc = up(v->b_desc, node_desc, (size_t)n_nodes * 32)) != VO_OK ||
      (rc = up(v->b_w, node_weight, (size_t)n_nodes * 8)) != VO_OK) {
    for (vo::DevBuf *b : {&v->b_cs, &v->b_ch, &v->b_wid, &v->b_desc, &v->b_w}) b->release();
    delete v;
    return rc;
  }
  v->V.n_nodes = n_nodes, v->V.depth = depth_L;
  v->V.child_start = v->b_cs.as<int>(), v->V.children = v->b_ch.as<int>(), v->V.word_id = v->b_wid.as<int>();
  v->V.desc = v->b_desc.as<uint32_t>(), v->V.weight = v->b_w.as<double>();
  *out = v;
  return VO_OK;
}

void vo_vocab_destroy(vo_vocab *v) {
  if (!v) return;
  for (vo::DevBuf *b : {&v->b_cs, &v->b_ch, &v->b_wid, &v->b_desc, &v->b_w}) b->release();
  delete v;
}

int vo_bow_transform(const vo_vocab *v, int n, const uint8_t *desc, int levelsup, int32_t *word_id, double *weight,
                     int32_t *node_id) {
  if (!v || n < 0 || (n > 0 && (!desc || !word_id || !weight || !node_id))) return VO_ERR_INVALID;
  if (n == 0) return VO_OK;
  thread_local vo::ScratchBuf d_f, d_w, d_wt, d_n;
  VO_CHECK(d_f.reserve((size_t)n * 32));
  VO_CHECK(d_w.reserve((size_t)n * 4));
  VO_CHECK(d_wt.reserve((size_t)n * 8));
  VO_CHECK(d_n.reserve((size_t)n * 4));
  hipStream_t st = vo::thread_stream();
  VO_CHECK(vo::copy_h2d(d_f.p, desc, (size_t)n * 32, st, "vo_bow_transform"));
  hipLaunchKernelGGL(k_bow_transform, dim3((n + 255) / 256), dim3(256), 0, st, v->V, n, d_f.as<uint32_t>(), levelsup,
                     d_w.as<int>(), d_wt.as<double>(), d_n.as<int>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(word_id, d_w.p, (size_t)n * 4, st, "vo_bow_transform"));
  VO_CHECK(vo::copy_d2h(weight, d_wt.p, (size_t)n * 8, st, "vo_bow_transform"));
  VO_CHECK(vo::copy_d2h(node_id, d_n.p, (size_t)n * 4, st, "vo_bow_transform"));
  return vo::stream_sync(st, "vo_bow_transform");
}

}  // extern "C"

// ---- trackRefKeyFrame's search with the current frames resident in a frame store (vo_common.h) --------------------
int vo::bow_search_resident(const vo_vocab *v, vo_frames *frames, int slot0, int B, const vo::RefKeyFrame *kfs, float ratio,
                            int check_rot, int levelsup, int32_t *dev_assigned, int cap, int32_t *dev_n_matches, hipStream_t st) {
  if (!v || !frames || B < 1 || !kfs || !dev_assigned || !dev_n_matches) return VO_ERR_INVALID;
  const vo::FrameStoreView fs = vo::frame_store_view(frames);
  if (fs.cap != cap || cap > kNodeMaxB) {
    vo::set_error("BoW search on resident frames: %d feature slots per frame (the store has %d, the kernel handles %d)", cap, fs.cap, kNodeMaxB);
    return VO_ERR_CAPACITY;
  }
  const char *what = "trackRefKeyFrame search";
  // Frame::computeBow (frame.cpp:248-253): the node of every feature at level L - levelsup, straight from the store's slots
  thread_local vo::ScratchBuf d_w, d_wt, d_node, d_img;
  thread_local vo::PinnedBuf stage;
  const size_t N = (size_t)B * cap;
  VO_CHECK(d_w.reserve(N * 4));
  VO_CHECK(d_wt.reserve(N * 8));
  VO_CHECK(d_node.reserve(N * 4));
  hipLaunchKernelGGL(k_bow_transform, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, v->V, (int)N,
                     reinterpret_cast<const uint32_t *>(fs.desc + (size_t)slot0 * cap * 32), levelsup, d_w.as<int>(), d_wt.as<double>(),
                     d_node.as<int>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(stage.reserve(N * 4 + (size_t)B * 4 + 64));
  int *h_node = reinterpret_cast<int *>(stage.data()), *h_n = h_node + N;
  VO_CHECK(vo::copy_d2h(h_node, d_node.p, N * 4, st, what));
  VO_CHECK(vo::copy_d2h(h_n, fs.n + slot0, (size_t)B * 4, st, what));
  VO_CHECK(vo::stream_sync(st, what));
  // host: per frame its FeatureVector (node ids ascending, the features of a node in index order), the walk over the
  // nodes it shares with its key-frame, the query list in the reference's visiting order (matcher.cpp:465-545)
  std::vector<uint8_t> img;
  auto put = [&](const void *src, size_t bytes) {
    const size_t off = (img.size() + 15) & ~(size_t)15;
    img.resize(off + std::max<size_t>(bytes, 16));
    if (bytes && src) memcpy(img.data() + off, src, bytes);
    return off;
  };
  const size_t args_off = put(nullptr, (size_t)B * sizeof(NodeArgs));
  struct Off { size_t q, bf, ok, adesc, aang, claims; int nq, nB; };
  std::vector<Off> off((size_t)B);
  size_t claims_bytes = 0;
  int max_b = 1;
  std::vector<int> order, queries;
  std::vector<uint32_t> bfeat;
  std::vector<int32_t> bnode, bstart;
  for (int f = 0; f < B; f++) {
    const vo::RefKeyFrame &K = kfs[f];
    const int nB = std::min(std::max(h_n[f], 0), cap);
    const int *node = h_node + (size_t)f * cap;
    order.resize((size_t)nB);
    for (int i = 0; i < nB; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return node[a] < node[b]; });
    bfeat.assign(order.begin(), order.end());
    bnode.clear(), bstart.clear();
    for (int i = 0; i < nB; i++)
      if (i == 0 || node[order[i]] != node[order[i - 1]]) bnode.push_back(node[order[i]]), bstart.push_back(i);
    bstart.push_back(nB);
    std::vector<uint32_t> bnode_u(bnode.begin(), bnode.end());
    const vo_bow_view bv{(int32_t)bnode.size(), bnode_u.data(), bstart.data(), bfeat.data()};
    queries.clear();
    if (K.n > 0 && nB > 0 && K.nodes)
      for_common_nodes(*K.nodes, bv, [&](int ia, int ib) {
        for (int t = K.nodes->start[ia]; t < K.nodes->start[ia + 1]; t++) {
          const int i1 = (int)K.nodes->feat[t];
          if (!K.valid[i1]) continue;  // `if (!mpk || mpk->isBad()) continue;` :475-477
          queries.push_back(i1), queries.push_back(bstart[ib]), queries.push_back(bstart[ib + 1]), queries.push_back(0);
        }
      });
    Off &o = off[f];
    o.nq = (int)queries.size() / 4, o.nB = nB;
    o.q = put(queries.data(), queries.size() * 4);
    o.bf = put(bfeat.data(), (size_t)nB * 4);
    std::vector<uint8_t> ok((size_t)std::max(nB, 1), 1);
    o.ok = put(ok.data(), ok.size());
    o.adesc = put(K.desc, (size_t)K.n * 32);
    o.aang = put(K.angle, (size_t)K.n * 4);
    o.claims = claims_bytes;
    claims_bytes += ((size_t)std::max(o.nq, 1) * 16 + 15) & ~(size_t)15;
    max_b = std::max(max_b, nB);
  }
  const size_t img_bytes = (img.size() + 255) & ~(size_t)255;
  VO_CHECK(d_img.reserve(img_bytes + claims_bytes + 256));
  uint8_t *d = d_img.as<uint8_t>();
  NodeArgs *hargs = reinterpret_cast<NodeArgs *>(img.data() + args_off);
  for (int f = 0; f < B; f++) {
    const Off &o = off[f];
    NodeArgs P{};
    P.mode = kNodeBow0, P.nq = o.nq, P.nA = kfs[f].n, P.nB = o.nB, P.check_rot = check_rot, P.ratio = ratio;
    for (int i = 0; i < 16; i++) P.sf[i] = 1.f;
    P.queries = reinterpret_cast<const int4 *>(d + o.q), P.bfeat = reinterpret_cast<const uint32_t *>(d + o.bf);
    P.descA = reinterpret_cast<const uint4 *>(d + o.adesc);
    P.descB = reinterpret_cast<const uint4 *>(fs.desc + (size_t)(slot0 + f) * cap * 32);
    P.angA = reinterpret_cast<const float *>(d + o.aang), P.angB = fs.angle + (size_t)(slot0 + f) * cap;
    P.b_ok = d + o.ok;
    P.claims = reinterpret_cast<int4 *>(d + img_bytes + o.claims);
    P.match = dev_assigned + (size_t)f * cap;
    P.n_matches = dev_n_matches + f;
    hargs[f] = P;
  }
  thread_local vo::PinnedBuf up;
  VO_CHECK(up.reserve(img.size()));
  memcpy(up.data(), img.data(), img.size());
  VO_CHECK(vo::copy_h2d(d, up.data(), img.size(), st, what));
  const size_t lds = (((size_t)max_b + 15) & ~(size_t)15) * 5;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_node_replay, hipFuncAttributeMaxDynamicSharedMemorySize, kNodeMaxB * 5);
  hipLaunchKernelGGL(k_node_replay, dim3((unsigned)B), dim3(64), lds, st, reinterpret_cast<const NodeArgs *>(d + args_off));
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::stream_sync(st, what));  // `up` is reused by the calling thread's next call
  return VO_OK;
}

