// orb.hip -- ORB extractor for gfx950 (MI355X): image pyramid, per-cell FAST-9-16 + NMS,
// oct-tree key-point distribution, intensity-centroid orientation, 7x7 Gaussian blur and steered
// BRIEF, all device-resident and batched over frames.  Replaces ORB_SLAM2::ORBextractor
// (reference src/ORBextractor.cpp); every kernel cites the lines it stands in for.
//
// Compiled with -ffp-contract=off: the float expressions that decide integer results
// (fastAtan2 polynomial, x*b + y*a sample coordinates, pt *= scale) must round exactly like the
// x86-64 reference build, which has no FMA.
#include "vo_common.h"

#include <cmath>
#include <mutex>
#include <vector>

namespace {

using namespace vo;

constexpr int kMaxLevels = 16;
constexpr int kEdge = 19;          // EDGE_THRESHOLD  ORBextractor.cpp:76
constexpr int kBorder = kEdge - 3; // minBorderX/Y    :778-779
constexpr int kHalfPatch = 15;     // HALF_PATCH_SIZE :75
constexpr int kTileP = 72;         // LDS pitch of a FAST cell tile (cell <= 66 px incl. 6 px overlap)
constexpr int kFastCpw = 4;        // cells a wavefront of k_fast_cell takes one after the other
constexpr int kMaxList = 1024;     // oct-tree node list capacity per level (quota <= kMaxList-4)

// Per-lane constants of k_describe, lane-major so that a lane fetches them with nine 16-byte loads:
//  [0..11]  the lane's 12 pixels of the 749-pixel orientation disc (filled by vo_orb_create from umax; pixel
//           lane + 64 i, padded to 768 with the centre) as offsets (v + 15) * 64 + (u + 15) in the staged window
//  [12..14] their u as signed bytes, four pixels per dword; [15..17] their v (padding entries weigh 0)
//  [18..33] steered-BRIEF pattern as floats, [word k][x0, y0, x1, y1] = test 64 k + lane
__constant__ __attribute__((aligned(16))) uint32_t c_desc_tab[64][36];
const int8_t h_pattern[1024] = {  // host copy: 256 x (x0, y0, x1, y1)
#include "orb_pattern.inc"
};

struct LevelGeom {
  int w, h, pitch;
  long long pyr_off;   // byte offset in the per-frame pyramid block (levels >= 1)
  long long blur_off;  // byte offset in the per-frame blurred block
  int nCols, nRows, wCell, hCell, maxBX, maxBY;
  int cellBase, nCells, capCell;
  long long slotBase;  // u32 offset in the per-frame cell-slot block
  int quota, capSel, selBase;
  int candCap, candBase;
  int nIni;
  float hX;
  float scale;
  int patchSize;
  int tileBase, tilesX, tilesY;
  int pad_[5];  // (formerly blur strip bookkeeping; kept so that the argument layout is unchanged)
};

struct OrbDev {
  int nlevels, ini_th, min_th, pad;
  int umax[16];
  LevelGeom lv[kMaxLevels];
};

struct FrameSrc {  // where level 0 lives (caller memory) and where levels >= 1 live (ours)
  const uint8_t *img0;
  long long img0_frame_stride;
  int img0_pitch;
  uint8_t *pyr;
  long long pyr_frame_stride;
  uint8_t *blur;
  long long blur_frame_stride;
};

// The blurred planes are read by nothing but the descriptor windows (39 x 39 pixels around a key-point), and a window row
// of a row-major plane costs a 128-byte line of its own: ~50 lines pulled through the texture path for 1.5 KB of pixels,
// which is what k_describe waits for.  They are therefore stored in TILES of 16 x 8 pixels = one
// 128-byte line each (tile rows of pitch / 16 tiles; the plane's height is padded to 8 rows): a window then touches
// 3-4 x 5-6 = 15-24 lines.  The blur kernels only change their store address; vo_orb_get_level un-tiles on the host.
__host__ __device__ __forceinline__ long long blur_tiled_off(int x, int y, int pitch) {
  return (long long)(y >> 3) * pitch * 8 + (x >> 4) * 128 + (y & 7) * 16 + (x & 15);
}

__device__ __forceinline__ const uint8_t *level_plane(const OrbDev &P, const FrameSrc &S, int l, int f,
                                                      int &pitch) {
  if (l == 0) {
    pitch = S.img0_pitch;
    return S.img0 + (long long)f * S.img0_frame_stride;
  }
  pitch = P.lv[l].pitch;
  return S.pyr + (long long)f * S.pyr_frame_stride + P.lv[l].pyr_off;
}

// ------------------------------------------------------------------------------------------
// K1  cv::resize(INTER_LINEAR, CV_8U) -- ComputePyramid, ORBextractor.cpp:1129.
// OpenCV 3.x fixed point: 11-bit coefficients (tables built on the host, orb_resize_tables()),
// horizontal pass in int32, vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.
// One thread per output pixel; source rows come through L1/L2 (each source byte is touched by
// <= 4 neighbouring threads).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(const uint8_t *src, long long s_frame_stride, int s_pitch,
                                                int sw, int sh, uint8_t *dst, long long d_frame_stride,
                                                int d_pitch, int dw, int dh, const int *xofs,
                                                const int *xab, const int *yofs, const int *yab) {
  const int dx = blockIdx.x * 64 + threadIdx.x;
  const int dy = blockIdx.y * 4 + threadIdx.y;
  if (dx >= dw || dy >= dh) return;
  const uint8_t *S = src + (long long)blockIdx.z * s_frame_stride;
  const int sx = xofs[dx];
  const int sx1 = min(sx + 1, sw - 1);
  const int ab = xab[dx];
  const int a0 = (short)(ab & 0xffff), a1 = ab >> 16;
  const int sy = yofs[dy];
  const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
  const int bb = yab[dy];
  const int b0 = (short)(bb & 0xffff), b1 = bb >> 16;
  const uint8_t *R0 = S + (long long)sy0 * s_pitch, *R1 = S + (long long)sy1 * s_pitch;
  const int r0 = R0[sx] * a0 + R0[sx1] * a1;
  const int r1 = R1[sx] * a0 + R1[sx1] * a1;
  int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
  v = min(max(v, 0), 255);
  dst[(long long)blockIdx.z * d_frame_stride + (long long)dy * d_pitch + dx] = (uint8_t)v;
}

// Tiled form used when source rows are 4-byte aligned.  A 256-thread workgroup walks a strip of
// 256 x 16 output tiles: the source rectangle of a tile (about 309 x 21 bytes at scale 1.2) travels
// HBM -> registers (coalesced dwords, all in flight) while the previous tile is consumed from LDS,
// then registers -> LDS; every thread makes 4 adjacent pixels of 4 rows from LDS.  The kernel is
// instruction-bound, so it is written branch-free: loads use clamped rows instead of predicates,
// tile copies run under a uniform trip count, outputs leave as whole dwords (row padding absorbs
// the tail).  The 4 outputs draw on at most 8 consecutive source bytes starting at column sx[0]
// (scale < 2): two v_alignbyte_b32 build that window from three aligned dwords, one v_perm_b32 per
// output spreads its two taps into 16-bit lanes (selector precomputed per column, shared by all
// rows) and v_dot2_i32_i16 against the packed coefficient pair (exactly the table word:
// a0 | a1 << 16) gives tap0*a0 + tap1*a1 in one instruction.
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2_i16(unsigned taps, unsigned coef) {
  // (the compiler emits v_mov_b32 0 + the two-operand v_dot2c_i32_i16 for the zero accumulator; the three-operand form with
  //  an inline 0 saves the 32 moves per thread and not a microsecond: 0.496 ms either way, round 4)
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, taps), __builtin_bit_cast(short2_t, coef), 0, false);
}

constexpr int kOctWaves = 7;  // waves per SIMD the oct-tree is compiled for (6: +7 %, 8: +2.5 %, round 4)
__device__ __forceinline__ int rz_h16(int h) { return h & ~15; }  // horizontal sum with the low four bits dropped (see the vertical blend)
constexpr int kRzL = 16;               // lanes (four-pixel groups) per tile row
constexpr int kRzF = 64 / kRzL;        // frames a wavefront works on side by side
constexpr int kRzW = 4 * kRzL, kRzH = 16;  // output tile of k_resize4: 64 x 16 pixels of kRzF frames
constexpr int kRzMaxDw = 32;           // dwords per source row of a tile (scale factors up to ~1.8)
constexpr int kRzPitch = 4 * kRzMaxDw;  // LDS row pitch (128 bytes: eight 16-byte chunks, what a lane group of the LDS-DMA staging fills)
constexpr int kRzMaxRows = 32;         // source rows of a tile

// One workgroup per 64 x 16 output tile of four consecutive frames; lane = frame * 16 + four-pixel group: a
// 256-pixel-wide tile wastes up to half of its lanes on the narrow levels, a 64-pixel one at most a fifth, and
// with four frames side by side the row bookkeeping (source rows, vertical coefficients) stays wave-uniform.
// Everything that depends only on the output column is precomputed per level with the handle (`gtab`: per
// four-pixel group the aligned source column, the byte offset of its 8-byte window, the four tap selectors and
// coefficient pairs; `btab`: per column block the first source dword and the dwords per row), and the source
// rectangles are staged by LDS-DMA (round 6; buffer_load_dwordx4 ... lds, dword-aligned source, hardware range check):
// LDS row 8 Q + s (128 bytes) = source row 2 Q + (s >> 2) of frame s & 3, one wave-instruction per group Q of eight
// stacked rows, lane = 8 s + 16-byte chunk -- no registers, no LDS stores (the round-2 form, a dword per lane through
// registers into rows of 132 bytes: 0.492 against 0.477 ms per 1024 frames).
// One tile per workgroup measured fastest (0.196 ms per 256 frames against 0.214 / 0.221 / 0.240 with 2 / 4 /
// 8 vertically consecutive tiles software-pipelined in one workgroup): many small workgroups overlap their
// loads and arithmetic across each other better than an in-kernel pipeline does.
__global__ __launch_bounds__(256) void k_resize4(const uint8_t *src, long long s_frame_stride, int s_pitch,
                                                 int sw, int sh, uint8_t *dst, long long d_frame_stride,
                                                 int d_pitch, int dw, int dh, const int *__restrict__ gtab,
                                                 const int *__restrict__ btab, const int *__restrict__ yofs,
                                                 const int *__restrict__ yab, int n_frames) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rz_tile[];
  constexpr int R = 4;
  const int tid = threadIdx.x, gx = tid & (kRzL - 1), fo = (tid & 63) / kRzL;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dx0 = blockIdx.x * kRzW;
  const int dx = min(dx0 + 4 * gx, ((dw - 1) & ~3));  // lanes past the row repeat its last group
  const int f0 = blockIdx.z * kRzF, nfr = min(kRzF, n_frames - f0);  // frames of this workgroup (uniform)
  const uint8_t *S = src + (long long)f0 * s_frame_stride;
  uint8_t *Dst = dst + (long long)(f0 + min(fo, nfr - 1)) * d_frame_stride;  // lanes past the batch repeat its last frame
  const bool store_ok = fo < nfr;
  // column constants of this thread's group
  const int4 *gt = reinterpret_cast<const int4 *>(gtab + 12 * (dx >> 2));
  const int4 g0v = gt[0], g1v = gt[1], g2v = gt[2];
  const int base = g0v.x;                 // aligned source column: three dwords from here hold the group's 8-byte window
  const unsigned woff = (unsigned)g0v.y;  // byte offset of the window in {w2,w1,w0}
  const unsigned sel[4] = {(unsigned)g0v.z, (unsigned)g0v.w, (unsigned)g1v.x, (unsigned)g1v.y};
  const unsigned ab[4] = {(unsigned)g1v.z, (unsigned)g1v.w, (unsigned)g2v.x, (unsigned)g2v.y};
  const int c0 = btab[2 * blockIdx.x], ndw = btab[2 * blockIdx.x + 1];  // first source column (aligned), dwords per row
  const int lastT = c0 + 4 * (ndw - 1);
  // o1/o2 are only clamped when base+4 / base+8 lie beyond the strip's last dword, and then every
  // column this thread needs sits in an earlier dword, so the bytes it selects stay valid
  const int lf = fo * kRzPitch;  // the lane's frame within a stacked LDS row group
  const int o0 = base - c0 + lf, o1 = min(base + 4, lastT) - c0 + lf, o2 = min(base + 8, lastT) - c0 + lf;
  // source rows of the tile
  const int dy0 = blockIdx.y * kRzH;
  const int r0 = min(max(yofs[dy0], 0), sh - 1);
  const int r1 = min(max(yofs[min(dy0 + kRzH, dh) - 1] + 1, 0), sh - 1);
  const int nq = ((r1 - r0 + 1) * kRzF + 7) >> 3;  // staging rounds (uniform)
  int sy_[R], bb_[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int dy = min(dy0 + wave * R + r, dh - 1);  // uniform per wave: scalar loads
    sy_[r] = yofs[dy];
    bb_[r] = yab[dy];
  }
  {
    // staging by LDS-DMA: lane = 8 x row slot + 16-byte chunk; a wave-instruction fills the 8 stacked rows 8 Q .. 8 Q + 7 (source rows
    // 2 Q and 2 Q + 1 of the four frames) at 128 bytes each -- the same stacking as above with the dwords of a chunk fetched together,
    // no registers, no LDS stores, two vector instructions of address arithmetic per 1 KB
    const int lane = tid & 63, rs = lane >> 3, c = lane & 7;
    const int nch = (ndw + 3) >> 2, ymax = sh - 1 - r0;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void *)S, 0, (int)((nfr - 1) * s_frame_stride) + sh * s_pitch, 0x00020000);
    // (the four frames' rows of a source row sit 128 bytes apart, i.e. in the same banks; an xor swizzle of the chunk position by the
    //  frame that spreads them over the 32 banks changed nothing: 0.477-0.479 ms either way -- the kernel does not wait for LDS)
    const int lane_off = min(rs & 3, nfr - 1) * (int)s_frame_stride + r0 * s_pitch + c0 + 16 * min(c, nch - 1);
    const bool hi = (rs >> 2) != 0;
    for (int Q = wave; Q < nq; Q += 4) {  // uniform per wavefront
      const int ra = __mul24(min(2 * Q, ymax), s_pitch), rb = __mul24(min(2 * Q + 1, ymax), s_pitch);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void *)((__attribute__((address_space(3))) uint8_t *)rz_tile + Q * 8 * kRzPitch), 16,
                                               lane_off + (hi ? rb : ra), 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int dyw = dy0 + wave * R;
  uint8_t *orow = Dst + (long long)dyw * d_pitch + dx;
  // Round 4: (i) consecutive output rows share a source row four times out of five at scale 1.2 (row r + 1's upper row
  // is row r's lower one: the row indices are wave-uniform, so the test is a scalar branch) -- its horizontal sums are
  // kept instead of being read and formed again; (ii) no clamp: the weights are non-negative and sum to 2048 in both
  // directions, so the value is <= (255 * 4 + 2) >> 2 = 255 by construction.  23.6 -> ~18 instructions per output pixel.
  int hp[4] = {0, 0, 0, 0}, prev_row = -1;
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (dyw + r >= dh) break;  // uniform per wave
    const int sy0 = min(max(sy_[r], 0), sh - 1) - r0, sy1 = min(max(sy_[r] + 1, 0), sh - 1) - r0;
    const int b0 = (short)(bb_[r] & 0xffff), b1 = bb_[r] >> 16;
    int g0[4], g1[4];  // horizontal sums >> 4 of the upper / lower source row
    if (sy0 == prev_row) {  // uniform
#pragma unroll
      for (int q = 0; q < 4; q++) g0[q] = hp[q];
    } else {
      const uint8_t *R0 = rz_tile + __mul24(sy0 * kRzF, kRzPitch);
      const unsigned p0 = *reinterpret_cast<const unsigned *>(R0 + o0), p1 = *reinterpret_cast<const unsigned *>(R0 + o1),
                     p2 = *reinterpret_cast<const unsigned *>(R0 + o2);
      const unsigned pl = __builtin_amdgcn_alignbyte(p1, p0, woff), ph = __builtin_amdgcn_alignbyte(p2, p1, woff);
#pragma unroll
      for (int q = 0; q < 4; q++) g0[q] = rz_h16(dot2_i16(__builtin_amdgcn_perm(ph, pl, sel[q]), ab[q]));
    }
    {
      const uint8_t *R1 = rz_tile + __mul24(sy1 * kRzF, kRzPitch);
      const unsigned q0 = *reinterpret_cast<const unsigned *>(R1 + o0), q1 = *reinterpret_cast<const unsigned *>(R1 + o1),
                     q2 = *reinterpret_cast<const unsigned *>(R1 + o2);
      const unsigned ql = __builtin_amdgcn_alignbyte(q1, q0, woff), qh = __builtin_amdgcn_alignbyte(q2, q1, woff);
#pragma unroll
      for (int q = 0; q < 4; q++) g1[q] = rz_h16(dot2_i16(__builtin_amdgcn_perm(qh, ql, sel[q]), ab[q]));
    }
    unsigned outw = 0;
    // (b * (h >> 4)) >> 16 in ONE instruction: v_mul_hi_u32_u24(b << 12, h & ~15) = (b 2^12 (h >> 4) 2^4) >> 32; both factors
    // are non-negative and below 2^24 (b <= 2048, h <= 255 * 2048), so the 24-bit form is exact.  g0 / g1 hold h & ~15.
    {
      const unsigned b0s = (unsigned)b0 << 12, b1s = (unsigned)b1 << 12;  // uniform
      unsigned v4[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        unsigned t0, t1;
        asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(t0) : "s"(b0s), "v"(g0[q]));
        asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(t1) : "s"(b1s), "v"(g1[q]));
        v4[q] = (t0 + t1 + 2u) >> 2;
        hp[q] = g1[q];
      }
      outw = v4[0] | (v4[1] << 8) | (v4[2] << 16) | (v4[3] << 24);
    }
    prev_row = sy1;
    if (store_ok) *reinterpret_cast<unsigned *>(orow + (long long)r * d_pitch) = outw;  // the tail lands in the row padding
  }
}

// ------------------------------------------------------------------------------------------
// K2  cv::FAST(cell, threshold, nms=true) per 30-px grid cell with the 20 -> 7 threshold fallback
// -- ComputeKeyPointsOctTree cell loop, ORBextractor.cpp:796-837.
// One WAVEFRONT per (cell, frame), four cells per 256-thread workgroup, no workgroup barrier
// anywhere: the kernel is instruction-issue bound, and ballot/mbcnt compaction inside one wave
// needs neither LDS counters nor barriers.  The cell sub-image (incl. the 6-px overlap) is staged
// in the wave's LDS region.  With S = max over the 16 nine-pixel arcs of the minimum |centre - ring|
// (bright or dark), "corner at threshold t" <=> S > t and the OpenCV cornerScore is S-1, independent of
// t.  S is only evaluated for pixels whose compass margin (an upper bound of S) exceeds the threshold in
// force.  NMS runs on the LDS score tile with entries below the cell's current threshold read as 0 and
// pixels outside the cell interior as 0 (Q-E2).  Survivors
// are written in raster order to the cell's slot: x | y<<12 | score<<24 with x,y already shifted
// by (j*wCell, i*hCell) like :830-831.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {  // LDS hand-off between lanes of one wavefront
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;  // explicit LDS pointers: 32-bit address arithmetic
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) uint8_t gmem_u8;  // global memory: loads are scalar base + lane offset

// Arc score of one pixel.  With A = min over the 16 nine-pixel arcs of the arc's maximum and B = max over
// the arcs of the arc's minimum (ring values, not differences), S = max(v - A, B - v); a nine-arc extremum
// is a 3-extremum of 3-extrema.  `ring` points at the tile pixel (-3, -3) from the centre so that every
// LDS offset is a non-negative immediate.  RP: row pitch of the tile in LDS.
template <int RP>
__device__ __forceinline__ int fast_arc_score(const lds_u8 *ring, int min_th) {
  constexpr int C = 3 * RP + 3;
  const int v = ring[C];
  int r[16];
  r[0] = ring[C + 3 * RP];
  r[1] = ring[C + 3 * RP + 1];
  r[2] = ring[C + 2 * RP + 2];
  r[3] = ring[C + RP + 3];
  r[4] = ring[C + 3];
  r[5] = ring[C - RP + 3];
  r[6] = ring[C - 2 * RP + 2];
  r[7] = ring[C - 3 * RP + 1];
  r[8] = ring[C - 3 * RP];
  r[9] = ring[C - 3 * RP - 1];
  r[10] = ring[C - 2 * RP - 2];
  r[11] = ring[C - RP - 3];
  r[12] = ring[C - 3];
  r[13] = ring[C + RP - 3];
  r[14] = ring[C + 2 * RP - 2];
  r[15] = ring[C + 3 * RP - 1];
  // three-extrema of consecutive ring pixels on the 16-bit VOP2 min / max (the class that issues at ~2.2 cycles, section 7
  // of DESIGN.md; the compiler's own v_min_u32 / v_max_u32 issue at 4.2): four ring positions per asm block, 12 operations
  unsigned mn3[16], mx3[16];
#pragma unroll
  for (int k = 0; k < 16; k += 4) {
    unsigned t0, t1;
    asm("v_min_u16 %8, %11, %12\n\t"
        "v_min_u16 %9, %13, %14\n\t"
        "v_min_u16 %0, %10, %8\n\t"
        "v_min_u16 %1, %8, %13\n\t"
        "v_min_u16 %2, %12, %9\n\t"
        "v_min_u16 %3, %9, %15\n\t"
        "v_max_u16 %8, %11, %12\n\t"
        "v_max_u16 %9, %13, %14\n\t"
        "v_max_u16 %4, %10, %8\n\t"
        "v_max_u16 %5, %8, %13\n\t"
        "v_max_u16 %6, %12, %9\n\t"
        "v_max_u16 %7, %9, %15"
        : "=&v"(mn3[k]), "=&v"(mn3[k + 1]), "=&v"(mn3[k + 2]), "=&v"(mn3[k + 3]), "=&v"(mx3[k]), "=&v"(mx3[k + 1]),
          "=&v"(mx3[k + 2]), "=&v"(mx3[k + 3]), "=&v"(t0), "=&v"(t1)
        : "v"((unsigned)r[k]), "v"((unsigned)r[(k + 1) & 15]), "v"((unsigned)r[(k + 2) & 15]), "v"((unsigned)r[(k + 3) & 15]),
          "v"((unsigned)r[(k + 4) & 15]), "v"((unsigned)r[(k + 5) & 15]));
  }
  unsigned Au = max(mx3[0], max(mx3[3], mx3[6])), Bu = min(mn3[0], min(mn3[3], mn3[6]));
#pragma unroll
  for (int k = 1; k < 16; k++) {
    Au = min(Au, max(mx3[k], max(mx3[(k + 3) & 15], mx3[(k + 6) & 15])));
    Bu = max(Bu, min(mn3[k], min(mn3[(k + 3) & 15], mn3[(k + 6) & 15])));
  }
  const int A = (int)Au, B = (int)Bu;
  const int S = max(v - A, B - v);
  return S > min_th ? S - 1 : 0;
}

__host__ __device__ __forceinline__ int fast_align16(int v) { return (v + 15) & ~15; }
// ------------------------------------------------------------------------------------------
// K2  k_fast_cell: per-cell FAST-9-16 + NMS, one wavefront per (cell, frame), four cells per workgroup, no
// workgroup barrier.  Round 3 rebuilt round 2's k_fast_wave around what the issue-rate calibration of this chip shows
// (profiles/r03_valu_issue_calibration.txt): 32-bit min / max, three-operand and packed integer forms issue at
// 4.2 cycles per wave64, 16-bit VOP2 forms at 2.2 (>= 4 waves per SIMD) -- and around the LDS pipe, which the
// round-2 kernel kept 64 % busy:
//  * the tile goes global memory -> LDS by LDS-DMA (global_load_lds_dwordx4, lane = 16-byte chunk of a 48-byte
//    tile row): two or three instructions per cell instead of 12 loads + 12 LDS stores and their addressing;
//  * the compass-margin pass runs on 16-bit VOP2 min / max / sub and does NOT leave its bound in the score
//    tile: the tile is zero except for the arc scores of the survivors, which is all the NMS needs (a pixel
//    whose margin does not exceed the threshold cannot score above it); the rare second round at minThFAST and
//    the rare overflow path recompute the margins instead of reading them back;
// Results are bit-identical to round 2's kernel and to the oracle (tests/test_gpu_orb.py); 0.253 -> 0.210 ms per 256
// frames.  Early-exit builds (1024 frames, round 4: 0.72 ms) return after staging at 0.275, after the first margin walk
// at 0.45, after the arc scores at 0.65; the minThFAST round costs 0.05 -- the phases are ADDITIVE (0.275 + 0.17 + 0.20 +
// 0.07 = 0.715).  What the first 0.275 is was settled in round 5 (profiles/r05_ab_fused.txt): a wave that takes 2 / 4 / 8
// cells in a PLAIN loop (no register prefetch: 63 registers, still 8 waves per SIMD) runs at 0.70 / 0.685 / 0.71 ms -- the
// dispatch of 835 k one-cell waves is worth 5 %, and 4 cells per wave is what the kernel does now (kFastCpw).  The rest of
// the 0.275 is the latency of the tile's LDS-DMA as seen by a kernel that does nothing else: the fused level pass, which
// stages 2.2x larger tiles with a quarter of the workgroups, measures the same 0.45 ms for "staging only" -- ~8 workgroups
// per CU each waiting ~3 us for its tile.  In the full kernel that wait is overlapped by the other resident waves' walks
// and scores; the multi-cell forms of round 4 that tried to hide it by a register prefetch paid more in occupancy (71
// registers, 7 waves: 0.76 ms) than there was to win.
// ------------------------------------------------------------------------------------------
#define VO_OP16(name, ins)                                                   \
  __device__ __forceinline__ unsigned name(unsigned a, unsigned b) {          \
    unsigned r;                                                               \
    asm(ins " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                        \
    return r;                                                                 \
  }
VO_OP16(max_u16, "v_max_u16")
VO_OP16(min_u16, "v_min_u16")
#undef VO_OP16

// LDS bytes of one wave of k_fast_cell: image tile, score tile (same geometry), survivor list
__host__ __device__ __forceinline__ int fast_cell_lds(int tp, int tile_rows, int list_cap) {
  return 2 * fast_align16(tp * tile_rows) + fast_align16(2 * list_cap);
}

// FAST-9-16 + NMS of ONE cell by ONE wavefront on a tile that is already in LDS (k_fast_cell stages a private tile per
// wave, k_level_pass a tile shared by the workgroup).  `tile`: the cell's pixel (iniX, iniY), row pitch TP; the score
// entry of a pixel lives `sdelta` bytes behind its image byte (a zeroed region of the same pitch whose entries around
// the cell's interior are never written); `plist`: the wave's survivor list.  Everything else as described above.
template <int TP>
__device__ __forceinline__ void fast_cell_body(lds_u8 *tile, int sdelta, lds_u16 *plist, int list_cap, int iw, int ih,
                                               int ini_th, int min_th, uint32_t *slot, int cap_cell, int xoff, int yoff,
                                               int *out_count, int lane) {
  // ---- margin walk.  A 9-arc always contains two neighbouring compass pixels (ring 0, 4, 8, 12), so the arc
  // score S is bounded by the compass margin m = max(mb - v, v - md), mb = min(max(p0,p8), max(p4,p12)),
  // md = max(min(p0,p8), min(p4,p12)): a pixel can only be a corner at threshold t if m > t.  Pixels with m > th
  // are compacted IN RASTER ORDER into the wave's LDS list as the LDS address of their (-3, -3) neighbour.
  // Lane mapping: 32 columns x 2 rows per step (64 x 1 for cells wider than 32).
  const int lw = iw <= 32 ? 5 : 6;  // uniform
  const int lx = lane & ((1 << lw) - 1), ly = lane >> lw, ri = 64 >> lw;
  lds_u8 *const b0 = tile + ly * TP + lx;
  struct Ring5 { unsigned v, p0, p4, p8, p12; };
  auto load5 = [&](const lds_u8 *b) { return Ring5{b[3 * TP + 3], b[6 * TP + 3], b[3 * TP + 6], b[3], b[3 * TP]}; };
  // lanes with m > th as a wave mask: nine 16-bit VOP2 instructions and the compare, in one block (no register
  // moves, no hazard padding between the dependent instructions: none of them writes a partial register)
  auto margin_gt = [&](const Ring5 &r, unsigned thv) {
    unsigned long long mask;
    unsigned t0, t1, t2, t3;
    asm("v_max_u16 %1, %6, %8\n\t"
        "v_max_u16 %2, %7, %9\n\t"
        "v_min_u16 %3, %6, %8\n\t"
        "v_min_u16 %4, %7, %9\n\t"
        "v_min_u16 %1, %1, %2\n\t"
        "v_max_u16 %3, %3, %4\n\t"
        "v_sub_u16 %1, %1, %5\n\t"
        "v_sub_u16 %3, %5, %3\n\t"
        "v_max_i16 %1, %1, %3\n\t"
        "v_cmp_gt_i16 %0, %1, %10"
        : "=s"(mask), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(r.v), "v"(r.p0), "v"(r.p4), "v"(r.p8), "v"(r.p12), "v"(thv));
    return mask;
  };
  // rows [r0, r1) of the interior; returns the number of survivors (uniform); entries past the list's capacity
  // pile up on its last 64 slots and the count tells the caller to take the chunked path
  auto walk = [&](int th, int r0, int r1) {
    int n = 0;
    const unsigned thv = (unsigned)th;
    if (lx < iw) {
      lds_u8 *b = b0 + r0 * TP;
      asm("" : "+v"(b));
      int row = r0;
      auto append = [&](unsigned long long mask, lds_u8 *bb) {
        const int base = min(n, list_cap - 64);  // scalar
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)base));
        if (__builtin_amdgcn_inverse_ballot_w64(mask)) plist[pos] = (unsigned short)(unsigned)(uintptr_t)bb;
        n += __popcll(mask);
      };
      for (; row + 2 * ri <= r1; row += 2 * ri, b += 2 * ri * TP) {
        lds_u8 *b1 = b + ri * TP;
        const Ring5 va = load5(b), vb = load5(b1);
        const unsigned long long ma = margin_gt(va, thv), mb = margin_gt(vb, thv);
        append(ma, b);
        append(mb, b1);
      }
      if (row + ri <= r1) {
        append(margin_gt(load5(b), thv), b);
        row += ri, b += ri * TP;
      }
      if (row < r1) {  // odd last row of a two-row step: the lanes of the first row only
        const unsigned long long m = margin_gt(load5(b), thv);
        append(ly == 0 ? m & 0xffffffffull : 0ull, b);
      }
    }
    return __builtin_amdgcn_readlane(n, 0);  // lane 0 always takes part
  };
  // Second-level bound for long lists (a cell redone at minThFAST lists ~1/5 of its pixels, noise for the most part):
  // the compass argument holds for ANY four ring pixels a quarter turn apart, so S is also bounded by the margin of the
  // rings (K, K + 4, K + 8, K + 12), K = 1, 2, 3.  Entries whose bound does not exceed th cannot be corners at th and, as
  // neighbours, score below every kept pixel (S - 1 < th): dropping them -- score entry left at 0 -- changes nothing.
  // The list is compacted in place, order preserved (a lane writes at or before its own position, after every lane of
  // the batch has read; LDS operations of one wavefront execute in order).  ~14 fast-class instructions per 64 entries
  // against ~105 for their arc scores; the second bound halves a minThFAST list, the third takes another third.
  auto refine_list = [&](int n, int th, auto load_ring) {
    int kept = 0;
    const unsigned thv = (unsigned)th;
    for (int base = 0; base < n; base += 64) {
      // lanes past the list repeat its last entry (no divergence around the asm block); their mask bits are cleared
      const unsigned short e = plist[min(base + lane, n - 1)];
      const int rem = n - base;  // uniform
      const unsigned long long valid = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
      const unsigned long long mask = margin_gt(load_ring((const lds_u8 *)(uintptr_t)(unsigned)e), thv) & valid;
      const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)kept));
      if (__builtin_amdgcn_inverse_ballot_w64(mask)) plist[pos] = e;
      kept += __popcll(mask);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS operations of one wavefront execute in order: the next batch
      __builtin_amdgcn_wave_barrier();                    // reads behind these writes (no memory fence needed)
    }
    return kept;
  };
  auto ring_k2 = [&](const lds_u8 *b) { return Ring5{b[3 * TP + 3], b[5 * TP + 5], b[TP + 5], b[TP + 1], b[5 * TP + 1]}; };
  auto ring_k1 = [&](const lds_u8 *b) { return Ring5{b[3 * TP + 3], b[6 * TP + 4], b[2 * TP + 6], b[2], b[4 * TP]}; };
  auto ring_k3 = [&](const lds_u8 *b) { return Ring5{b[3 * TP + 3], b[4 * TP + 6], b[4], b[2 * TP], b[6 * TP + 2]}; };
  auto score_list = [&](int n) {
    for (int i = lane; i < n; i += 64) {
      lds_u8 *b = (lds_u8 *)(uintptr_t)(unsigned)plist[i];
      b[sdelta + 3 * TP + 3] = (uint8_t)fast_arc_score<TP>(b, min_th);  // cornerScore, defined from minThFAST up
    }
  };
  // 3x3 non-maximum suppression at threshold th over the raster-ordered list; kept pixels go out in the same
  // order behind the `kept` already written.  Entries that are not survivors are 0, as OpenCV reads them.
  auto nms_list = [&](int n, int th, int kept) {
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      bool keep = false;
      int pos = 0, sc0 = 0;
      if (i < n) {
        const lds_u8 *b = (const lds_u8 *)(uintptr_t)(unsigned)plist[i];
        pos = (int)(b - tile);
        const lds_u8 *c = b + sdelta + 2 * TP + 2;  // score entry (-1, -1) from the pixel
        sc0 = c[TP + 1];
        const unsigned nb = max_u16(max_u16(max_u16(c[0], c[1]), max_u16(c[2], c[TP])),
                                    max_u16(max_u16(c[TP + 2], c[2 * TP]), max_u16(c[2 * TP + 1], c[2 * TP + 2])));
        keep = sc0 >= th && sc0 > (int)nb;
      }
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
      const int off = kept + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      if (keep && off < cap_cell) {
        const int y = pos / TP, x = pos - y * TP;  // interior coordinates; the cell-local ones are +3
        slot[off] = (uint32_t)(x + 3 + xoff) | ((uint32_t)(y + 3 + yoff) << 12) | ((uint32_t)sc0 << 24);
      }
      kept += __popcll(mask);
    }
    return kept;
  };
  int running = 0;
  for (int round = 0; round < 2; round++) {
    // a cell with no key-point at iniThFAST is redone at minThFAST (:820-824)
    const int th = round == 0 ? ini_th : min_th;
    const int np0 = walk(th, 0, ih);
    wave_sync();
    int np = np0;
    if (np <= list_cap - 64) {  // uniform, the usual case (the walk's appends clamp their base 64 entries before the end)
      // (only in the minThFAST round: at iniThFAST the further bounds reject ~15 % of a list -- three passes for nothing,
      //  measured +3 % on the kernel -- while they take a minThFAST list from ~170 entries to ~55)
      if (round == 1) {
        if (np > 64) np = refine_list(np, th, ring_k2);
        if (np > 64) np = refine_list(np, th, ring_k1);
        if (np > 64) np = refine_list(np, th, ring_k3);
      }
      score_list(np);
      wave_sync();
      running = nms_list(np, th, 0);
    } else {
      // Rare: more survivors than the list holds (noise-like texture).  Chunks of as many rows as fit the list
      // even if every pixel hits are scored one after the other, then -- all scores final -- suppressed one after
      // the other; chunks follow each other in raster order.
      const int chunk_rows = max((list_cap - 64) / 64, 1) * ri;
      for (int r0 = 0; r0 < ih; r0 += chunk_rows) {
        const int n = walk(th, r0, min(r0 + chunk_rows, ih));
        wave_sync();
        score_list(n);
        wave_sync();
      }
      running = 0;
      for (int r0 = 0; r0 < ih; r0 += chunk_rows) {
        const int n = walk(th, r0, min(r0 + chunk_rows, ih));
        wave_sync();
        running = nms_list(n, th, running);
        wave_sync();
      }
    }
    if (running > 0) break;  // :820 `if(vKeysCell.empty())` retry with minThFAST
    wave_sync();
  }
  if (lane == 0) *out_count = min(running, cap_cell);
}

// TP: tile row pitch in bytes (48: three 16-byte chunks, the LDS-DMA path; 72: the largest legal cell, staged
// through registers).  BYTEWISE: the caller's level-0 rows are not 4-byte aligned (copied byte by byte).
template <int TP, bool BYTEWISE>
__attribute__((amdgpu_waves_per_eu(8, 8))) __global__ __launch_bounds__(256) void k_fast_cell(OrbDev P, FrameSrc src, uint32_t *cell_slots,
                                                   long long slots_frame_stride, int *cell_count,
                                                   int cells_per_frame, int tile_rows, int list_cap,
                                                   const int *__restrict__ cell_tab, int cell_begin, int cell_end) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fast_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int f = blockIdx.y;
  // A wave takes kFastCpw consecutive cells in a plain loop (round 5, profiles/r05_ab_fused.txt: 0.72 -> 0.70 / 0.685 / 0.71 ms
  // per 1024 frames at 2 / 4 / 8 cells per wave -- the dispatch of 835 k one-cell waves was worth 5 %, not the 38 % an
  // early-exit build of the one-cell kernel suggested).
  for (int cpw = 0; cpw < kFastCpw; cpw++) {
  const int cell = cell_begin + (blockIdx.x * 4 + wave) * kFastCpw + cpw;  // (a level's cells, or all of them)
  if (cell >= cell_end) return;  // wave-uniform; the kernel has no workgroup barrier
  if (cpw) wave_sync();
  const int tile_bytes = fast_align16(TP * tile_rows);  // uniform
  lds_u8 *tile_raw = (lds_u8 *)fast_lds + wave * fast_cell_lds(TP, tile_rows, list_cap);
  lds_u16 *plist = (lds_u16 *)(tile_raw + 2 * tile_bytes);
  const int *cd = cell_tab + 16 * cell;
  const int l = cd[0], iniX = cd[1], iniY = cd[2], cw = cd[3], ch = cd[4], xoff = cd[5], yoff = cd[6];
  const int slot_off = cd[7], cap_cell = cd[8];
  int *out_count = cell_count + (long long)f * cells_per_frame + cell;
  const int iw = cw - 6, ih = ch - 6;
  if (iw <= 0 || ih <= 0) {  // :801, :811 (cw = 0 in the table), or no interior pixel
    if (lane == 0) *out_count = 0;
    continue;
  }
  int pitch;
  const uint8_t *img;
  if (l == 0) {
    pitch = src.img0_pitch;
    img = src.img0 + (long long)f * src.img0_frame_stride;
  } else {
    pitch = cd[9];
    img = src.pyr + (long long)f * src.pyr_frame_stride + (((long long)cd[11] << 32) | (unsigned)cd[10]);
  }
  // ---- stage the cell (incl. the 6-px overlap).  The tile keeps the source's dword alignment: LDS column 0 is
  // image column iniX - (iniX & 3).
  const bool bytewise = BYTEWISE && l == 0;
  const int ox = bytewise ? 0 : (iniX & 3);
  lds_u8 *tile = tile_raw + ox;
  const uint8_t *g0 = img + (long long)iniY * pitch + (iniX - ox);
  if (!bytewise && TP == 48) {
    // LDS-DMA: lane + 64 k <-> 16-byte chunk (row = idx / 3, chunk = idx % 3), LDS address 16 idx.  A row's 48
    // bytes may reach past the cell (into the next cell's pixels or the row padding; the plane has at least 13
    // more rows behind the last tile row, so never past the allocation).
    gmem_u8 *gb = (gmem_u8 *)g0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int idx = lane + 64 * k;
      const unsigned row = (unsigned)idx / 3u, chunk = (unsigned)idx - 3u * row;
      if (64 * k < 3 * ch && idx < 3 * ch)
        __builtin_amdgcn_global_load_lds(gb + (row * (unsigned)pitch + 16u * chunk), tile_raw + 1024 * k, 16, 0, 0);
    }
  } else if (!bytewise) {
    constexpr int LW = TP <= 64 ? 16 : 32, RG = 64 / LW, NB = 6;  // dword columns, rows per group, groups per batch
    const int npr = (ox + cw + 3) >> 2;
    const int d = lane & (LW - 1), r = lane / LW;
    if (d < npr) {
      const unsigned goff = (unsigned)(r * pitch + 4 * d);
      lds_u8 *lt = tile_raw + r * TP + 4 * d;
      const int ng = (ch + RG - 1) / RG, last0 = max(ch - RG, 0);
      for (int gb = 0; gb < ng; gb += NB) {
        uint32_t v[NB];
        int row0[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) {
          row0[q] = min(min(gb + q, ng - 1) * RG, last0);
          long long ro = (long long)row0[q] * pitch;
          asm("" : "+s"(ro));
          v[q] = *reinterpret_cast<const uint32_t *>(g0 + ro + goff);
        }
#pragma unroll
        for (int q = 0; q < NB; q++) {
          int lo = row0[q] * TP;
          asm("" : "+s"(lo));
          *(__attribute__((address_space(3))) uint32_t *)(lt + lo) = v[q];
        }
      }
    }
  } else {
    constexpr int NB = 8;
    for (int col = lane; col < cw; col += 64) {
      for (int rb = 0; rb < ch; rb += NB) {
        uint8_t v[NB];
        int row[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) {
          row[q] = min(rb + q, ch - 1);
          v[q] = g0[(long long)row[q] * pitch + col];
        }
#pragma unroll
        for (int q = 0; q < NB; q++) tile_raw[row[q] * TP + col] = v[q];
      }
    }
  }
  // the score tile starts at zero (under the loads' latency): only survivors ever get an entry
  for (int i = lane; i < tile_bytes / 16; i += 64)
    *(__attribute__((address_space(3))) u32x4_t *)(tile_raw + tile_bytes + 16 * i) = u32x4_t{0u, 0u, 0u, 0u};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();
  fast_cell_body<TP>(tile, tile_bytes, plist, list_cap, iw, ih, P.ini_th, P.min_th,
                     cell_slots + (long long)f * slots_frame_stride + slot_off, cap_cell, xoff, yoff, out_count, lane);
  }
}

// ------------------------------------------------------------------------------------------
// E3/E4  ExtractorNode::DivideNode + ORBextractor::DistributeOctTree, ORBextractor.cpp:487-769.
// One 256-thread workgroup per (level, frame).  The std::list of nodes is an LDS array in list
// order; keys never move -- each key carries the list position of its node.  One pass =
//   (1) order the expandable nodes (count > 1): list order in the breadth phase (:599-657), by
//       (size desc, creation order desc) in the "careful" phase (:679-744; Q-E3 tie rule),
//   (2) count keys per child with LDS atomics, (3) prefix sums give the cut-off where the list
//       reaches N and every node's new list position (children are push_front'ed, so later-
//       processed parents come first and n4,n3,n2,n1 within a parent), (4) relabel keys.
// The best-response key of each final node (first wins on ties, :748-766) is an atomicMax over
// (response, -candidate index).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan(int v, int *wsum /*LDS[5]*/, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  __syncthreads();  // protect wsum reuse
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < 4; w++) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return base + x - v;
}

template <int CAP>  // node-list capacity: 256 when every level's quota fits (16 KB of LDS), else 1024
struct OctLds {
  unsigned short x0[2][CAP], y0[2][CAP], x1[2][CAP], y1[2][CAP];
  unsigned short cnt[2][CAP], seq[2][CAP];
  unsigned short prank[CAP];   // processing rank of an expandable node, 0xffff otherwise
  unsigned short order[CAP];   // rank -> list position
  unsigned short newpos_old[CAP];
  unsigned short newpos_child[CAP * 4];
  int ccount[CAP * 4];
  int cprefix[CAP];            // inclusive prefix over ranks of nonempty-children counts
  unsigned int best[CAP];
  int wsum[8];
  int s_m, s_cutoff, s_newsize, s_nexp, s_total;
};

__device__ __forceinline__ int quadrant_of(int kx, int ky, int x0, int y0, int x1, int y1) {
  const int midx = x0 + ((x1 - x0 + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)   :489
  const int midy = y0 + ((y1 - y0 + 1) >> 1);
  return (kx < midx ? 0 : 1) + (ky < midy ? 0 : 2);  // n1,n2,n3,n4            :522-534
}

// (7 waves per SIMD = 7 workgroups per CU: the kernel is a chain of latencies, co-resident workgroups are its
// throughput; 0.326 -> 0.283 ms per 1024 frames against the compiler's own choice of 88 registers / 5 waves)
template <int CAP>
__attribute__((amdgpu_waves_per_eu(kOctWaves, kOctWaves))) __global__ __launch_bounds__(256) void k_octree(OrbDev P, const uint32_t *cell_slots,
                                                long long slots_frame_stride, const int *cell_count,
                                                int cells_per_frame, uint32_t *key_data,
                                                unsigned short *key_label, int keys_per_frame,
                                                int *cand_count, uint32_t *sel, int sel_per_frame,
                                                int *nk, int *err_flag) {
  __shared__ OctLds<CAP> S;
  const int tid = threadIdx.x;
  const int l = blockIdx.y, f = blockIdx.x;  // every frame's level 0 first: the longest workgroups start first, the short ones fill the tail
  const LevelGeom &L = P.lv[l];
  uint32_t *kd = key_data + (long long)f * keys_per_frame + L.candBase;
  unsigned short *kl = key_label + (long long)f * keys_per_frame + L.candBase;
  const int *cc = cell_count + (long long)f * cells_per_frame + L.cellBase;
  const uint32_t *slots = cell_slots + (long long)f * slots_frame_stride + L.slotBase;
  const int N = L.quota;

  // ---- gather the per-cell lists in reference order (cell row-major, raster inside a cell).  The cells'
  // exclusive offsets go to LDS (the child counters double as scratch); then thread t takes candidates t,
  // t + 256, ...: a binary search finds the cell of each, so all of a thread's slot loads are independent and
  // in flight together (walking a cell's list entry by entry costs a global round trip per entry), and the
  // first 8 stay in registers for the passes below.
  constexpr int KR = 8;
  uint32_t kreg[KR];
  int lreg[KR];
  int n = 0;
  bool inreg;
  if (L.nCells <= CAP * 4) {
    int *coff = S.ccount;  // [nCells]: exclusive offsets
    for (int base = 0; base < L.nCells; base += 256) {
      const int c = base + tid;
      const int cnt = c < L.nCells ? cc[c] : 0;
      int tot;
      const int off = n + block_excl_scan(cnt, S.wsum, &tot);
      if (c < L.nCells) coff[c] = off;
      n += tot;
    }
    __syncthreads();
    const int ncand = min(n, L.candCap);
    inreg = ncand <= 256 * KR;
    auto slot_of = [&](int k) {  // last cell whose offset is <= k (empty cells share offsets with their successor)
      int lo = 0, hi = L.nCells - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (coff[mid] <= k) lo = mid; else hi = mid - 1;
      }
      return (long long)lo * L.capCell + (k - coff[lo]);
    };
#pragma unroll
    for (int e = 0; e < KR; e++) {
      const int k = tid + 256 * e;
      kreg[e] = k < ncand ? slots[slot_of(k)] : 0u;
      lreg[e] = 0;
    }
#pragma unroll
    for (int e = 0; e < KR; e++) {
      const int k = tid + 256 * e;
      if (k < ncand) kd[k] = kreg[e];
    }
    for (int k = tid + 256 * KR; k < ncand; k += 256) kd[k] = slots[slot_of(k)];
  } else {
    for (int base = 0; base < L.nCells; base += 256) {
      const int c = base + tid;
      const int cnt = c < L.nCells ? cc[c] : 0;
      int tot;
      const int off = n + block_excl_scan(cnt, S.wsum, &tot);
      // four entries per round trip: written one by one, every load would wait behind the previous store
      // (the compiler cannot prove that `kd` and `slots` do not alias)
      for (int e0 = 0; e0 < cnt; e0 += 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = slots[(long long)c * L.capCell + min(e0 + u, cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (e0 + u < cnt && off + e0 + u < L.candCap) kd[off + e0 + u] = v[u];
      }
      n += tot;
    }
    __syncthreads();
    __threadfence_block();
    inreg = min(n, L.candCap) <= 256 * KR;
#pragma unroll
    for (int e = 0; e < KR; e++) {
      const int k = tid + 256 * e;
      kreg[e] = (inreg && k < min(n, L.candCap)) ? kd[k] : 0u;
      lreg[e] = 0;
    }
  }
  if (n > L.candCap) {
    if (tid == 0) atomicExch(err_flag, 1);
    n = L.candCap;
  }
  if (tid == 0) cand_count[f * P.nlevels + l] = n;
  __syncthreads();
  __threadfence_block();
  // Keys and their node labels stay in registers for the passes below when the level has at most 256 x 8
  // candidates (the usual case by far): a pass is then LDS traffic and barriers only, instead of two global
  // round trips per key.  `for_keys(body)`: body(k, key, label) may change the label.
  auto for_keys = [&](auto body) {
    if (inreg) {
#pragma unroll
      for (int e = 0; e < KR; e++) {
        const int k = tid + 256 * e;
        if (k < n) body(k, kreg[e], lreg[e]);
      }
    } else {
      for (int k = tid; k < n; k += 256) {
        int lab = kl[k];
        const int lab0 = lab;
        body(k, kd[k], lab);
        if (lab != lab0) kl[k] = (unsigned short)lab;
      }
    }
  };

  // ---- root nodes (:549-590)
  int cur = 0;
  const int nIni = L.nIni;
  const int H = L.maxBY - kBorder;
  for (int i = tid; i < CAP * 4; i += 256) S.ccount[i] = 0;
  __syncthreads();
  if (!inreg)
    for (int k = tid; k < n; k += 256) kl[k] = 0xffff;  // for_keys stores a label only when it changes
  for_keys([&](int, uint32_t kv, int &lab) {
    const int kx = kv & 0xfff;
    int b = (int)((float)kx / L.hX);
    b = min(max(b, 0), nIni - 1);
    lab = b;  // provisional: root index
    atomicAdd(&S.ccount[b], 1);
  });
  __syncthreads();
  int size = 0;
  {
    // compact non-empty roots (nIni is tiny: serial on thread 0)
    if (tid == 0) {
      int s = 0;
      for (int i = 0; i < nIni && s < CAP; i++) {
        const int c = S.ccount[i];
        S.newpos_old[i] = 0xffff;
        if (c == 0) continue;
        S.x0[0][s] = (unsigned short)(int)(L.hX * (float)i);
        S.x1[0][s] = (unsigned short)(int)(L.hX * (float)(i + 1));
        S.y0[0][s] = 0;
        S.y1[0][s] = (unsigned short)H;
        S.cnt[0][s] = (unsigned short)min(c, 65535);
        S.seq[0][s] = (unsigned short)s;
        S.newpos_old[i] = (unsigned short)s;
        s++;
      }
      S.s_newsize = s;
    }
    __syncthreads();
    size = S.s_newsize;
    for_keys([&](int, uint32_t, int &lab) { lab = S.newpos_old[lab]; });
    __syncthreads();
  }

  bool careful = false;
  bool finish = (n == 0);
  int guard = 0;
  while (!finish && guard++ < 64) {
    const int prevSize = size;
    // (1) expandable nodes and their processing order
    //     list-order rank first (also the creation order of the candidate array)
    int myflag[4], myrank[4];
    int local = 0;
    for (int e = 0; e < 4; e++) {
      const int pos = tid * 4 + e;
      myflag[e] = (pos < size && S.cnt[cur][pos] > 1) ? 1 : 0;
      local += myflag[e];
    }
    int m;
    int ex = block_excl_scan(local, S.wsum, &m);
    for (int e = 0; e < 4; e++) {
      const int pos = tid * 4 + e;
      myrank[e] = ex;
      ex += myflag[e];
      if (pos < CAP) S.prank[pos] = myflag[e] ? (unsigned short)myrank[e] : 0xffff;
      if (myflag[e]) S.order[myrank[e]] = (unsigned short)pos;
    }
    __syncthreads();
    if (careful && m > 1) {
      // rank by (count desc, seq desc): :690 sorts ascending and walks from the back.  One expandable node per
      // thread (in list-order rank j), its packed key compared with all m keys through broadcast LDS reads.
      for (int j = tid; j < m; j += 256) {
        const int p2 = S.order[j];
        S.cprefix[j] = (int)(((unsigned int)S.cnt[cur][p2] << 16) | S.seq[cur][p2]);  // scratch until step (3)
      }
      __syncthreads();
      int mypos[CAP / 256], myr[CAP / 256];
#pragma unroll
      for (int u = 0; u < CAP / 256; u++) {
        const int j = tid + 256 * u;
        mypos[u] = -1, myr[u] = 0;
        if (j >= m) continue;
        mypos[u] = S.order[j];
        const unsigned int mykey = (unsigned int)S.cprefix[j];
        int r = 0;
        for (int i = 0; i < m; i++) r += ((unsigned int)S.cprefix[i] > mykey) ? 1 : 0;
        myr[u] = r;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < CAP / 256; u++)
        if (mypos[u] >= 0) {
          S.prank[mypos[u]] = (unsigned short)myr[u];
          S.order[myr[u]] = (unsigned short)mypos[u];
        }
      __syncthreads();
    }
    // (2) key pass 1: child occupancy
    for (int i = tid; i < m * 4; i += 256) S.ccount[i] = 0;
    __syncthreads();
    for_keys([&](int, uint32_t kv, int &lab) {
      const int pos = lab;
      const int r = S.prank[pos];
      if (r == 0xffff) return;
      const int q = quadrant_of(kv & 0xfff, (kv >> 12) & 0xfff, S.x0[cur][pos], S.y0[cur][pos], S.x1[cur][pos],
                                S.y1[cur][pos]);
      atomicAdd(&S.ccount[r * 4 + q], 1);
    });
    __syncthreads();
    // (3) prefix over processing ranks: nonempty children, cut-off
    {
      int run_c = 0;  // running inclusive prefix carried across chunks
      int cutoff = m - 1;
      bool found = false;
      for (int base = 0; base < m; base += 256) {
        const int r = base + tid;
        int c = 0;
        if (r < m)
          for (int q = 0; q < 4; q++) c += S.ccount[r * 4 + q] > 0;
        int tot;
        const int exs = block_excl_scan(c, S.wsum, &tot);
        const int inc = run_c + exs + c;
        if (r < m) S.cprefix[r] = inc;
        run_c += tot;
      }
      __syncthreads();
      if (careful) {
        // first rank where the list reaches N (:741 break): size + (children - parents) >= N
        if (tid == 0) S.s_cutoff = m - 1;
        __syncthreads();
        for (int r = tid; r < m; r += 256) {
          const int sz = prevSize + S.cprefix[r] - (r + 1);
          if (sz >= N) atomicMin(&S.s_cutoff, r);
        }
        __syncthreads();
        cutoff = S.s_cutoff;
        found = true;
      }
      (void)found;
      const int totalChildren = m > 0 ? S.cprefix[cutoff] : 0;
      // (4) new list: children first
      const int nxt = cur ^ 1;
      if (tid == 0) S.s_nexp = 0;
      __syncthreads();
      int nexp_local = 0;
      for (int r = tid; r <= cutoff && r < m; r += 256) {
        const int pos = S.order[r];
        const int base_r = totalChildren - S.cprefix[r];  // children of later-processed parents precede
        const int px0 = S.x0[cur][pos], py0 = S.y0[cur][pos], px1 = S.x1[cur][pos], py1 = S.y1[cur][pos];
        const int midx = px0 + ((px1 - px0 + 1) >> 1), midy = py0 + ((py1 - py0 + 1) >> 1);
        int after = 0;  // nonempty children with a higher quadrant come first (n4 pushed last)
        for (int q = 3; q >= 0; q--) {
          const int c = S.ccount[r * 4 + q];
          if (c == 0) {
            S.newpos_child[r * 4 + q] = 0xffff;
            continue;
          }
          const int np = base_r + after;
          after++;
          S.newpos_child[r * 4 + q] = (unsigned short)np;
          S.x0[nxt][np] = (unsigned short)((q & 1) ? midx : px0);
          S.x1[nxt][np] = (unsigned short)((q & 1) ? px1 : midx);
          S.y0[nxt][np] = (unsigned short)((q & 2) ? midy : py0);
          S.y1[nxt][np] = (unsigned short)((q & 2) ? py1 : midy);
          S.cnt[nxt][np] = (unsigned short)min(c, 65535);
          S.seq[nxt][np] = (unsigned short)(r * 4 + q);
          nexp_local += c > 1;
        }
      }
      if (nexp_local) atomicAdd(&S.s_nexp, nexp_local);
      // kept old nodes follow, in their old order
      int keptlocal = 0, kf[4];
      for (int e = 0; e < 4; e++) {
        const int pos = tid * 4 + e;
        const int r = pos < size ? S.prank[pos] : 0xffff;
        kf[e] = (pos < size && (r == 0xffff || r > cutoff)) ? 1 : 0;
        keptlocal += kf[e];
      }
      int keptTotal;
      int kex = block_excl_scan(keptlocal, S.wsum, &keptTotal);
      for (int e = 0; e < 4; e++) {
        const int pos = tid * 4 + e;
        if (!kf[e]) {
          if (pos < CAP) S.newpos_old[pos] = 0xffff;
          continue;
        }
        const int np = totalChildren + kex;
        kex++;
        S.newpos_old[pos] = (unsigned short)np;
        if (np < CAP) {
          S.x0[nxt][np] = S.x0[cur][pos];
          S.x1[nxt][np] = S.x1[cur][pos];
          S.y0[nxt][np] = S.y0[cur][pos];
          S.y1[nxt][np] = S.y1[cur][pos];
          S.cnt[nxt][np] = S.cnt[cur][pos];
          S.seq[nxt][np] = S.seq[cur][pos];
        }
      }
      __syncthreads();
      const int newsize = totalChildren + keptTotal;
      // (5) key pass 2: relabel
      for_keys([&](int, uint32_t kv, int &lab) {
        const int pos = lab;
        const int r = S.prank[pos];
        if (r != 0xffff && r <= cutoff) {
          const int q = quadrant_of(kv & 0xfff, (kv >> 12) & 0xfff, S.x0[cur][pos], S.y0[cur][pos],
                                    S.x1[cur][pos], S.y1[cur][pos]);
          lab = S.newpos_child[r * 4 + q];
        } else {
          lab = S.newpos_old[pos];
        }
      });
      __syncthreads();
      const int nToExpand = S.s_nexp;
      cur = nxt;
      size = newsize;
      if (size > CAP - 4) {
        if (tid == 0) atomicExch(err_flag, 2);
        finish = true;
      }
      if (size >= N || size == prevSize)
        finish = true;  // :662-665 / :746-747
      else if (!careful && size + nToExpand * 3 > N)
        careful = true;  // :667
    }
    __syncthreads();
  }

  // ---- best response per node, first key wins ties (:748-766)
  for (int i = tid; i < CAP; i += 256) S.best[i] = 0;
  __syncthreads();
  for_keys([&](int k, uint32_t kv, int &lab) {
    const int pos = lab;
    if (pos >= CAP) return;
    const unsigned int v = ((kv >> 24) << 16) | (unsigned int)(65535 - k);
    atomicMax(&S.best[pos], v);
  });
  __syncthreads();
  uint32_t *out = sel + (long long)f * sel_per_frame + L.selBase;
  const int nout = min(size, L.capSel);
  for (int i = tid; i < nout; i += 256) out[i] = kd[65535 - (S.best[i] & 0xffff)];
  if (tid == 0) nk[f * P.nlevels + l] = (n == 0) ? 0 : nout;
}

// ------------------------------------------------------------------------------------------
// K4  cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on the unpadded level (:1093-1094).
// OpenCV <= 3.4.0 8-bit path: kernel quantised to {18,34,49,55,49,34,18}/256 per pass, row pass
// in int32, column pass (v + 2^15) >> 16 saturated.  64x16 output tile per workgroup; the
// (64+6)x(16+6) source tile and the 22x64 row sums live in LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
// same result for -n < i < 2n - 1 (the blur apron of 3 with n >= 4), without the loop
__device__ __forceinline__ int reflect101_near(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void k_blur(OrbDev P, FrameSrc src, unsigned level_mask) {
  __shared__ uint8_t t[22][72];
  __shared__ int hs[22][64];
  const int kq[7] = {18, 34, 49, 55, 49, 34, 18};
  const int tid = threadIdx.x, f = blockIdx.y;
  int tile = blockIdx.x;
  int l = 0;
  while (l + 1 < P.nlevels && tile >= P.lv[l + 1].tileBase) l++;
  const LevelGeom &L = P.lv[l];
  if (!((level_mask >> l) & 1u)) return;
  tile -= L.tileBase;
  const int ty = tile / L.tilesX, tx = tile - ty * L.tilesX;
  const int x0 = tx * 64, y0 = ty * 16;
  int pitch;
  const uint8_t *img = level_plane(P, src, l, f, pitch);
  for (int idx = tid; idx < 22 * 70; idx += 256) {
    const int r = idx / 70, c = idx - r * 70;
    const int sy = reflect101(y0 + r - 3, L.h), sx = reflect101(x0 + c - 3, L.w);
    t[r][c] = img[(long long)sy * pitch + sx];
  }
  __syncthreads();
  for (int idx = tid; idx < 22 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) acc += kq[i] * t[r][c + i];
    hs[r][c] = acc;
  }
  __syncthreads();
  uint8_t *dst = src.blur + (long long)f * src.blur_frame_stride + L.blur_off;
  for (int idx = tid; idx < 16 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    if (x0 + c >= L.w || y0 + r >= L.h) continue;
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) acc += kq[j] * hs[r + j][c];
    const int v = (acc + (1 << 15)) >> 16;
    dst[blur_tiled_off(x0 + c, y0 + r, L.pitch)] = (uint8_t)min(v, 255);
  }
}

// Fast path of the same filter.  A wavefront owns 16 four-pixel groups x 36 rows of FOUR consecutive frames
// (lane = frame * 16 + group: the level geometry, hence the row walk, is the same for all of them, and 16-group
// blocks fit the level widths far better than 64-group ones) and walks them top to bottom: per source row
// every lane loads the three aligned dwords around its group (row base in scalar registers, the lane's frame
// and column in a constant register offset), forms the four 7-tap row sums with v_alignbyte + v_dot4_u32_u8,
// keeps a 7-row sliding window of row sums in registers and emits one packed dword of output per row.  No
// LDS, 4-byte coalesced loads and stores.  The groups at the two ends of a row need pixels reflected about
// the first / last column (BORDER_REFLECT_101): blocks that contain them (wave-uniform) rebuild the affected
// dwords with byte permutes whose selectors are constant per lane.
constexpr int kBlurB = 16;                   // groups per block
constexpr int kBlurF = 64 / kBlurB;          // frames per wavefront
constexpr int kBlurMinW = 24, kBlurMinH = 4; // smaller levels take the generic kernel
constexpr int kBlurRows = 36;                // rows per job: 36 + 6 apron rows = six batches of seven

__global__ __launch_bounds__(256) void k_blur_groups(FrameSrc src, int lv0_generic, const int *__restrict__ job_tab,
                                                     int n_jobs, int n_frames) {
  const int lane = threadIdx.x & 63;
  const int job = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (job >= n_jobs) return;
  // wave-uniform job whose geometry comes from a table built with the handle: scalar registers for the whole walk
  const int *sd = job_tab + 16 * job;
  const int l = sd[0], cb = sd[1], sy = sd[2], Lh = sd[3], Lpitch = sd[4], ngroups = sd[5], w = sd[11];
  if (l == 0 && lv0_generic) return;
  const int f0 = blockIdx.y * kBlurF, fo = lane / kBlurB;
  const bool fvalid = f0 + fo < n_frames;
  const int fl = fvalid ? fo : 0;  // lanes past the batch repeat its first frame (not stored)
  const int G = min(cb * kBlurB + (lane & (kBlurB - 1)), ngroups - 1), x = 4 * G;
  int pitch;
  const uint8_t *img;
  unsigned fsrc;
  if (l == 0) {
    pitch = src.img0_pitch;
    img = src.img0 + (long long)f0 * src.img0_frame_stride;
    fsrc = (unsigned)(fl * src.img0_frame_stride);
  } else {
    pitch = sd[8];
    img = src.pyr + (long long)f0 * src.pyr_frame_stride + (((long long)sd[10] << 32) | (unsigned)sd[9]);
    fsrc = (unsigned)(fl * src.pyr_frame_stride);
  }
  // the lane's part of the store address: its frame and the tile column of its quad (lanes 4 t .. 4 t + 3 = the four column
  // groups of one 16-pixel tile; lanes past the row end keep their own tile column: what they store is row padding)
  const int Gu = cb * kBlurB + (lane & (kBlurB - 1));
  const bool tile_ok = fvalid && 4 * (Gu & ~3) < Lpitch && (Gu & ~3) < ngroups;
  uint8_t *dst = src.blur + (long long)f0 * src.blur_frame_stride + (((long long)sd[7] << 32) | (unsigned)sd[6]) +
                 (long long)fl * src.blur_frame_stride + (Gu >> 2) * 128;
  __shared__ __attribute__((aligned(16))) unsigned tr_all[4][4][64];
  unsigned (*tr)[64] = tr_all[threadIdx.x >> 6];
  // the three dwords of the lane: columns x-4.., x.., x+4..; at the row ends the neighbour is replaced by the
  // group itself (never read outside the row) and rebuilt below
  const unsigned vC = fsrc + (unsigned)x, vL = vC - (G > 0 ? 4u : 0u), vR = vC + (x + 4 < pitch ? 4u : 0u);
  // edge blocks: window byte k = 0..11 is pixel x - 4 + k; beyond the row it is the pixel reflected about the
  // last column, which lies at window byte k' >= 1 whenever a stored pixel needs it
  const bool edge_job = cb == 0 || (cb + 1) * kBlurB * 4 + 4 > w;  // uniform
  unsigned selC = 0x07060504u, selRA = 0x07060504u, selRB = 0u, selL = 0x03020100u;
  bool caseB = false, left = false;
  if (edge_job) {
    auto srck = [&](int k) {
      int i = x - 4 + k;
      if (i > w - 1) i = 2 * (w - 1) - i;
      return min(max(i - (x - 4), 0), 11);
    };
    selC = 0, selRA = 0, selRB = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int kc = srck(4 + b), kr = srck(8 + b);
      selC |= (unsigned)min(kc, 7) << (8 * b);           // perm(C, L): 0..3 = L, 4..7 = C
      selRA |= (unsigned)min(max(kr - 4, 0), 7) << (8 * b);  // perm(R, C): 0..3 = C, 4..7 = R
      selRB |= (unsigned)min(kr, 7) << (8 * b);          // perm(C, L)
    }
    caseB = x + 4 > w - 1;  // the right neighbour lies entirely beyond the row
    left = G == 0;          // pixels -4..-1 are pixels 4..1: bytes R0, C3, C2, C1 of perm(R, C)
    selL = 0x01020304u;
  }
  const int y0 = sy * kBlurRows, y1 = min(Lh, y0 + kBlurRows);
  const unsigned K0 = 18u | (34u << 8) | (49u << 16) | (55u << 24);
  const unsigned K1 = 49u | (34u << 8) | (18u << 16);
  // Row sums (< 2^16) of the last seven rows, as a ring of PAIRS of consecutive rows packed into one register
  // (slot s = rows s and s + 1): rows are taken seven at a time, so row u of a batch always closes slot u - 1
  // and the taps of an output row are slots u + 1, u + 3, u + 5 (mod 7) plus the new row itself -- all
  // compile-time indices, no register shuffling, and the column pass is three v_dot2_u32_u16 and one 24-bit
  // multiply-add per pixel.  All 21 loads of a batch are issued before the first result is stored.  (Written
  // row by row, every load waits behind the previous row's store -- the compiler cannot prove that `dst` and
  // `img` do not alias -- and a wave pays one memory round trip per row.)
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 W01 = {18, 34}, W23 = {49, 55}, W45 = {49, 34};
  unsigned pr[7][4], prev[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 7; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) pr[j][q] = 0;
  constexpr int RB = 7;
  for (int yb = y0 - 3; yb < y1 + 3; yb += RB) {
    unsigned Lr[RB], Cr[RB], Rr[RB];
#pragma unroll
    for (int u = 0; u < RB; u++) {
      const int yq = min(yb + u, y1 + 2);  // rows past the strip repeat its last one (unused)
      const uint8_t *row = img + (long long)reflect101_near(yq, Lh) * pitch;  // scalar
      Lr[u] = *reinterpret_cast<const unsigned *>(row + vL);
      Cr[u] = *reinterpret_cast<const unsigned *>(row + vC);
      Rr[u] = *reinterpret_cast<const unsigned *>(row + vR);
    }
#pragma unroll
    for (int u = 0; u < RB; u++) {
      const int yy = yb + u;
      if (yy >= y1 + 3) break;  // uniform
      unsigned Lw = Lr[u], Cw = Cr[u], Rw = Rr[u];
      if (edge_job) {  // uniform
        const unsigned l2 = __builtin_amdgcn_perm(Rw, Cw, selL);
        const unsigned c2 = __builtin_amdgcn_perm(Cw, Lw, selC);
        const unsigned ra = __builtin_amdgcn_perm(Rw, Cw, selRA), rb = __builtin_amdgcn_perm(Cw, Lw, selRB);
        Lw = left ? l2 : Lw;
        Rw = caseB ? rb : ra;
        Cw = c2;
      }
      // pixel q sits at byte 4+q of (L,C,R); its taps are bytes q+1 .. q+7
      unsigned hn[4];
      hn[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 1), K0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 1), K1, 0u, false), false);
      hn[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 2), K0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 2), K1, 0u, false), false);
      hn[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 3), K0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 3), K1, 0u, false), false);
      hn[3] = __builtin_amdgcn_udot4(Cw, K0, __builtin_amdgcn_udot4(Rw, K1, 0u, false), false);
      unsigned outw = 0;
      unsigned a4[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        pr[(u + 6) % 7][q] = prev[q] | (hn[q] << 16);  // rows (u - 1, u)
        prev[q] = hn[q];
        // taps: rows u-6, u-5 | u-4, u-3 | u-2, u-1 | u with weights 18, 34 | 49, 55 | 49, 34 | 18
        unsigned acc = (unsigned)__mul24(18, (int)hn[q]) + (1u << 15);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pr[(u + 5) % 7][q]), W45, acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pr[(u + 3) % 7][q]), W23, acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pr[(u + 1) % 7][q]), W01, acc, false);
        a4[q] = min(acc, 0x00ffffffu);  // byte 2 = min(acc >> 16, 255)
      }
      // the four saturated bytes (byte 2 of each sum) by two byte permutes and an OR: 7 instructions per group instead of 12
      outw = __builtin_amdgcn_perm(a4[1], a4[0], 0x0c0c0602u) | __builtin_amdgcn_perm(a4[3], a4[2], 0x06020c0cu);
      // Tiled plane: a row's dword per lane would be a 4-byte piece in each of sixteen 128-byte lines (measured: the blur
      // 50 % slower).  Four output rows are transposed through LDS instead -- lane (tile t, column group j) then holds the
      // four dwords of row j of its tile and stores them as ONE 16-byte tile row: as many line touches per row as the
      // row-major layout had.  (Rows come in groups of four aligned to four: a strip starts at a multiple of 36.)
      if (yy >= y0 + 3) {  // uniform
        const int y = yy - 3;
        tr[y & 3][lane] = outw;
        if ((y & 3) == 3 || y == y1 - 1) {  // uniform: the group is complete (or the strip ends)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          const int yr = (y & ~3) + (lane & 3);
          const u32x4_t v = *reinterpret_cast<const u32x4_t *>(&tr[lane & 3][lane & ~3]);
          if (tile_ok && yr <= y) *reinterpret_cast<u32x4_t *>(dst + blur_tiled_off(0, yr, Lpitch)) = v;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next group's writes stay behind these reads)
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
}


// K4 on the matrix cores (round 6).  k_blur_groups runs at two thirds of the VALU issue rate (227 M vector instructions per
// 1024 frames, ~15 per pixel) with the matrix pipe idle; both passes of the filter are products with a banded Toeplitz matrix
// of 8-bit weights {18,34,49,55,49,34,18}, and integer MFMA is exact, so the result stays bit-identical:
//   row pass     H[y][x] = sum_k P[y][k] Wh[k][x]: v_mfma_i32_32x32x32_i8, A = 32 rows x 16 consecutive pixels per lane -- a
//                16-byte load, xor 0x80 = (p - 128) as int8 --, B = the band (a constant operand: per strip from a table built
//                with the handle, BORDER_REFLECT_101 at the left / right edge folded into its entries), K = the 64 source
//                columns [x0 - 16, x0 + 48) in two steps, C = 128 so that the accumulator is the row sum - 2^15, a signed 16-bit number;
//   hand-over    the accumulator of that product has its COLUMN on the lane and 16 rows in its registers -- exactly the A
//                operand of the next product if the order of its K slots is chosen to match (slot 4 g + j of lane half h = row
//                8 g + 4 h + j; the band operand is permuted accordingly): no lane movement, no LDS.  The 16-bit sums are
//                split into a high and a low byte plane (4 v_perm per 4 values; the low plane xor 0x80: byte - 128);
//   column pass  V^T[x][y] = sum_k H[k][x] Wv[k][y] for both planes, against the tile's own 32 rows and the first six of the
//                next tile; value = acc_hi * 256 + acc_lo with the rounding and the three offsets riding in acc_lo's C operand;
//   store        M = x lands in the registers and N = y on the lanes, so a lane holds 4 x 4 consecutive pixels of ONE row:
//                (v >> 16) saturated by v_sat_pk_u8_i16, two v_permlane32_swap give every lane a whole 16-byte row of a 16 x 8
//                tile of the blurred plane, and 8 consecutive lanes write one 128-byte tile: whole lines, no LDS transpose.
// A wavefront walks a strip of 32 columns downwards, 32 rows per step (rows beyond the top / bottom are the reflected rows
// themselves, fetched by address); 6 MFMAs and ~90 vector instructions per 1024 pixels instead of ~240.
typedef int bm_i32x4 __attribute__((ext_vector_type(4)));
typedef int bm_i32x16 __attribute__((ext_vector_type(16)));
constexpr int kBmMinW = 64, kBmMinH = 16;  // smaller levels keep the other kernels
constexpr int kBmSegRows = 160;            // output rows per job (5 tiles of 32; one extra row-pass tile per job): 96 / 160 / 256 / 480
                                           // rows measured 0.546 / 0.516 / 0.624 / 0.548 ms per 1024 frames (level 0 = three equal jobs)

__device__ __forceinline__ unsigned bm_sat_pk(unsigned v) {
  unsigned r;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// (4 waves per SIMD: with a register budget of at most 256 the compiler keeps the accumulators in VGPRs -- at the default it put
//  them in AGPRs and paid a v_accvgpr_read per result register --, and FOUR is also what the kernel runs fastest at: 0.66 / 0.53 /
//  0.507 / 0.524 / 0.55 ms per 1024 frames at 2 / 3 / 4 / 5 / 6 resident waves per SIMD, the register count allows 6)
__attribute__((amdgpu_waves_per_eu(4, 4))) __global__ __launch_bounds__(256, 4) void k_blur_mfma(FrameSrc src, const int *__restrict__ job_tab, int n_jobs,
                                                   const int *__restrict__ tab, int bv_off) {
  const int lane = threadIdx.x & 63;
  const int job = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (job >= n_jobs) return;
  const int *sd = job_tab + 16 * job;  // wave-uniform: scalar loads
  const int l = sd[0], x0 = sd[1], r0 = sd[2], n_vt = sd[3], cmax = sd[5], Lh = sd[6], y_end = sd[13], Lpitch = sd[11];
  const int f = blockIdx.y, m = lane & 31, kg = lane >> 5;
  int pitch;
  const uint8_t *img;
  if (l == 0) {
    pitch = src.img0_pitch;
    img = src.img0 + (long long)f * src.img0_frame_stride;
  } else {
    pitch = sd[4];
    img = src.pyr + (long long)f * src.pyr_frame_stride + (((long long)sd[10] << 32) | (unsigned)sd[9]);
  }
  uint8_t *dst = src.blur + (long long)f * src.blur_frame_stride + (((long long)sd[8] << 32) | (unsigned)sd[7]);
  // the band operands: row pass (this strip's, two K steps), column pass (against the tile itself / the next tile)
  const bm_i32x4 *bh = reinterpret_cast<const bm_i32x4 *>(tab + sd[12]);
  const bm_i32x4 Bh0 = bh[lane], Bh1 = bh[64 + lane];
  const bm_i32x4 *bv = reinterpret_cast<const bm_i32x4 *>(tab + bv_off);
  const bm_i32x4 Bv1 = bv[lane], Bv2 = bv[64 + lane];
  // Source tiles (32 rows x the 64 bytes [x0 - 16, x0 + 48) of each) travel HBM -> LDS by LDS-DMA, two instructions of 16 rows x
  // 4 pieces of 16 bytes (LDS address = 16 * lane): a lane's operand -- 16 bytes of each of 32 DIFFERENT rows per instruction --
  // fetched straight into registers touched 32 lines per instruction and the texture-address path became the bound (measured:
  // products alone 0.25 ms, + loads 0.22, + stores 0.16 per 1024 frames).  Pieces are kept inside the row (clamped; the band
  // table was built with the same clamp: a slot that holds a duplicate or an unused column has weight 0) and stored at
  // (piece ^ row bits 2-3) so that the operand reads -- lanes = rows, 64 bytes apart -- spread over the banks.
  __shared__ __attribute__((aligned(16))) uint8_t stage_all[4][2][2048];
  lds_u8 *stage = (lds_u8 *)stage_all[threadIdx.x >> 6];
  const int drow = lane >> 2;                                  // row of the lane's piece within a half tile
  const int dcol = min(max(x0 - 16 + 16 * ((lane & 3) ^ ((drow >> 2) & 3)), 0), cmax);
  const bm_i32x16 Z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // The C operands of the two chains are constants (row pass: 128, so that the accumulator is the row sum minus 2^15, a signed
  // 16-bit number whose high byte is its own signed byte and whose low byte needs the xor only; column pass: the three offsets
  // and the rounding).  They are rebuilt from one register per product (16 v_mov of the fast class): kept as loop-invariant
  // 16-register tuples they cost 32 registers -- a wavefront per SIMD -- and the same moves, because the product overwrites C.
  auto splat = [](int v) {
    asm volatile("" : "+v"(v));
    bm_i32x16 c;
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = v;
    return c;
  };

  auto dma_tile = [&](int k) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int y = min(r0 + 32 * k + 16 * i + drow, Lh + 2);
      const uint8_t *g = img + (long long)reflect101_near(y, Lh) * pitch + dcol;
      __builtin_amdgcn_global_load_lds((gmem_u8 *)g, stage + 2048 * (k & 1) + 1024 * i, 16, 0, 0);
    }
  };
  // the lane's operand pieces of tile k: row m, pieces kg and 2 + kg (one asm block with its own wait: the compiler does not
  // see these reads, so it does not put a vmcnt(0) -- which would also wait for the tile just requested -- in front of them)
  const unsigned rd0 = (unsigned)(m * 64 + ((kg ^ ((m >> 2) & 3)) << 4)), rd1 = (unsigned)(m * 64 + (((2 + kg) ^ ((m >> 2) & 3)) << 4));
  auto read_tile = [&](int k, bm_i32x4 &a0, bm_i32x4 &a1) {
    const unsigned base = (unsigned)(uintptr_t)(stage + 2048 * (k & 1));
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a0), "=&v"(a1)
                 : "v"(base + rd0), "v"(base + rd1)
                 : "memory");
  };
  // row pass of one tile -> the two byte planes as the column pass's A operand
  auto row_pass = [&](bm_i32x4 a0, bm_i32x4 a1, bm_i32x4 &lo, bm_i32x4 &hi) {
#pragma unroll
    for (int i = 0; i < 4; i++) a0[i] ^= (int)0x80808080, a1[i] ^= (int)0x80808080;
    bm_i32x16 acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, Bh0, splat(128), 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, Bh1, acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const unsigned t01 = __builtin_amdgcn_perm((unsigned)acc[4 * g + 1], (unsigned)acc[4 * g], 0x05010400u);
      const unsigned t23 = __builtin_amdgcn_perm((unsigned)acc[4 * g + 3], (unsigned)acc[4 * g + 2], 0x05010400u);
      lo[g] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
      hi[g] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
    }
  };

  // the band operands have arrived before the loop: inside it the compiler would otherwise wait for them -- and with them for
  // the tile it has just requested -- at their first use of every iteration
  asm volatile("" ::"v"(Bh0), "v"(Bh1), "v"(Bv1), "v"(Bv2));
  bm_i32x4 p0, p1, lo, hi;
  dma_tile(0);
  dma_tile(1);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // tile 0 has landed (vmcnt counts in issue order: tile 1's two stay out)
  read_tile(0, p0, p1);
  row_pass(p0, p1, lo, hi);
  const int xt_ok = x0 + 16 * kg < Lpitch;
  for (int k = 0; k < n_vt; k++) {
    // outstanding, in issue order: tile k + 1 (two DMAs), the previous iteration's store -> all but the youngest one.  (Every
    // iteration DOES issue its store: a job has ceil(rows / 32) tiles, so each tile holds a row below y_end, and the lanes of
    // lane half 0 always own an existing tile column -- the count below relies on it.)
    if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    bm_i32x4 n0, n1;
    read_tile(k + 1, n0, n1);
    if (k + 1 < n_vt) dma_tile(k + 2);  // uniform; into the buffer tile k left; in flight during this tile's products
    bm_i32x4 lo2, hi2;
    row_pass(n0, n1, lo2, hi2);
    bm_i32x16 al = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, Bv1, splat(257 * (128 * 257) + (1 << 15)), 0, 0, 0);
    bm_i32x16 ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, Bv1, Z, 0, 0, 0);
    al = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo2, Bv2, al, 0, 0, 0);
    ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi2, Bv2, ah, 0, 0, 0);
    lo = lo2, hi = hi2;
    // lane (y = m, kg): registers 4 g + j = column x0 + 8 g + 4 kg + j of row y
    unsigned d[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      unsigned v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = ((unsigned)ah[4 * g + j] << 8) + (unsigned)al[4 * g + j];
      const unsigned s01 = bm_sat_pk(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u));
      const unsigned s23 = bm_sat_pk(__builtin_amdgcn_perm(v[3], v[2], 0x07060302u));
      d[g] = s01 | (s23 << 16);
    }
    // lane halves exchange so that lane (y, kg) holds the 16 bytes of row y of tile column x0 / 16 + kg: d0 d2 d1 d3
    const auto w02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
    const auto w13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
    const int y = r0 + 3 + 32 * k + m;
    if (xt_ok && y < y_end) {
      const u32x4_t out = {w02[0], w02[1], w13[0], w13[1]};
      *reinterpret_cast<u32x4_t *>(dst + blur_tiled_off(x0 + 16 * kg, y, Lpitch)) = out;
    }
  }
}

#include "orb_level_pass.inc"

// ------------------------------------------------------------------------------------------
// K3 + K5  IC_Angle (:79-107) + computeOrbDescriptor (:110-151) + the coordinate bookkeeping of
// :845-855 and :1102-1108.  One wavefront per key-point: the 749-pixel disc is reduced with
// cross-lane adds (two 31-wide rows per step), lane 0's fastAtan2 is evaluated by every lane,
// then lane t evaluates tests t, t+64, t+128, t+192 and four 64-bit ballots are the descriptor.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2, OpenCV 3.x
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// cos/sin of the descriptor steering angle: double Cody-Waite reduction + fixed-order minimax
// polynomials, rounded once to float (DESIGN.md "steered BRIEF trig").
__device__ __forceinline__ void cos_sin_f(float angle_rad, float &cs, float &sn) {
  const double x = (double)angle_rad;
  const double kd = floor(x * 6.36619772367581382433e-01 + 0.5);
  const int k = (int)kd;
  const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = ps * z + -2.50507602534068634195e-08;
  ps = ps * z + 2.75573137070700676789e-06;
  ps = ps * z + -1.98412698298579493134e-04;
  ps = ps * z + 8.33333333332248946124e-03;
  ps = ps * z + -1.66666666666666324348e-01;
  const double s = r + r * (z * ps);
  double pc = -1.13596475577881948265e-11;
  pc = pc * z + 2.08757232129817482790e-09;
  pc = pc * z + -2.75573143513906633035e-07;
  pc = pc * z + 2.48015872894767294178e-05;
  pc = pc * z + -1.38888888888741095749e-03;
  pc = pc * z + 4.16666666666666019037e-02;
  const double c = (1.0 - 0.5 * z) + z * (z * pc);
  double co, si;
  switch (k & 3) {
    case 0: co = c, si = s; break;
    case 1: co = -s, si = c; break;
    case 2: co = -c, si = -s; break;
    default: co = s, si = -c; break;
  }
  cs = (float)co;
  sn = (float)si;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// per key-point record of a k_describe batch (LDS)
struct DescRec {
  unsigned long long disc_base;  // level plane, pixel (px - 15, py - 15)
  unsigned long long blur_base;  // blurred plane, pixel (px - 19, py - 19)
  int pitch, bpitch;             // row pitch of the two planes
  int m10, m01;                  // intensity-centroid moments
  float a, b;                    // cos, sin of the steering angle
  int bytewise, pad;             // level-0 rows of the caller's image are not 16-byte aligned
  int wh, pad2;                  // level width | height << 16 (unused by this kernel)
};

// LDS values every lane reads from the same address, moved to scalar registers
__device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni_f32(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ gmem_u8 *uni_ptr(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (gmem_u8 *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ int wave_sum_i32(int x) {  // sum over the 64 lanes, result uniform (SGPR)
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);  // row_mirror: every lane holds its row's sum
  return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
         __builtin_amdgcn_readlane(x, 48);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// LDS row pitch of a staged window: four 16-byte chunks.  (Round 4 measured 80 and 96 -- pitches that walk the rows through
// all 32 banks instead of 16 -- against VERDICT r3's reading of the 48 % bank-conflict cycles: 0.746 / 0.747 / 0.749 ms per
// 1024 frames for 64 / 80 / 96, i.e. no effect: the LDS pipe is not what the kernel waits for, its window fetches are.)
constexpr int kWinPitch = 64;
constexpr int kWinBytes = 40 * kWinPitch;  // 39 rows

// Stage the ROWS x (<= 64 - 15) byte window whose top-left pixel is `origin` (row pitch `pitch`, rows 16-byte
// aligned) into LDS with 16-byte loads: lane + 64 j -> row (lane >> 2) + 16 j, chunk lane & 3, so the LDS
// address is simply 16 lane + 1024 j.  Rows past the window repeat its last row.  Returns the column of
// `origin` in the staged rows (0..15).  The memory pipe spends the same 16 cycles on a wavefront's byte
// gather as on these 1 KB loads: staging cuts its work per key-point by 4x.
// CH: 16-byte chunks fetched per row.  Three cover 48 bytes: enough for the 31-pixel orientation window at any
// alignment (15 + 31 <= 48) and for the 39-pixel descriptor window when it starts in the first ten bytes of its
// chunk (9 + 39 <= 48) -- the kernel is bound by the bytes its loads pull through the texture path (TA busy ~80 %),
// so the fourth chunk is only fetched when a window needs it.
template <int ROWS, int CH>
__device__ __forceinline__ void window_issue(unsigned long long origin, int pitch, int lane, u32x4 (&v)[(ROWS * CH + 63) / 64], int &ox) {
  constexpr int NJ = (ROWS * CH + 63) / 64;
  ox = uni_i32((int)(unsigned)origin & 15);
  gmem_u8 *base = uni_ptr(origin) - ox;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int idx = lane + 64 * j;
    const int row = min(CH == 4 ? idx >> 2 : idx / 3, ROWS - 1), chunk = CH == 4 ? idx & 3 : idx % 3;
    const unsigned goff = (unsigned)(row * pitch + 16 * chunk);
    v[j] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(base + goff);
  }
}
// the same for a window of a TILED plane (blur_tiled_off): chunk c of window row r is the 16-byte row (y0 + r) & 7 of tile
// ((x0 >> 4) + c, (y0 + r) >> 3); consecutive lanes walk down the rows of a tile, so a load instruction touches a third
// of the lines it touched in a row-major plane
template <int ROWS, int CH>
__device__ __forceinline__ void window_issue_tiled(unsigned long long plane, int pitch, int xy0, int lane, u32x4 (&v)[(ROWS * CH + 63) / 64], int &ox) {
  constexpr int NJ = (ROWS * CH + 63) / 64;
  const int x0 = uni_i32(xy0 & 0xffff), y0 = uni_i32(xy0 >> 16);
  ox = x0 & 15;
  gmem_u8 *base = uni_ptr(plane) + (x0 >> 4) * 128;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int idx = lane + 64 * j;
    const int row = min(CH == 4 ? idx >> 2 : idx / 3, ROWS - 1), chunk = CH == 4 ? idx & 3 : idx % 3;
    const int y = y0 + row;
    const unsigned goff = (unsigned)((y >> 3) * (pitch * 8) + (y & 7) * 16 + chunk * 128);
    v[j] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(base + goff);
  }
}
template <int ROWS, int CH, int NJ>
__device__ __forceinline__ void window_store(lds_u8 *slot, int lane, const u32x4 (&v)[NJ]) {
#pragma unroll
  for (int j = 0; j < (ROWS * CH + 63) / 64; j++) {
    const int idx = lane + 64 * j;
    const int row = CH == 4 ? idx >> 2 : idx / 3, chunk = CH == 4 ? idx & 3 : idx % 3;
    if (row < kWinBytes / kWinPitch)  // the slot ends after row 39 (lanes past the window repeat its last row: harmless)
      *(__attribute__((address_space(3))) u32x4 *)(slot + kWinPitch * row + 16 * chunk) = v[j];
  }
}

constexpr int kDescNK = 2;  // key-points in flight per wavefront of k_describe (1 and 4 measured slower, round 2)

// One workgroup per batch of 64 key-points of a frame, four phases separated by workgroup barriers:
//  0  lane = key-point: level search, selected key, plane addresses -> LDS records
//  1  wave = key-point (16 per wave, NK in flight): the 31 x 31 window around the key-point is staged in LDS,
//     the lane's 12 pixels of the 749-pixel disc are LDS byte reads at constant offsets, moments as signed
//     byte dot products (pixel - 128; the disc's u and v sum to zero, so the offset cancels exactly), DPP row
//     reduction
//  2  lane = key-point: fastAtan2, cos / sin (one evaluation for 64 key-points instead of one per wave),
//     key-point record out
//  3  wave = key-point: the 39 x 39 window of the blurred plane is staged in LDS (the rotated pattern points
//     scatter over it); lane t rotates the pattern points of tests t, t+64, t+128, t+192 (packed FP32
//     multiplies and adds, rounding by the 1.5 * 2^23 constant so that the integer falls out of the mantissa
//     and feeds the address arithmetic), eight LDS byte gathers, four 64-bit ballots are the descriptor.
// (The blur-on-demand form of this kernel -- phase 3 on the 45 x 45 RAW window -- was round 6's first step and is superseded by
//  k_describe_win below, which visits the window once; tools/attic/k_describe_od_r06.hip.txt.)
template <int NK>  // key-points a wave keeps in flight in phases 1 and 3
__global__ __launch_bounds__(256, 4) void k_describe(OrbDev P, FrameSrc src, const uint32_t *sel,
                                                  int sel_per_frame, const int *nk, int *counts, int capacity,
                                                  vo_keypoint *kps, uint8_t *desc, int lv0_bytewise,
                                                  int batches_per_frame, int n_frames, int *err_flag) {
  __shared__ DescRec rec[64];
  __shared__ __attribute__((aligned(16))) uint8_t win_lds[4][NK][kWinBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroups go round-robin over the 8 XCDs (each with its own L2): all batches of a frame run on one XCD
  // so that the windows of its key-points, which overlap line by line, are served by that L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int f = (slot / batches_per_frame) * 8 + xcd, g0 = (slot % batches_per_frame) * 64;
  if (f >= n_frames) return;
  // the frame's per-level key-point offsets: a prefix sum over its (at most 8) level counts, scalar
  int op[kMaxLevels + 1];
  {
    const int *nkp = nk + f * P.nlevels;
    int acc = 0;
#pragma unroll
    for (int i = 0; i < kMaxLevels; i++) {
      op[i] = acc;
      if (i < P.nlevels) acc += nkp[i];
    }
    op[kMaxLevels] = acc;
  }
  const int total = min(op[kMaxLevels], capacity);
  if (g0 == 0 && tid == 0 && counts) counts[f] = total;
  if (g0 == 0 && tid == 0 && op[kMaxLevels] > capacity) atomicExch(err_flag, 3);  // key-points dropped: reported by vo_orb_sync
  const int nv = min(total - g0, 64);  // key-points of this batch (uniform)
  if (nv <= 0) return;
  // per-lane constants of phase 1 (disc offsets and weights), independent of the key-point: in flight under phase 0
  u32x4 tab[9];
#pragma unroll
  for (int i = 0; i < 5; i++) tab[i] = reinterpret_cast<const u32x4 *>(c_desc_tab[lane])[i];
  uint32_t kv_own = 0;
  int l_own = 0, px_own = 0, py_own = 0;
  if (tid < nv) {
    const int g = g0 + tid;
    int l = 0, obase = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; i++)
      if (i < P.nlevels && g >= op[i]) l = i, obase = op[i];
    const LevelGeom &L = P.lv[l];  // lane-dependent level: vector loads, once per 64 key-points
    const uint32_t kv = sel[(long long)f * sel_per_frame + L.selBase + (g - obase)];
    const int px = (int)(kv & 0xfff) + kBorder, py = (int)((kv >> 12) & 0xfff) + kBorder;  // :849-850
    int pitch;
    const uint8_t *img = level_plane(P, src, l, f, pitch);
    const int bp = L.pitch;
    const uint8_t *bl = src.blur + (long long)f * src.blur_frame_stride + L.blur_off;
    DescRec r;
    r.disc_base = (unsigned long long)(uintptr_t)(img + (long long)(py - kHalfPatch) * pitch + (px - kHalfPatch));
    r.blur_base = (unsigned long long)(uintptr_t)bl;  // the plane; the window's origin (px - 19, py - 19) rides in `pad`
    r.pad = (px - kEdge) | ((py - kEdge) << 16);
    r.pitch = pitch, r.bpitch = bp;
    r.wh = L.w | (L.h << 16), r.pad2 = 0;
    r.m10 = r.m01 = 0;
    r.a = r.b = 0.f;
    r.bytewise = l == 0 && lv0_bytewise;
    rec[tid] = r;
    kv_own = kv, l_own = l, px_own = px, py_own = py;
  }
  __syncthreads();
  // ---- phase 1: moments
  const int k0 = wave * 16, k1 = min(k0 + 16, nv);
  for (int k = k0; k < k1; k += NK) {
    uint32_t va[NK][12];
    bool staged[NK];
    u32x4 wv[NK][2];
    int ox[NK];
#pragma unroll
    for (int s = 0; s < NK; s++) {
      const int kk = min(k + s, k1 - 1);  // uniform; past the end the last key-point is redone (not stored)
      staged[s] = uni_i32(rec[kk].bytewise) == 0;
      if (staged[s]) {
        window_issue<2 * kHalfPatch + 1, 3>(rec[kk].disc_base, uni_i32(rec[kk].pitch), lane, wv[s], ox[s]);
      } else {  // byte gathers straight from the caller's image
        gmem_u8 *base = uni_ptr(rec[kk].disc_base);
        const int pp = uni_i32(rec[kk].pitch);
#pragma unroll
        for (int i = 0; i < 12; i++) {
          const uint32_t c = tab[i >> 2][i & 3];
          va[s][i] = base[__umul24(c >> 6, (unsigned)pp) + (c & 63)];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NK; s++)
      if (staged[s]) window_store<2 * kHalfPatch + 1, 3>((lds_u8 *)win_lds[wave][s], lane, wv[s]);
    wave_sync();
#pragma unroll
    for (int s = 0; s < NK; s++)
      if (staged[s]) {
        const lds_u8 *w0 = (const lds_u8 *)win_lds[wave][s] + ox[s];
#pragma unroll
        for (int i = 0; i < 12; i++) va[s][i] = w0[tab[i >> 2][i & 3]];
      }
#pragma unroll
    for (int s = 0; s < NK; s++) {
      int s10 = 0, s01 = 0;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        // four pixels per dword (two byte permutes and one OR-XOR), minus 128
        const uint32_t w = (__builtin_amdgcn_perm(va[s][4 * j + 1], va[s][4 * j], 0x0c0c0400u) |
                            __builtin_amdgcn_perm(va[s][4 * j + 3], va[s][4 * j + 2], 0x04000c0cu)) ^ 0x80808080u;
        s10 = __builtin_amdgcn_sdot4((int)tab[3][j], (int)w, s10, false);
        s01 = __builtin_amdgcn_sdot4((int)(j == 0 ? tab[3][3] : tab[4][j - 1]), (int)w, s01, false);
      }
      const int m10 = wave_sum_i32(s10), m01 = wave_sum_i32(s01);
      if (lane == 0 && k + s < k1) rec[k + s].m10 = m10, rec[k + s].m01 = m01;
    }
    wave_sync();  // the windows are overwritten by the next group
  }
  // the lane's pattern rows (phase 3): in flight under phase 2; the registers of the disc constants are free now
#pragma unroll
  for (int i = 4; i < 9; i++) tab[i] = reinterpret_cast<const u32x4 *>(c_desc_tab[lane])[i];
  __syncthreads();
  // ---- phase 2: angle, cos / sin, key-point record
  if (tid < nv) {
    const float angle = fast_atan2_deg((float)rec[tid].m01, (float)rec[tid].m10);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);  // :109
    float a, b;
    cos_sin_f(angle * factorPI, a, b);
    rec[tid].a = a, rec[tid].b = b;
    const LevelGeom &L = P.lv[l_own];
    vo_keypoint kp;
    float fx = (float)px_own, fy = (float)py_own;
    if (l_own != 0) {  // :1102-1108
      fx *= L.scale;
      fy *= L.scale;
    }
    kp.x = fx;
    kp.y = fy;
    kp.size = (float)L.patchSize;
    kp.angle = angle;
    kp.response = (float)(kv_own >> 24);
    kp.octave = l_own;
    kp.class_id = -1;
    kps[(long long)f * capacity + g0 + tid] = kp;
  }
  __syncthreads();
  // ---- phase 3: descriptors
  constexpr float kMagic = 12582912.f;  // 1.5 * 2^23: x + kMagic has rint(x) + 0x400000 in its low 24 bits
  for (int k = k0; k < k1; k += NK) {
    u32x4 wv[NK][3];
    int ox[NK];
    bool narrow[NK];
#pragma unroll
    for (int s = 0; s < NK; s++) {
      const int kk = min(k + s, k1 - 1);
      const unsigned long long bb = rec[kk].blur_base;
      const int xy0 = rec[kk].pad;
      narrow[s] = uni_i32(xy0 & 15) <= 48 - (2 * kEdge + 1);  // uniform: three chunks per row suffice
      if (narrow[s]) {
        u32x4 w2[2];
        window_issue_tiled<2 * kEdge + 1, 3>(bb, uni_i32(rec[kk].bpitch), xy0, lane, w2, ox[s]);
        wv[s][0] = w2[0], wv[s][1] = w2[1];
      } else {
        window_issue_tiled<2 * kEdge + 1, 4>(bb, uni_i32(rec[kk].bpitch), xy0, lane, wv[s], ox[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < NK; s++) {
      if (narrow[s]) {
        const u32x4 w2[2] = {wv[s][0], wv[s][1]};
        window_store<2 * kEdge + 1, 3>((lds_u8 *)win_lds[wave][s], lane, w2);
      } else {
        window_store<2 * kEdge + 1, 4>((lds_u8 *)win_lds[wave][s], lane, wv[s]);
      }
    }
    wave_sync();
    uint32_t t[NK][8];
#pragma unroll
    for (int s = 0; s < NK; s++) {
      const int kk = min(k + s, k1 - 1);
      const float a = uni_f32(rec[kk].a), b = uni_f32(rec[kk].b);
      // (r + 19) * 64 + (q + 19 + ox) from the raw bit patterns 0x4B400000 + integer (arithmetic mod 2^32)
      const unsigned fold = (unsigned)(kEdge * kWinPitch + kEdge + ox[s]) - (0x4B400000u * (unsigned)kWinPitch + 0x4B400000u);
      const lds_u8 *wl = (const lds_u8 *)win_lds[wave][s];
      const v2f ba = {b, a}, anb = {a, -b}, mg = {kMagic, kMagic};
#pragma unroll
      for (int w = 0; w < 4; w++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int c = 18 + 4 * w + 2 * e;
          const float x = __uint_as_float(tab[c >> 2][c & 3]), y = __uint_as_float(tab[(c + 1) >> 2][(c + 1) & 3]);
          const v2f xx = {x, x}, yy = {y, y};
          const v2f rq = (xx * ba + yy * anb) + mg;  // (x b + y a, x a - y b), each rounded to nearest even
          const unsigned o = (__float_as_uint(rq.x) << 6) + __float_as_uint(rq.y) + fold;
          t[s][2 * w + e] = wl[o];
        }
    }
#pragma unroll
    for (int s = 0; s < NK; s++) {
      unsigned long long wd[4];
#pragma unroll
      for (int w = 0; w < 4; w++) wd[w] = __builtin_amdgcn_ballot_w64(t[s][2 * w] < t[s][2 * w + 1]);
      // lane w < 4 stores word w
      const unsigned long long mine = lane == 0 ? wd[0] : lane == 1 ? wd[1] : lane == 2 ? wd[2] : wd[3];
      if (lane < 4 && k + s < k1) reinterpret_cast<unsigned long long *>(desc + ((long long)f * capacity + g0 + k + s) * 32)[lane] = mine;
    }
    wave_sync();  // the windows are overwritten by the next group
  }
}


// ------------------------------------------------------------------------------------------
// K3 + K4 + K5 in one visit of the window (round 6; the default, VO_ORB_OPT_DESCRIBE_BLUR = 0).  No blurred pyramid is made:
// the 45 x 45 RAW window of a key-point -- the 39 x 39 window of the tests plus the blur's three pixels a side; it contains the
// 31 x 31 orientation window -- is brought into LDS ONCE and stays there from the moments to the tests; the blur of a whole
// pyramid (1.9 GB of traffic per 1024 frames) becomes ~1.5 M blurred pixels per frame computed where they are read.
// A wavefront owns kDwKpw = 4 key-points of ONE level from start to end (no workgroup barrier, no key-point records in LDS):
//  0  scalar: level of the workgroup, plane, one buffer descriptor; the lanes' key-point (lane & 3) moves to scalar registers
//     by v_readlane
//  1  the four windows are requested together, straight into LDS (buffer_load_dwordx4 ... lds: lane + 64 j <-> 16-byte chunk
//     idx % 3 of window row idx / 3, LDS address 16 idx = 48 row + 16 chunk, i.e. row pitch 48) at the exact origin px - 22
//     (byte-aligned 16-byte loads are legal on gfx950, tools/microbench/unaligned_load.hip) with hardware range checking;
//     BORDER_REFLECT_101: rows by address, columns by a byte fix-up of the staged rows (key-points within 22 px of the left /
//     right edge only); the moments are byte reads of the disc at (22 + v, 22 + u) of the staged window, signed byte dot
//     products (pixel - 128: the disc's u and v sum to zero, so the offset cancels), DPP wave sums
//  2  lanes 0..3: fastAtan2, cos / sin, key-point record
//  3  blur IN PLACE with the int8 matrix-core scheme of k_blur_mfma on v_mfma_i32_16x16x64_i8: row pass = 3 x 3 products (16
//     rows x 64 source columns against the band of 16 output columns; the accumulator -- column on the lane, four rows per
//     16-row tile in its registers -- is the A operand of the column pass as it stands: K slot 4 T + r of lane quarter q = row
//     16 T + 4 q + r), column pass = 3 x 3 x 2 byte planes; all constants ride in spare K slots (K slots 48..63 of the row pass
//     are one 16-byte chunk behind each window: two bytes 0x81 against weight 64 give its +128; the fourth dword of the column
//     pass's A operand is a per-lane constant against weights 16 / 127: +128 on the low plane, +33152 on the high one = the three
//     offsets and the rounding), so no accumulator is ever initialised; then the tests: lane t rotates the pattern points of
//     tests t, t + 64, t + 128, t + 192 (packed FP32, rounding by the 1.5 * 2^23 constant), eight LDS byte gathers, four ballots
// LDS: 4 x 4 windows of 48 x 48 + 16 bytes = 36.3 KB per workgroup, four workgroups per CU; <= 128 registers keeps the MFMA
// results in VGPRs.  profiles/r06_ab_describe.txt: 0.82 ms per 1024 frames against 1.17 for blurred planes + k_describe.
// ------------------------------------------------------------------------------------------
constexpr int kDwKpw = 4;
constexpr int kDwWaves = 4;  // wavefronts per workgroup (they do not interact; 1 / 2 measured 0.88 / 0.89 ms against 0.81)
constexpr int kDwPitch = 48;
constexpr int kDwWin = 48 * kDwPitch + 16;
// the lane's 12 disc pixels as byte offsets (22 + v) * 48 + (22 + u) in such a window (vo_orb_create)
__constant__ __attribute__((aligned(16))) uint32_t c_disc48[64][12];

// 64-lane sum whose total lands in lane 63 (DPP row reductions, then row_bcast:15 / row_bcast:31): uniform result
__device__ __forceinline__ int wave_sum63_i32(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);  // row_mirror: every lane holds its row's sum
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(x, 63);
}

// Workgroup -> (frame, level, 16 key-point slots of that level's selection): the level is uniform, so everything a window's
// address needs except the key-point itself comes from scalar registers, and the key-points' load does not wait for the level
// counts (their prefix sum is only needed for the output position): the chain in front of the window loads is ONE round trip
// (selection entry) instead of three (counts -> level record -> selection entry).
__global__ __launch_bounds__(64 * kDwWaves, 16 / kDwWaves) void k_describe_win(OrbDev P, FrameSrc src, const uint32_t *sel, int sel_per_frame, const int *nk,
                                                      int *counts, int capacity, vo_keypoint *kps, uint8_t *desc,
                                                      int groups_per_frame, int n_frames, int *err_flag, const int *od_tab) {
  __shared__ __attribute__((aligned(16))) uint8_t win_lds[kDwWaves][kDwKpw][kDwWin];
  typedef int od_i32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // all groups of a frame on one XCD (their windows overlap line by line: one L2 serves them)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int f = (slot / groups_per_frame) * 8 + xcd, gb = slot % groups_per_frame;
  if (f >= n_frames) return;
  int l = 0, gstart = 0;
  {
    int acc = 0;
    for (int i = 0; i < P.nlevels; i++) {
      if (gb >= acc) l = i, gstart = acc;
      acc += (P.lv[i].capSel + kDwWaves * kDwKpw - 1) / (kDwWaves * kDwKpw);
    }
  }
  l = __builtin_amdgcn_readfirstlane(l);
  const LevelGeom &L = P.lv[l];
  const int idx0 = (gb - gstart) * (kDwWaves * kDwKpw) + wave * kDwKpw;  // the wavefront's first slot in the level's selection
  // the key-points: requested before anything that depends on the counts (slots past the level's count hold stale entries:
  // never used; the index stays inside the level's block)
  const uint32_t kv = sel[(long long)f * sel_per_frame + L.selBase + min(idx0 + (lane & 3), L.capSel - 1)];
  int op_l = 0, total = 0;
  {
    const int *nkp = nk + f * P.nlevels;
    for (int i = 0; i < P.nlevels; i++) {
      const int c = nkp[i];
      if (i < l) op_l += c;
      total += c;
    }
  }
  const int nkl = nk[f * P.nlevels + l];
  if (gb == 0 && tid == 0 && counts) counts[f] = min(total, capacity);
  if (gb == 0 && tid == 0 && total > capacity) atomicExch(err_flag, 3);  // key-points dropped: reported by vo_orb_sync
  total = min(total, capacity);
  const int k0 = op_l + idx0;                                  // output position of the wavefront's first key-point
  const int nv = min(min(nkl - idx0, total - k0), kDwKpw);     // key-points of this wavefront (uniform)
  if (nv <= 0) return;                                         // (the kernel has no workgroup barrier)
  int pitch;
  const uint8_t *img = level_plane(P, src, l, f, pitch);
  const int LW = L.w, LH = L.h;
  const int px_v = (int)(kv & 0xfff) + kBorder, py_v = (int)((kv >> 12) & 0xfff) + kBorder;  // :849-850
  // per-lane constants: disc offsets / weights, pattern, band operands, window chunk of the lane
  u32x4 dof[3], tab[6];
#pragma unroll
  for (int i = 0; i < 3; i++) dof[i] = reinterpret_cast<const u32x4 *>(c_disc48[lane])[i];
#pragma unroll
  for (int i = 0; i < 6; i++) tab[i] = reinterpret_cast<const u32x4 *>(c_desc_tab[lane])[3 + i];  // [0..1]: weights, [1.5..5]: pattern
  od_i32x4 Bh[3], Bv[3];
  {
    const od_i32x4 *tb = reinterpret_cast<const od_i32x4 *>(od_tab);
#pragma unroll
    for (int X = 0; X < 3; X++) Bh[X] = tb[64 * X + lane];
    // the column pass's band for output tile Y is the one for tile 0 moved down by Y row tiles (K slot 4 T + r = row 16 T + 4 q + r:
    // the tile index is the dword index; dword 3 = the constant slots, the same for every Y): one load instead of three
    Bv[0] = tb[64 * 3 + lane];
    Bv[1] = od_i32x4{0, Bv[0].x, Bv[0].y, Bv[0].w};
    Bv[2] = od_i32x4{0, 0, Bv[0].x, Bv[0].w};
  }
  const int m = lane & 15, kq = lane >> 4;
  const int A3lo = kq < 2 ? 0x01010101 : 0;           // (the constants of the column pass: the low plane's 128, the high plane's 33152 = the three offsets and the rounding)
  const int A3hi = kq < 2 ? 0x05050505 : 0x20202020;
  lds_u8 *const W0 = (lds_u8 *)win_lds[wave][0];
  // K slots 48..63 of the row pass: bytes 0x80 (= 0 after the operand's xor), the last two 0x81 (+1 against weights 64 + 64 = the
  // row pass's 128); written once, never overwritten
  if (lane < kDwKpw) *(__attribute__((address_space(3))) u32x4 *)(W0 + kDwWin * lane + 48 * kDwPitch) = u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x81818080u};
  // ---- phase 1: the four windows (slots past the wavefront's count repeat its last key-point; never stored)
  int px[kDwKpw], py[kDwKpw];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, LH * pitch, 0x00020000);
  int wrow[3], wch[3];  // the lane's chunk of load j: window row, 16 x chunk
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int idx = lane + 64 * j;
    wrow[j] = min(idx / 3, 44), wch[j] = 16 * (idx % 3);
  }
  // (the level is uniform: row x pitch + chunk of the lane's three loads once per group; a window strictly inside the plane's
  //  rows then needs no vector instruction per load -- its origin rides in the load's scalar offset)
  int wro[3];
#pragma unroll
  for (int j = 0; j < 3; j++) wro[j] = (int)__umul24((unsigned)wrow[j], (unsigned)pitch) + wch[j];
#pragma unroll
  for (int s = 0; s < kDwKpw; s++) {
    const int ss = min(s, nv - 1);
    px[s] = __builtin_amdgcn_readlane(px_v, ss), py[s] = __builtin_amdgcn_readlane(py_v, ss);
    const bool inside = py[s] >= 22 && py[s] + 22 < LH;  // uniform: no row of the window is reflected
    // strictly inside: not even the plane's first or last row is touched, so every byte of the window is in the plane or in the
    // bytes that follow a row inside it -- nothing for the range check to catch (the scalar offset is not part of that check)
    const bool strictly = py[s] >= 23 && py[s] + 23 < LH;
    const int base = (py[s] - 22) * pitch + (px[s] - 22);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      int off, soff = 0;
      // (an offset in front of the plane is a huge unsigned one: the range check returns zeros, as it does behind the plane)
      if (strictly) {
        off = wro[j], soff = base;
      } else if (inside) {
        off = wro[j] + base;
      } else {
        const int y = reflect101_near(py[s] - 22 + wrow[j], LH);
        off = __mul24(y, pitch) + (px[s] - 22) + wch[j];
      }
      if (j < 2 || lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(W0 + kDwWin * s + 1024 * j), 16, off, soff, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();
#pragma unroll
  for (int s = 0; s < kDwKpw; s++) {
    lds_u8 *R = W0 + kDwWin * s;
    if (px[s] < 22 || px[s] + 22 > LW - 1) {  // uniform: BORDER_REFLECT_101 of the columns beyond the plane (at most six a side)
      if (px[s] < 22 && py[s] <= 22) {
        // a window over the plane's first pixel: the first chunk of raw row 0 starts in front of the plane, and the range check
        // drops the WHOLE 16-byte load, its in-plane bytes included -- they are fetched one by one (staged row 22 - py)
        if (lane >= 22 - px[s] && lane < 16) R[kDwPitch * (22 - py[s]) + lane] = __builtin_amdgcn_raw_buffer_load_b8(rs, lane - (22 - px[s]), 0, 0);
        wave_sync();
      }
      if (lane < 48) {
        for (int c = 0; c < 22 - px[s]; c++) R[kDwPitch * lane + c] = R[kDwPitch * lane + 2 * (22 - px[s]) - c];
        for (int c = LW - (px[s] - 22); c < 45; c++) R[kDwPitch * lane + c] = R[kDwPitch * lane + 2 * (LW - 1 - (px[s] - 22)) - c];
      }
      wave_sync();
    }
  }
  // moments (:79-107): the disc is rows / columns 7..37 of the window, which no fix-up touches
  int m10_v = 0, m01_v = 0;
  {
    uint32_t va[kDwKpw][12];
#pragma unroll
    for (int s = 0; s < kDwKpw; s++)
#pragma unroll
      for (int i = 0; i < 12; i++) va[s][i] = (W0 + kDwWin * s)[dof[i >> 2][i & 3]];
#pragma unroll
    for (int s = 0; s < kDwKpw; s++) {
      int s10 = 0, s01 = 0;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const uint32_t w = (__builtin_amdgcn_perm(va[s][4 * j + 1], va[s][4 * j], 0x0c0c0400u) |
                            __builtin_amdgcn_perm(va[s][4 * j + 3], va[s][4 * j + 2], 0x04000c0cu)) ^ 0x80808080u;
        s10 = __builtin_amdgcn_sdot4((int)tab[0][j], (int)w, s10, false);
        s01 = __builtin_amdgcn_sdot4((int)(j == 0 ? tab[0][3] : tab[1][j - 1]), (int)w, s01, false);
      }
      const int m10 = wave_sum63_i32(s10), m01 = wave_sum63_i32(s01);
      if ((lane & 3) == s) m10_v = m10, m01_v = m01;
    }
  }
  // ---- phase 2: angle, cos / sin, key-point record (lanes 0..3; the other lanes repeat them)
  float ca_v, sb_v;
  {
    const float angle = fast_atan2_deg((float)m01_v, (float)m10_v);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);  // :109
    cos_sin_f(angle * factorPI, ca_v, sb_v);
    if (lane < nv) {
      vo_keypoint kp;
      float fx = (float)px_v, fy = (float)py_v;
      if (l != 0) {  // :1102-1108
        fx *= L.scale;
        fy *= L.scale;
      }
      kp.x = fx;
      kp.y = fy;
      kp.size = (float)L.patchSize;
      kp.angle = angle;
      kp.response = (float)(kv >> 24);
      kp.octave = l;
      kp.class_id = -1;
      kps[(long long)f * capacity + k0 + lane] = kp;
    }
  }
  // ---- phase 3: blur in place, tests
  int aoff[3];
#pragma unroll
  for (int T = 0; T < 3; T++) aoff[T] = kq < 3 ? kDwPitch * (16 * T + m) + 16 * kq : 48 * kDwPitch;
  od_i32x4 a[kDwKpw][3];
#pragma unroll
  for (int s = 0; s < kDwKpw; s++)
#pragma unroll
    for (int T = 0; T < 3; T++) {
      const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *)(W0 + kDwWin * s + aoff[T]);
      a[s][T] = od_i32x4{(int)(v.x ^ 0x80808080u), (int)(v.y ^ 0x80808080u), (int)(v.z ^ 0x80808080u), (int)(v.w ^ 0x80808080u)};
    }
  wave_sync();  // (every lane has its rows: the blurred windows may now overwrite the raw ones)
  const od_i32x4 Z = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < kDwKpw; s++) {
    lds_u8 *R = W0 + kDwWin * s;
#pragma unroll
    for (int X = 0; X < 3; X++) {
      od_i32x4 lo, hi;
#pragma unroll
      for (int T = 0; T < 3; T++) {
        const od_i32x4 acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][T], Bh[X], Z, 0, 0, 0);  // row sums - 2^15, rows 16 T + 4 q + r
        const unsigned t01 = __builtin_amdgcn_perm((unsigned)acc[1], (unsigned)acc[0], 0x05010400u);
        const unsigned t23 = __builtin_amdgcn_perm((unsigned)acc[3], (unsigned)acc[2], 0x05010400u);
        lo[T] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
        hi[T] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
      }
      lo[3] = A3lo, hi[3] = A3hi;
#pragma unroll
      for (int Y = 0; Y < 3; Y++) {
        const od_i32x4 al = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo, Bv[Y], Z, 0, 0, 0);
        const od_i32x4 ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi, Bv[Y], Z, 0, 0, 0);
        unsigned v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = ((unsigned)ah[j] << 8) + (unsigned)al[j];
        const unsigned s01 = bm_sat_pk(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u));
        const unsigned s23 = bm_sat_pk(__builtin_amdgcn_perm(v[3], v[2], 0x07060302u));
        // lane (row 16 Y + m, quarter kq): blurred columns 16 X + 4 kq .. + 3 of that row
        *(__attribute__((address_space(3))) unsigned *)(R + kDwPitch * (16 * Y + m) + 16 * X + 4 * kq) = s01 | (s23 << 16);
      }
    }
  }
  wave_sync();
  constexpr float kMagic = 12582912.f;  // 1.5 * 2^23: x + kMagic has rint(x) + 0x400000 in its low 24 bits
  uint32_t t[kDwKpw][8];
#pragma unroll
  for (int s = 0; s < kDwKpw; s++) {
    const float ca = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(ca_v), s));
    const float sb = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(sb_v), s));
    // (r + 19) * 48 + (q + 19) from the raw bit patterns: v_mad_u32_u24 takes the low 24 bits of rint(x)'s pattern, 0x400000 + r
    const unsigned fold = (unsigned)(kEdge * kDwPitch + kEdge) - (0x400000u * (unsigned)kDwPitch + 0x4B400000u);
    const lds_u8 *wl = W0 + kDwWin * s;
    const v2f ba = {sb, ca}, anb = {ca, -sb}, mg = {kMagic, kMagic};
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int c = 6 + 4 * w + 2 * e;  // (pattern floats: entries 18.. of the lane's table row = tab[1.5..])
        const float x = __uint_as_float(tab[c >> 2][c & 3]), y = __uint_as_float(tab[(c + 1) >> 2][(c + 1) & 3]);
        const v2f xx = {x, x}, yy = {y, y};
        const v2f rq = (xx * ba + yy * anb) + mg;  // (x b + y a, x a - y b), each rounded to nearest even
        const unsigned o = __umul24(__float_as_uint(rq.x), (unsigned)kDwPitch) + __float_as_uint(rq.y) + fold;
        t[s][2 * w + e] = wl[o];
      }
  }
#pragma unroll
  for (int s = 0; s < kDwKpw; s++) {
    unsigned long long wd[4];
#pragma unroll
    for (int w = 0; w < 4; w++) wd[w] = __builtin_amdgcn_ballot_w64(t[s][2 * w] < t[s][2 * w + 1]);
    const unsigned long long mine = lane == 0 ? wd[0] : lane == 1 ? wd[1] : lane == 2 ? wd[2] : wd[3];
    if (lane < 4 && s < nv) reinterpret_cast<unsigned long long *>(desc + ((long long)f * capacity + k0 + s) * 32)[lane] = mine;
  }
}

}  // namespace

// ============================================================================================
// host side
// ============================================================================================
struct vo_orb {
  int nfeatures = 0, nlevels = 0, ini_th = 0, min_th = 0;
  float scale_factor = 0;
  float scale[kMaxLevels], inv_scale[kMaxLevels];
  int quota[kMaxLevels];
  int umax[16];
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t side = nullptr;          // blur runs here, concurrently with FAST / oct-tree (no data dependence)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork0 = nullptr, ev_fast0 = nullptr;
  vo_orb_stage_hook hook = nullptr;    // vo_orb_set_stage_hook: called when a stage's launches have been enqueued
  void *hook_user = nullptr;
  int early_level0 = 0;  // vo_orb_set_option(VO_ORB_OPT_EARLY_LEVEL0): measured -0.6 % on the extraction alone, nothing on the tracked step
  // geometry (valid for cfg_w x cfg_h)
  int cfg_w = 0, cfg_h = 0;
  OrbDev dev;
  long long pyr_frame = 0, blur_frame = 0, slots_frame = 0;
  int cells_frame = 0, keys_frame = 0, sel_frame = 0, tiles_frame = 0, max_kp = 0;
  int fast_tp = 48, fast_rows = 0, fast_interior = 0;  // k_fast_cell: LDS pitch, tile rows, list capacity
  int cell_tab_off = 0, strip_tab_off = 0;             // per-cell / per-strip geometry tables (ints into `tables`)
  size_t fast_lds = 0;
  bool oct_small = false;
  size_t rz_lds[kMaxLevels] = {0};
  bool rz_tiled[kMaxLevels] = {false};                             // the level's tiles fit k_resize4's LDS layout
  int rz_gtab_off[kMaxLevels] = {0}, rz_btab_off[kMaxLevels] = {0};  // its column tables (ints into `tables`)
  int blur_jobs = 0;               // k_blur_groups jobs per frame quad
  int blur_job0[kMaxLevels + 1] = {0};  // first job of every level (per-level launches next to fused levels)
  // k_level_pass (orb_level_pass.inc): per level whether it takes the fused pass, its tile pitch / rows, block table, LDS bytes
  int od_tab_off = 0;              // band operands of k_describe_win's window blur (6 x 64 x 4 ints into `tables`)
  int desc_blur = 0;               // vo_orb_set_option(VO_ORB_OPT_DESCRIBE_BLUR): 0 k_describe blurs its windows itself (default), 1 blurred planes
  bool blur_valid = false;         // the blurred planes of the last call exist (vo_orb_get_level(blurred) makes them on demand)
  bool last_lv0_rows16 = false;
  int last_lv0_unaligned = 0;
  int fused = 0;                   // vo_orb_set_option(VO_ORB_OPT_FUSED_LEVEL_PASS): 0 the three kernels (default), 1 k_level_pass (profiles/r05_ab_fused.txt)
  bool lp_ok[kMaxLevels] = {false};
  int lp_tp[kMaxLevels] = {0}, lp_tile_rows[kMaxLevels] = {0}, lp_score_rows[kMaxLevels] = {0}, lp_blocks[kMaxLevels] = {0};
  int lp_tab_off[kMaxLevels] = {0}, lp_list_cap[kMaxLevels] = {0};
  size_t lp_lds[kMaxLevels] = {0};
  int lp_redge_x[kMaxLevels] = {0}, lp_redge_j0[kMaxLevels] = {0};
  unsigned lp_selA[kMaxLevels] = {0}, lp_selB[kMaxLevels] = {0}, lp_selP[kMaxLevels] = {0};
  unsigned blur_generic_mask = 0;  // levels blurred by the generic kernel
  // k_blur_mfma: job table (16 ints per job: a strip of 32 columns x <= kBmSegRows rows), the column pass's band operands, the
  // levels it takes (the others keep k_blur_groups / k_blur)
  int blur_mfma = 0;               // vo_orb_set_option(VO_ORB_OPT_BLUR_KERNEL): 0 = matrix cores (default), 1 = k_blur_groups
  int bm_tab_off = 0, bm_bv_off = 0, bm_jobs = 0;
  int bm_job0[kMaxLevels + 1] = {0};
  unsigned bm_mask = 0;
  std::vector<int> tab_off;  // per level: offsets of xofs,xab,yofs,yab in tables
  vo::DevBuf tables, pyr, blur, slots, cellcnt, keydata, keylabel, candcnt, sel, nk, err;
  vo::DevBuf in_img, out_kp, out_desc, out_cnt;
  int batch_cap = 0;
  // last call
  FrameSrc last_src{};
  int last_frames = 0;
  // per-stage event timing
  bool timing = false;
  std::vector<hipEvent_t> ev;       // (VO_ORB_STAGES + 1) events per timed call
  int timed_calls = 0;
};

namespace {

int cv_round_f(float v) { return (int)lrintf(v); }
int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}

// coefficient tables of cv::resize(INTER_LINEAR) for CV_8U (OpenCV 3.x resize.cpp): per output
// column the left source column and two 11-bit weights, same per row.
void orb_resize_tables(int sw, int sh, int dw, int dh, std::vector<int> &xofs, std::vector<int> &xab,
                       std::vector<int> &yofs, std::vector<int> &yab) {
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  auto sat16 = [](float v) {
    int i = cv_round_f(v);
    return std::min(32767, std::max(-32768, i));
  };
  xofs.resize(dw), xab.resize(dw), yofs.resize(dh), yab.resize(dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= sw - 1) fx = 0, sx = sw - 1;
    xofs[dx] = sx;
    xab[dx] = (sat16((1.f - fx) * 2048) & 0xffff) | (sat16(fx * 2048) << 16);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    yofs[dy] = sy;
    yab[dy] = (sat16((1.f - fy) * 2048) & 0xffff) | (sat16(fy * 2048) << 16);
  }
}

int align_up(int v, int a) { return (v + a - 1) / a * a; }

// The blur of levels [l0, l1) of `n_frames` frames: k_blur_mfma where the level takes it (16-byte aligned rows: our own planes
// always, the caller's image if it is), else k_blur_groups (aligned dwords), else -- levels too small for either, a caller image
// that is not 4-byte aligned -- the generic LDS kernel.
void launch_blur_levels(vo_orb *h, const FrameSrc &S, int n_frames, bool lv0_rows16, int lv0_unaligned, hipStream_t bs, int l0, int l1) {
  const OrbDev &D = h->dev;
  const int *T = h->tables.as<int>();
  if (l1 <= l0) return;
  const unsigned lmask = (l1 >= 32 ? ~0u : (1u << l1) - 1u) & ~((1u << l0) - 1u);
  unsigned mf = h->blur_mfma == 0 ? (h->bm_mask & lmask) : 0u;
  if (!lv0_rows16) mf &= ~1u;
  // (the jobs of consecutive levels are consecutive: one launch per run of levels of the same kind)
  for (int l = l0; l < l1;) {
    const bool is_mf = (mf >> l) & 1u;
    int e = l + 1;
    while (e < l1 && (((mf >> e) & 1u) != 0) == is_mf) e++;
    if (is_mf) {
      const int jb = h->bm_job0[l], je = h->bm_job0[e];
      if (je > jb)
        hipLaunchKernelGGL(k_blur_mfma, dim3((je - jb + 3) / 4, n_frames), dim3(256), 0, bs, S, T + h->bm_tab_off + 16 * jb, je - jb, T,
                           h->bm_bv_off);
    } else {
      const int jb = h->blur_job0[l], je = h->blur_job0[e];
      if (je > jb)
        hipLaunchKernelGGL(k_blur_groups, dim3((je - jb + 3) / 4, (n_frames + kBlurF - 1) / kBlurF), dim3(256), 0, bs, S, lv0_unaligned,
                           T + h->strip_tab_off + 16 * jb, je - jb, n_frames);
    }
    l = e;
  }
  const unsigned gmask = (h->blur_generic_mask | (lv0_unaligned ? 1u : 0u)) & lmask & ~mf;
  if (gmask) hipLaunchKernelGGL(k_blur, dim3(h->tiles_frame, n_frames), dim3(256), 0, bs, D, S, gmask);
}


// ---- k_level_pass geometry of level l (orb_level_pass.inc): block table appended to `tables`; false: the level keeps the
// separate kernels.  xo / yo: source column / row of every column / row of level l + 1 (empty for the last level).
bool plan_level_pass(vo_orb *h, int l, const std::vector<int> &xo, const std::vector<int> &yo, std::vector<int> &tables) {
  const LevelGeom &L = h->dev.lv[l];
  h->lp_ok[l] = false;
  if (L.nCols < 1 || L.nRows < 1 || L.w < 48 || L.h < 24 || (L.pitch & 15)) return false;
  const int w = L.w, hh = L.h, wal = align_up(w, 16), hal = align_up(hh, 8);
  const bool has_next = !xo.empty();
  const int dw = has_next ? (int)xo.size() : 0, dh = has_next ? (int)yo.size() : 0;
  struct Span { int b0, b1, t0, c0, nc, f0, f1; };  // owned [b0, b1), tile origin, first cell column / row, cells, FAST range
  for (int TP : {96, 112, 128}) {
    // ---- columns
    std::vector<Span> cols;
    bool ok = true;
    const int nBX = (L.nCols + 1) / 2;
    auto fx0 = [&](int k) { return kBorder + 2 * k * L.wCell; };
    auto fx1 = [&](int k) { return std::min(kBorder + (2 * k + std::min(2, L.nCols - 2 * k)) * L.wCell + 6, L.maxBX); };
    int B = 0;
    for (int k = 0; k < nBX && ok; k++) {
      const int t0 = std::max(0, B - 16);
      if (fx0(k) < t0 || fx1(k) + 2 > t0 + TP) { ok = false; break; }  // (+2: the score entries of the block's second cell column sit one byte further)
      int hi = std::min(t0 + TP - 12, B + 96), nb;
      if (k + 1 < nBX) {
        hi = std::min(hi, fx0(k + 1) + 16);
        hi = std::min(hi, w - 8);
        nb = hi & ~15;
        if (nb < std::max(fx1(k + 1) + 2 - TP + 16, B + 16)) { ok = false; break; }
      } else {
        nb = wal <= hi ? wal : (std::min(hi, w - 8) & ~15);
        if (nb < B + 16) { ok = false; break; }
      }
      cols.push_back({B, nb, t0, 2 * k, std::min(2, L.nCols - 2 * k), fx0(k), fx1(k)});
      B = nb;
    }
    while (ok && B < wal) {  // what is left of the plane behind the last cell column
      const int t0 = std::max(0, B - 16), hi = std::min(t0 + TP - 12, B + 96);
      const int nb = wal <= hi ? wal : (std::min(hi, w - 8) & ~15);
      if (nb < B + 16) { ok = false; break; }
      cols.push_back({B, nb, t0, 0, 0, 0, 0});
      B = nb;
    }
    if (!ok) continue;
    // ---- rows
    std::vector<Span> rows;
    const int nBY = (L.nRows + 1) / 2;
    auto fy0 = [&](int i) { return kBorder + 2 * i * L.hCell; };
    auto fy1 = [&](int i) { return std::min(kBorder + (2 * i + std::min(2, L.nRows - 2 * i)) * L.hCell + 6, L.maxBY); };
    const int rows_cap = 2 * L.hCell + 16;
    B = 0;
    for (int i = 0; i < nBY; i++) {
      const int t0 = std::min(fy0(i), std::max(0, B - 4));
      int nb;
      if (i + 1 < nBY) nb = std::min((fy0(i + 1) + 4) & ~7, hal);
      else nb = std::min(hal, (t0 + rows_cap - 4) & ~7);
      if (nb < B + 8) { ok = false; break; }
      rows.push_back({B, nb, t0, 2 * i, std::min(2, L.nRows - 2 * i), fy0(i), fy1(i)});
      B = nb;
    }
    while (ok && B < hal) {
      const int t0 = std::max(0, B - 4);
      const int nb = std::min(hal, (t0 + rows_cap - 4) & ~7);
      if (nb < B + 8) { ok = false; break; }
      rows.push_back({B, nb, t0, 0, 0, 0, 0});
      B = nb;
    }
    if (!ok) continue;
    // ---- blocks
    int TR = 0, SR = 4, max_segs = 1;
    std::vector<int> tab;
    for (const Span &r : rows)
      for (const Span &c : cols) {
        int e[LP_INTS] = {0};
        const bool cells = r.nc > 0 && c.nc > 0;
        int t0y = r.t0;
        const bool bottom = r.b1 + 4 > hh;
        if (bottom) t0y = std::max(0, std::min(t0y, 2 * hh - 2 - (r.b1 + 3)));  // the rows the mirrored apron reads
        if (!cells) t0y = std::max(0, std::min(t0y, r.b0 - 4));
        const int yend = std::min(hh, std::max(cells ? r.f1 : 0, r.b1 + 4));
        e[LP_TX0] = c.t0, e[LP_TY0] = t0y, e[LP_TH] = yend - t0y;
        e[LP_BX0] = c.b0, e[LP_BX1] = c.b1, e[LP_BY0] = r.b0, e[LP_BY1] = r.b1;
        TR = std::max(TR, e[LP_TH]);
        int nc = 0;
        for (int q = 0; q < 4; q++) e[LP_CELL0 + q] = -1;
        if (cells)
          for (int ci = 0; ci < r.nc; ci++)
            for (int cj = 0; cj < c.nc; cj++) {
              e[LP_CELL0 + nc] = L.cellBase + (r.c0 + ci) * L.nCols + (c.c0 + cj);
              e[LP_CPOS0 + nc] = cj | (ci << 8);
              nc++;
            }
        e[LP_NCELLS] = nc;
        e[LP_FY0] = r.f0;
        if (cells) SR = std::max(SR, r.f1 - r.f0 + 4);
        if (has_next) {
          auto first_ge = [](const std::vector<int> &v, int step, int n, int bound) {  // first index i (of n, stride `step` in v) with v >= bound
            int i = 0;
            while (i < n && std::max(v[std::min((size_t)i * step, v.size() - 1)], 0) < bound) i++;
            return i;
          };
          const int ngr = (dw + 3) / 4;
          e[LP_G0] = first_ge(xo, 4, ngr, c.b0), e[LP_G1] = c.b1 >= wal ? ngr : first_ge(xo, 4, ngr, c.b1);
          e[LP_D0] = first_ge(yo, 1, dh, r.b0), e[LP_D1] = r.b1 >= hal ? dh : first_ge(yo, 1, dh, r.b1);
          if (e[LP_G1] - e[LP_G0] > 32) ok = false;
          // (every source row / column these read lies in the tile: sy + 1 <= by1 <= tile end, base + 11 <= bx1 + 10 < tx0 + TP)
          if (e[LP_D1] > e[LP_D0] && std::min(std::max(yo[e[LP_D1] - 1] + 1, 0), hh - 1) >= t0y + e[LP_TH]) ok = false;
        }
        e[LP_EDGE] = (c.b0 == 0 ? LP_EDGE_L : 0) | (c.b1 > ((w - 1) & ~7) ? LP_EDGE_R : 0) | (r.b0 == 0 ? LP_EDGE_T : 0) |
                     (bottom ? LP_EDGE_B : 0);
        const int segs = (c.b1 - c.b0) / 8;
        e[LP_SEGS] = segs, e[LP_R] = std::min(12, 64 / segs), e[LP_INV] = (65536 + segs - 1) / segs;
        for (int ln = 0; ln < 64; ln++)
          if ((int)(((unsigned)ln * (unsigned)e[LP_INV]) >> 16) != ln / segs) ok = false;
        max_segs = std::max(max_segs, segs);
        if (c.b1 - c.t0 + 12 > TP && !(c.b1 >= wal && wal - c.t0 + 4 <= TP && w + 11 <= c.t0 + TP)) ok = false;
        tab.insert(tab.end(), e, e + LP_INTS);
      }
    if (!ok) continue;
    const int list_cap = std::max(128, (L.wCell * L.hCell + 1) / 2);
    const size_t tile_bytes = (size_t)TP * TR, score = (size_t)fast_align16(TP * SR), lists = 4 * (size_t)fast_align16(2 * list_cap),
                 rings = 2 * (size_t)kLpRing * lp_ring_stride(kLpColSplit ? (max_segs + 1) / 2 : max_segs);
    const size_t lds = tile_bytes + std::max(score + lists, rings);
    if (lds > 64 * 1024 || (tile_bytes & 15)) continue;
    while (tables.size() % 4) tables.push_back(0);
    h->lp_tab_off[l] = (int)tables.size();
    tables.insert(tables.end(), tab.begin(), tab.end());
    h->lp_ok[l] = true, h->lp_tp[l] = TP, h->lp_tile_rows[l] = TR, h->lp_score_rows[l] = SR, h->lp_blocks[l] = (int)(rows.size() * cols.size());
    h->lp_list_cap[l] = list_cap, h->lp_lds[l] = lds;
    // right plane edge: window byte offset of column w - 1 in its segment, the dwords to rebuild and their byte selectors
    const int cW = ((w - 1) & 7) + 4, j0 = (cW + 1) >> 2;
    unsigned sel[2] = {0, 0};
    for (int j = j0; j <= j0 + 1; j++)
      for (int t = 0; t < 4; t++) {
        const int kk = 4 * j + t;
        int o = kk <= cW ? kk : 2 * cW - kk;
        o = std::min(std::max(o, 4 * (j0 - 1)), 4 * j0 + 3);
        sel[j - j0] |= (unsigned)(o - 4 * (j0 - 1)) << (8 * t);
      }
    h->lp_redge_x[l] = (w - 1) & ~7, h->lp_redge_j0[l] = j0, h->lp_selA[l] = sel[0], h->lp_selB[l] = sel[1];
    // the segment in front: column w - 1 sits at its window offset c = 12 + (w - 1) % 8; only offsets 13..15 can lie behind it
    h->lp_selP[l] = 0;
    if (((w - 1) & 7) < 3) {
      const int c = 12 + ((w - 1) & 7);
      unsigned sp = 0;
      for (int t = 0; t < 4; t++) {
        const int kk = 12 + t;
        const int o = std::min(std::max(kk <= c ? kk : 2 * c - kk, 8), 15);
        sp |= (unsigned)(o - 8) << (8 * t);
      }
      h->lp_selP[l] = sp;  // (never 0: byte 0 selects offset 12 = value 4)
    }
    return true;
  }
  return false;
}

int configure(vo_orb *h, int w, int h_img, int n_frames) {
  if (w != h->cfg_w || h_img != h->cfg_h) {
    // kernels of the previous call may still be reading the tables on the (non-blocking) handle stream
    VO_HIP_CHECK(hipStreamSynchronize(h->stream));
    OrbDev &D = h->dev;
    memset(&D, 0, sizeof(D));
    D.nlevels = h->nlevels;
    D.ini_th = h->ini_th;
    D.min_th = h->min_th;
    memcpy(D.umax, h->umax, sizeof(D.umax));
    long long pyr = 0, blur = 0, slots = 0;
    int cells = 0, keys = 0, sel = 0, tiles = 0, maxkp = 0;
    std::vector<int> tables;
    std::vector<std::vector<int>> lv_xo(h->nlevels + 1), lv_yo(h->nlevels + 1);  // source column / row of level l's columns / rows
    h->tab_off.assign(h->nlevels * 4, 0);
    int pw = w, ph = h_img;
    for (int l = 0; l < h->nlevels; l++) {
      LevelGeom &L = D.lv[l];
      L.w = cv_round_f((float)w * h->inv_scale[l]);  // ORBextractor.cpp:1120
      L.h = cv_round_f((float)h_img * h->inv_scale[l]);
      if (L.w > 4095 || L.h > 4095) {
        vo::set_error("image level %d is %dx%d; the packed key format supports <= 4095", l, L.w, L.h);
        return VO_ERR_INVALID;
      }
      L.pitch = align_up(L.w, 64);
      L.pyr_off = pyr;
      if (l > 0) pyr += (long long)L.pitch * L.h;
      L.blur_off = blur;
      blur += (long long)L.pitch * align_up(L.h, 8);  // (tiles of 8 rows)
      L.scale = h->scale[l];
      L.patchSize = (int)(31 * h->scale[l]);  // :842
      L.maxBX = L.w - kEdge + 3;
      L.maxBY = L.h - kEdge + 3;
      const float width = (float)(L.maxBX - kBorder), height = (float)(L.maxBY - kBorder);
      L.nCols = width > 0 ? (int)(width / 30.f) : 0;  // :791-794
      L.nRows = height > 0 ? (int)(height / 30.f) : 0;
      if (L.nCols < 1 || L.nRows < 1) {
        L.nCols = L.nRows = 0;
        L.wCell = L.hCell = 1;
      } else {
        L.wCell = (int)ceilf(width / L.nCols);
        L.hCell = (int)ceilf(height / L.nRows);
        if (L.wCell + 9 > kTileP || L.hCell + 6 > kTileP - 6) {
          vo::set_error("FAST cell %dx%d exceeds the LDS tile", L.wCell, L.hCell);
          return VO_ERR_INVALID;
        }
      }
      L.cellBase = cells;
      L.nCells = L.nCols * L.nRows;
      cells += L.nCells;
      L.capCell = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);  // strict-> NMS: no two kept pixels touch
      L.slotBase = slots;
      slots += (long long)L.nCells * L.capCell;
      L.quota = h->quota[l];
      const float ratio = (L.maxBY - kBorder) > 0 ? (float)(L.maxBX - kBorder) / (L.maxBY - kBorder) : 0.f;
      L.nIni = (int)roundf(ratio);  // :549
      if (L.nCells > 0 && L.nIni < 1) {
        vo::set_error("level %d is taller than wide (nIni = 0): the reference divides by zero here", l);
        return VO_ERR_INVALID;
      }
      if (L.nIni < 1) L.nIni = 1;
      L.hX = (float)(L.maxBX - kBorder) / L.nIni;
      L.capSel = std::max(L.quota + 4, 4 * L.nIni);
      if (L.capSel > kMaxList - 4) {
        vo::set_error("level %d quota %d exceeds the oct-tree list capacity %d", l, L.quota, kMaxList - 8);
        return VO_ERR_CAPACITY;
      }
      L.selBase = sel;
      sel += L.capSel;
      maxkp += L.capSel;
      long long cc = (long long)L.nCells * L.capCell;
      L.candCap = (int)std::min<long long>(cc, 65535);
      L.candBase = keys;
      keys += L.candCap;
      L.tilesX = (L.w + 63) / 64;
      L.tilesY = (L.h + 15) / 16;
      L.tileBase = tiles;
      tiles += L.tilesX * L.tilesY;
      if (l > 0) {
        std::vector<int> xo, xa, yo, ya;
        orb_resize_tables(pw, ph, L.w, L.h, xo, xa, yo, ya);
        lv_xo[l] = xo, lv_yo[l] = yo;
        h->tab_off[l * 4 + 0] = (int)tables.size();
        tables.insert(tables.end(), xo.begin(), xo.end());
        h->tab_off[l * 4 + 1] = (int)tables.size();
        tables.insert(tables.end(), xa.begin(), xa.end());
        h->tab_off[l * 4 + 2] = (int)tables.size();
        tables.insert(tables.end(), yo.begin(), yo.end());
        h->tab_off[l * 4 + 3] = (int)tables.size();
        tables.insert(tables.end(), ya.begin(), ya.end());
        int mdw = 1, mrows = 1;  // largest source rectangle of a 64 x 16 output tile
        // k_resize4's column tables: per column block the first source dword and the dwords per row ...
        while (tables.size() % 4) tables.push_back(0);
        h->rz_btab_off[l] = (int)tables.size();
        for (int x0 = 0; x0 < L.w; x0 += kRzW) {
          const int c0 = xo[x0] & ~3, c1 = std::min(xo[std::min(x0 + kRzW, L.w) - 1] + 1, pw - 1);
          const int ndw = ((c1 - c0) >> 2) + 1;
          mdw = std::max(mdw, ndw);
          tables.push_back(c0);
          tables.push_back(ndw);
        }
        // ... and per four-pixel group: aligned source column, byte offset of the 8-byte window, four byte-permute
        // selectors (tap 0 / tap 1 of each output into 16-bit lanes; 0x0c = constant 0) and coefficient pairs
        while (tables.size() % 4) tables.push_back(0);
        h->rz_gtab_off[l] = (int)tables.size();
        for (int g = 0; g < (L.w + 3) / 4; g++) {
          int sx[4];
          for (int q = 0; q < 4; q++) sx[q] = xo[std::min(4 * g + q, L.w - 1)];
          int e[12] = {sx[0] & ~3, sx[0] & 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int q = 0; q < 4; q++) {
            const int i0 = sx[q] - sx[0], i1 = std::min(sx[q] + 1, pw - 1) - sx[0];  // 0 <= i0 <= i1 <= 7 for scale < 2
            e[2 + q] = (int)((unsigned)i0 | 0x0c00u | ((unsigned)i1 << 16) | 0x0c000000u);
            e[6 + q] = xa[std::min(4 * g + q, L.w - 1)];
          }
          tables.insert(tables.end(), e, e + 12);
        }
        for (int y0 = 0; y0 < L.h; y0 += kRzH) {
          const int r0 = std::min(std::max(yo[y0], 0), ph - 1);
          const int r1 = std::min(std::max(yo[std::min(y0 + kRzH, L.h) - 1] + 1, 0), ph - 1);
          mrows = std::max(mrows, r1 - r0 + 1);
        }
        // kRzF stacked rectangles at the constant pitch, staged in whole rounds of 8 stacked rows
        h->rz_lds[l] = (size_t)kRzPitch * (((size_t)mrows * kRzF + 7) / 8 * 8) + 64;
        h->rz_tiled[l] = mdw <= kRzMaxDw && mrows <= kRzMaxRows;
      }
      pw = L.w, ph = L.h;
    }
    h->pyr_frame = align_up((int)std::max<long long>(pyr, 64), 256);
    h->blur_frame = align_up((int)blur, 256);
    h->slots_frame = slots;
    h->cells_frame = cells;
    {
      // k_fast_cell's per-cell geometry (16 ints per cell): level, cell origin and size incl. the 6-px overlap
      // (size 0: the cell is skipped, :801 / :811), key offset of the cell, slot block, plane pitch / offset
      while (tables.size() % 16) tables.push_back(0);
      // k_blur_groups: 16 ints per job of 16 groups x 32 rows; levels too small for it go to the generic kernel
      h->strip_tab_off = (int)tables.size();
      h->blur_jobs = 0;
      h->blur_generic_mask = 0;
      for (int l = 0; l < h->nlevels; l++) {
        const LevelGeom &L = D.lv[l];
        h->blur_job0[l] = h->blur_job0[l + 1] = h->blur_jobs;
        if (L.w < kBlurMinW || L.h < kBlurMinH) {
          h->blur_generic_mask |= 1u << l;
          continue;
        }
        const int ng = (L.w + 3) / 4, ncb = (ng + kBlurB - 1) / kBlurB;
        for (int sy = 0; sy < (L.h + kBlurRows - 1) / kBlurRows; sy++)
          for (int cb = 0; cb < ncb; cb++) {
            const int e[16] = {l, cb, sy, L.h, L.pitch, ng,
                               (int)(unsigned)(L.blur_off & 0xffffffffLL), (int)(L.blur_off >> 32), L.pitch,
                               (int)(unsigned)(L.pyr_off & 0xffffffffLL), (int)(L.pyr_off >> 32), L.w, 0, 0, 0, 0};
            tables.insert(tables.end(), e, e + 16);
            h->blur_jobs++;
          }
        h->blur_job0[l + 1] = h->blur_jobs;
      }
      {
        // k_blur_mfma: band operands and jobs.  Row pass, strip x0 of a level of width w: K slot (step s, lane half kg, byte
        // b) holds source column clamp(x0 - 16 + 32 s + 16 kg) + b (the kernel clamps its 16-byte pieces the same way); entry
        // [slot][n] = the weight with which that column enters output column x0 + n, BORDER_REFLECT_101 folded in (a column
        // that several slots hold is given to the first).  Column pass: slot (kg, b) = row 8 (b >> 2) + 4 kg + (b & 3) of the
        // row-pass tile (the order its accumulator registers have), against the tile itself (Bv1) and the next one (Bv2).
        static const int kw[7] = {18, 34, 49, 55, 49, 34, 18};
        while (tables.size() % 4) tables.push_back(0);
        h->bm_bv_off = (int)tables.size();
        for (int which = 0; which < 2; which++)
          for (int lane = 0; lane < 64; lane++)
            for (int q = 0; q < 4; q++) {
              unsigned wd = 0;
              for (int j = 0; j < 4; j++) {
                const int n = lane & 31, kg = lane >> 5, ys = 8 * q + 4 * kg + j + 32 * which, t = ys - n;
                if (t >= 0 && t <= 6) wd |= (unsigned)kw[t] << (8 * j);
              }
              tables.push_back((int)wd);
            }
        std::vector<std::pair<std::vector<int>, int>> seen;  // distinct row-pass operands of this handle
        std::vector<int> jobs;
        h->bm_jobs = 0, h->bm_mask = 0;
        for (int l = 0; l < h->nlevels; l++) {
          const LevelGeom &L = D.lv[l];
          h->bm_job0[l] = h->bm_job0[l + 1] = h->bm_jobs;
          if (L.w < kBmMinW || L.h < kBmMinH) continue;
          h->bm_mask |= 1u << l;
          const int cmax = align_up(L.w, 16) - 16;
          std::vector<int> strip_off;
          for (int x0 = 0; x0 < L.w; x0 += 32) {
            std::vector<int> mat(2 * 64 * 4, 0);
            for (int n = 0; n < 32 && x0 + n < L.w; n++) {
              int wcol[7], wval[7], nw = 0;  // the columns output x0 + n reads, reflected, with their summed weights
              for (int d = -3; d <= 3; d++) {
                int c = x0 + n + d;
                if (L.w == 1) c = 0;
                else while (c < 0 || c >= L.w) c = c < 0 ? -c : 2 * L.w - 2 - c;
                int i = 0;
                while (i < nw && wcol[i] != c) i++;
                if (i == nw) wcol[nw] = c, wval[nw] = 0, nw++;
                wval[i] += kw[d + 3];
              }
              for (int sstep = 0; sstep < 2; sstep++)
                for (int kg = 0; kg < 2; kg++) {
                  const int cb = std::min(std::max(x0 - 16 + 32 * sstep + 16 * kg, 0), cmax);
                  for (int b = 0; b < 16; b++)
                    for (int i = 0; i < nw; i++)
                      if (wcol[i] == cb + b && wval[i] != 0) {
                        mat[(sstep * 64 + kg * 32 + n) * 4 + (b >> 2)] |= wval[i] << (8 * (b & 3));
                        wval[i] = 0;  // given to the first slot that holds the column
                      }
                }
              for (int i = 0; i < nw; i++)
                if (wval[i] != 0) {
                  vo::set_error("k_blur_mfma: column %d of level %d is outside the strip window at x0 = %d", wcol[i], l, x0);
                  return VO_ERR_INVALID;
                }
            }
            int off = -1;
            for (auto &e : seen)
              if (e.first == mat) off = e.second;
            if (off < 0) {
              off = (int)tables.size();
              tables.insert(tables.end(), mat.begin(), mat.end());
              seen.emplace_back(std::move(mat), off);
            }
            strip_off.push_back(off);
          }
          for (int ys = 0; ys < L.h; ys += kBmSegRows) {
            const int ye = std::min(L.h, ys + kBmSegRows);
            for (int x0 = 0, si = 0; x0 < L.w; x0 += 32, si++) {
              const int e[16] = {l, x0, ys - 3, (ye - ys + 31) / 32, L.pitch, cmax, L.h,
                                 (int)(unsigned)(L.blur_off & 0xffffffffLL), (int)(L.blur_off >> 32),
                                 (int)(unsigned)(L.pyr_off & 0xffffffffLL), (int)(L.pyr_off >> 32), L.pitch, strip_off[si], ye, 0, 0};
              jobs.insert(jobs.end(), e, e + 16);
              h->bm_jobs++;
            }
          }
          h->bm_job0[l + 1] = h->bm_jobs;
        }
        while (tables.size() % 16) tables.push_back(0);
        h->bm_tab_off = (int)tables.size();
        tables.insert(tables.end(), jobs.begin(), jobs.end());
      }
      {
        // k_describe_win: band operands of its window blur on v_mfma_i32_16x16x64_i8, lane (n = lane & 15, kq = lane >> 4),
        // byte b of the lane's 16.  Row pass X = 0..2 (16 output columns each): slot (kq, b) = staged column 16 kq + b, which enters
        // output column 16 X + n with weight w[16 kq + b - (16 X + n)]; slots 62 and 63 (the two 0x81 bytes) weigh 64.  Column
        // pass Y = 0..2: slot (kq, 4 T + r) = row 16 T + 4 kq + r of the row-pass result (the order its accumulator registers have),
        // weight w[row - (16 Y + n)]; T = 3: the constant slots (16 for the lane quarters 0 and 1, 127 for 2 and 3).
        static const int kw[7] = {18, 34, 49, 55, 49, 34, 18};
        while (tables.size() % 4) tables.push_back(0);
        h->od_tab_off = (int)tables.size();
        for (int which = 0; which < 2; which++)
          for (int X = 0; X < 3; X++)
            for (int lane = 0; lane < 64; lane++)
              for (int d = 0; d < 4; d++) {
                unsigned wd = 0;
                for (int j = 0; j < 4; j++) {
                  const int n = lane & 15, kq = lane >> 4, b = 4 * d + j;
                  int v;
                  if (which == 0) {
                    const int t = 16 * kq + b - (16 * X + n);
                    v = (t >= 0 && t <= 6) ? kw[t] : 0;
                    if (kq == 3 && b >= 14) v = 64;
                  } else {
                    const int T = b >> 2, r = b & 3, t = 16 * T + 4 * kq + r - (16 * X + n);
                    v = T < 3 ? ((t >= 0 && t <= 6) ? kw[t] : 0) : (kq < 2 ? 16 : 127);
                  }
                  wd |= (unsigned)v << (8 * j);
                }
                tables.push_back((int)wd);
              }
      }
      h->cell_tab_off = (int)tables.size();
      for (int l = 0; l < h->nlevels; l++) {
        const LevelGeom &L = D.lv[l];
        for (int ci = 0; ci < L.nCells; ci++) {
          const int ci_i = ci / L.nCols, ci_j = ci - ci_i * L.nCols;
          const int iniX = kBorder + ci_j * L.wCell, iniY = kBorder + ci_i * L.hCell;
          const bool skip = iniX >= L.maxBX - 6 || iniY >= L.maxBY - 3;
          const int maxX = std::min(iniX + L.wCell + 6, L.maxBX), maxY = std::min(iniY + L.hCell + 6, L.maxBY);
          const long long so = L.slotBase + (long long)ci * L.capCell;
          if (so > 0x7fffffffLL) {
            vo::set_error("cell slot block of %lld entries per frame exceeds the 32-bit cell table", so);
            return VO_ERR_CAPACITY;
          }
          const int e[16] = {l, iniX, iniY, skip ? 0 : maxX - iniX, skip ? 0 : maxY - iniY, ci_j * L.wCell, ci_i * L.hCell,
                             (int)so, L.capCell, L.pitch, (int)(unsigned)(L.pyr_off & 0xffffffffLL), (int)(L.pyr_off >> 32),
                             0, 0, 0, 0};
          tables.insert(tables.end(), e, e + 16);
        }
      }
    }
    for (int l = 0; l < h->nlevels; l++) {
      // the next level is produced by the fused pass only if k_resize4's column-group table exists for it
      const bool next = l + 1 < h->nlevels;
      const bool next_ok = !next || (h->rz_tiled[l + 1] && (double)D.lv[l].w / D.lv[l + 1].w < 1.99 && D.lv[l + 1].pitch >= ((D.lv[l + 1].w + 3) & ~3));
      static const std::vector<int> none;
      if (!next_ok || !plan_level_pass(h, l, next ? lv_xo[l + 1] : none, next ? lv_yo[l + 1] : none, tables)) h->lp_ok[l] = false;
      if (h->lp_ok[l] && h->lp_lds[l] > 48 * 1024) {
        const void *fn = h->lp_tp[l] == 96 ? (const void *)k_level_pass<96> : h->lp_tp[l] == 112 ? (const void *)k_level_pass<112> : (const void *)k_level_pass<128>;
        VO_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lp_lds[l]));
      }
    }
    {
      int mw = 1, mh = 1;
      for (int l = 0; l < h->nlevels; l++)
        if (D.lv[l].nCols > 0) mw = std::max(mw, D.lv[l].wCell), mh = std::max(mh, D.lv[l].hCell);
      int mcap = 0;
      for (int l = 0; l < h->nlevels; l++) mcap = std::max(mcap, std::max(D.lv[l].capSel, D.lv[l].nIni));
      h->oct_small = mcap <= 256 - 4;
      h->fast_tp = mw + 9 <= 48 && mh + 6 <= 64 ? 48 : kTileP;  // 48: three 16-byte chunks per row, <= 192 chunks per tile
      h->fast_rows = mh + 6;
      h->fast_interior = std::max(128, (mw * mh + 1) / 2);  // survivor-list entries: half the cell's pixels (see k_fast_cell)
      h->fast_lds = 4 * (size_t)fast_cell_lds(h->fast_tp, h->fast_rows, h->fast_interior);
      if (h->fast_lds > 64 * 1024) {
        for (const void *fn : {(const void *)k_fast_cell<48, false>, (const void *)k_fast_cell<48, true>,
                               (const void *)k_fast_cell<kTileP, false>, (const void *)k_fast_cell<kTileP, true>})
          VO_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->fast_lds));
      }
    }
    h->keys_frame = align_up(keys, 64);
    h->sel_frame = sel;
    h->tiles_frame = tiles;
    h->max_kp = maxkp;
    VO_CHECK(h->tables.reserve(std::max<size_t>(tables.size() * sizeof(int), 64)));
    if (!tables.empty())
      VO_HIP_CHECK(hipMemcpy(h->tables.p, tables.data(), tables.size() * sizeof(int), hipMemcpyHostToDevice));  // synchronous: `tables` is a local
    h->cfg_w = w;
    h->cfg_h = h_img;
    h->batch_cap = 0;
  }
  if (n_frames > h->batch_cap) {
    const size_t B = (size_t)n_frames;
    VO_CHECK(h->pyr.reserve(B * h->pyr_frame));
    VO_CHECK(h->blur.reserve(B * h->blur_frame));
    VO_CHECK(h->slots.reserve(std::max<size_t>(B * h->slots_frame * 4, 64)));
    VO_CHECK(h->cellcnt.reserve(std::max<size_t>(B * h->cells_frame * 4, 64)));
    VO_CHECK(h->keydata.reserve(B * h->keys_frame * 4));
    VO_CHECK(h->keylabel.reserve(B * h->keys_frame * 2));
    VO_CHECK(h->candcnt.reserve(B * kMaxLevels * 4));
    VO_CHECK(h->sel.reserve(B * h->sel_frame * 4));
    VO_CHECK(h->nk.reserve(B * kMaxLevels * 4));
    if (!h->err.p) {
      VO_CHECK(h->err.reserve(64));
      VO_HIP_CHECK(hipMemsetAsync(h->err.p, 0, 64, h->stream));
    }
    h->batch_cap = n_frames;
  }
  return VO_OK;
}

int run_pipeline(vo_orb *h, const uint8_t *dev_images, int n_frames, int w, int hh, int stride,
                 size_t frame_stride, vo_keypoint *dkp, uint8_t *ddesc, int capacity, int32_t *dcounts) {
  VO_CHECK(configure(h, w, hh, n_frames));
  const OrbDev &D = h->dev;
  hipStream_t st = h->stream;  // (not const: the early level-0 launches borrow the launch lambdas for the side stream)
  FrameSrc S;
  S.img0 = dev_images;
  S.img0_frame_stride = (long long)frame_stride;
  S.img0_pitch = stride;
  S.pyr = h->pyr.as<uint8_t>();
  S.pyr_frame_stride = h->pyr_frame;
  S.blur = h->blur.as<uint8_t>();
  S.blur_frame_stride = h->blur_frame;
  h->last_src = S;
  h->last_frames = n_frames;
  // (the device error flag is sticky across calls: vo_orb_sync reads and clears it, so an overflow inside an
  // asynchronous batch is never lost between two syncs)
  // kernels that read aligned dwords need 4-byte aligned caller rows; otherwise level 0 takes byte paths
  const int lv0_unaligned =
      ((reinterpret_cast<uintptr_t>(dev_images) | (uintptr_t)stride | (uintptr_t)frame_stride) & 3) ? 1 : 0;
  // k_describe stages level-0 windows with 16-byte loads when the caller's rows allow it
  const int lv0_not16 =
      ((reinterpret_cast<uintptr_t>(dev_images) | (uintptr_t)stride | (uintptr_t)frame_stride) & 15) ? 1 : 0;
  hipEvent_t *ev = nullptr;
  if (h->timing) {
    const size_t need = (size_t)(h->timed_calls + 1) * (VO_ORB_STAGES + 1);
    while (h->ev.size() < need) {
      hipEvent_t e;
      VO_HIP_CHECK(hipEventCreate(&e));
      h->ev.push_back(e);
    }
    ev = h->ev.data() + (size_t)h->timed_calls * (VO_ORB_STAGES + 1);
    h->timed_calls++;
  }
#define VO_STAGE_MARK(i)                               \
  do {                                                 \
    if (ev) VO_HIP_CHECK(hipEventRecord(ev[i], st));   \
    if (h->hook && (i) > 0) h->hook((i)-1, st, h->hook_user); \
  } while (0)
  VO_STAGE_MARK(0);
  const int *T = h->tables.as<int>();
  // level l from level l - 1 (sequential by construction, :1129)
  auto launch_resize = [&](int l) {
    const LevelGeom &L = D.lv[l], &Pv = D.lv[l - 1];
    const uint8_t *sp = l == 1 ? dev_images : S.pyr + Pv.pyr_off;
    const long long sfs = l == 1 ? (long long)frame_stride : h->pyr_frame;
    const int spitch = l == 1 ? stride : Pv.pitch;
    // 4 outputs span <= 3*scale + 2 source columns; with the aligned start that fits 12 bytes for
    // scale factors below 2 and needs 4-byte aligned source rows with readable padding to the pitch
    const bool aligned = ((reinterpret_cast<uintptr_t>(sp) | (uintptr_t)spitch | (uintptr_t)sfs) & 3) == 0 &&
                         (double)Pv.w / L.w < 1.99 && ((Pv.w + 3) & ~3) <= spitch && h->rz_tiled[l] && L.pitch >= ((L.w + 3) & ~3);
    if (aligned) {
      dim3 grid((L.w + kRzW - 1) / kRzW, (L.h + kRzH - 1) / kRzH, (n_frames + kRzF - 1) / kRzF);
      hipLaunchKernelGGL(k_resize4, grid, dim3(256), h->rz_lds[l], st, sp, sfs, spitch, Pv.w, Pv.h, S.pyr + L.pyr_off,
                         (long long)h->pyr_frame, L.pitch, L.w, L.h, T + h->rz_gtab_off[l], T + h->rz_btab_off[l],
                         T + h->tab_off[l * 4 + 2], T + h->tab_off[l * 4 + 3], n_frames);
    } else {
      dim3 grid((L.w + 63) / 64, (L.h + 3) / 4, n_frames), block(64, 4);
      hipLaunchKernelGGL(k_resize, grid, block, 0, st, sp, sfs, spitch, Pv.w, Pv.h, S.pyr + L.pyr_off,
                         (long long)h->pyr_frame, L.pitch, L.w, L.h, T + h->tab_off[l * 4 + 0],
                         T + h->tab_off[l * 4 + 1], T + h->tab_off[l * 4 + 2], T + h->tab_off[l * 4 + 3]);
    }
  };
  auto launch_fast = [&](int cell_begin, int cell_end) {
    if (cell_end <= cell_begin) return;
    const dim3 grid((cell_end - cell_begin + 4 * kFastCpw - 1) / (4 * kFastCpw), n_frames);
    auto fast = h->fast_tp == 48 ? (lv0_unaligned ? k_fast_cell<48, true> : k_fast_cell<48, false>)
                                 : (lv0_unaligned ? k_fast_cell<kTileP, true> : k_fast_cell<kTileP, false>);
    hipLaunchKernelGGL(fast, grid, dim3(256), h->fast_lds, st, D, S, h->slots.as<uint32_t>(), h->slots_frame,
                       h->cellcnt.as<int>(), h->cells_frame, h->fast_rows, h->fast_interior, T + h->cell_tab_off, cell_begin, cell_end);
  };
  // (launch_blur_levels: k_blur_mfma where the level takes it, else k_blur_groups, else the generic LDS kernel)
  const bool lv0_rows16 = !lv0_not16 && stride >= ((D.lv[0].w + 15) & ~15);
  h->last_lv0_rows16 = lv0_rows16, h->last_lv0_unaligned = lv0_unaligned;
  auto launch_blur = [&](hipStream_t bs, int l0, int l1) { launch_blur_levels(h, S, n_frames, lv0_rows16, lv0_unaligned, bs, l0, l1); };
  // One fused pass per level (orb_level_pass.inc) where the level's geometry takes it: its FAST cells, its blurred tiles and the
  // next level's rows from ONE staged tile.  Level 0 needs 16-byte aligned caller rows for the LDS-DMA chunks.
  bool any_fused = false;
  if (h->fused)
    for (int l = 0; l < D.nlevels; l++) any_fused = any_fused || (h->lp_ok[l] && !(l == 0 && lv0_not16));
  // on-demand blur: k_describe blurs the windows it reads; no blurred plane is made (the fused pass makes its own tiles)
  const bool od = h->desc_blur == 0 && !any_fused;
  h->blur_valid = !od;
  const bool overlap = !ev && h->side != nullptr && !any_fused && !od;
  if (any_fused) {
    for (int l = 0; l < D.nlevels; l++) {
      const LevelGeom &L = D.lv[l];
      if (h->lp_ok[l] && !(l == 0 && lv0_not16)) {
        LevelPassArgs A{};
        A.src = l == 0 ? dev_images : S.pyr + L.pyr_off;
        A.s_frame_stride = l == 0 ? (long long)frame_stride : h->pyr_frame;
        A.s_pitch = l == 0 ? stride : L.pitch, A.sw = L.w, A.sh = L.h;
        A.blur = S.blur + L.blur_off, A.b_frame_stride = h->blur_frame, A.b_pitch = L.pitch;
        if (l + 1 < D.nlevels) {
          const LevelGeom &N = D.lv[l + 1];
          A.dst = S.pyr + N.pyr_off, A.d_frame_stride = h->pyr_frame, A.d_pitch = N.pitch, A.dw = N.w, A.dh = N.h;
          A.gtab = T + h->rz_gtab_off[l + 1], A.yofs = T + h->tab_off[(l + 1) * 4 + 2], A.yab = T + h->tab_off[(l + 1) * 4 + 3];
        }
        A.blk_tab = T + h->lp_tab_off[l], A.cell_tab = T + h->cell_tab_off;
        A.cell_slots = h->slots.as<uint32_t>(), A.slots_frame_stride = h->slots_frame, A.cell_count = h->cellcnt.as<int>();
        A.cells_per_frame = h->cells_frame;
        A.tile_rows = h->lp_tile_rows[l], A.score_rows = h->lp_score_rows[l], A.list_cap = h->lp_list_cap[l];
        A.ini_th = D.ini_th, A.min_th = D.min_th;
        A.redge_x = h->lp_redge_x[l], A.redge_j0 = h->lp_redge_j0[l], A.redge_selA = h->lp_selA[l], A.redge_selB = h->lp_selB[l], A.redge_selP = h->lp_selP[l];
        const dim3 grid(h->lp_blocks[l], n_frames);
        if (h->lp_tp[l] == 96) hipLaunchKernelGGL(k_level_pass<96>, grid, dim3(256), h->lp_lds[l], st, A);
        else if (h->lp_tp[l] == 112) hipLaunchKernelGGL(k_level_pass<112>, grid, dim3(256), h->lp_lds[l], st, A);
        else hipLaunchKernelGGL(k_level_pass<128>, grid, dim3(256), h->lp_lds[l], st, A);
      } else {  // this level through the separate kernels
        if (l + 1 < D.nlevels) launch_resize(l + 1);
        launch_fast(L.cellBase, L.cellBase + L.nCells);
        launch_blur(st, l, l + 1);
      }
    }
    VO_STAGE_MARK(1);  // (instrumented mode: stage 0 carries the whole chain of level passes, stages 1 and 4 are empty)
  } else {
    // Outside the instrumented mode the work that needs the caller's image only -- FAST on the cells of level 0 (a third of
    // all cells) and level 0's blur -- starts on the side stream at once, next to the resize chain (seven dependent
    // launches, the small ones latency-bound); the blur of the other levels follows there when the pyramid is complete.
    // FAST -> oct-tree -> offsets do not touch the blurred planes: the side stream joins before the descriptors, level 0's
    // cells before the oct-tree.
    const bool early0 = overlap && h->early_level0 && D.nlevels > 1 && D.lv[0].nCells > 0;
    if (early0) {
      VO_HIP_CHECK(hipEventRecord(h->ev_fork0, st));
      VO_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork0, 0));
      hipStream_t keep = st;
      st = h->side;  // (launch_fast launches on `st`)
      launch_fast(D.lv[0].cellBase, D.lv[0].cellBase + D.lv[0].nCells);
      st = keep;
      VO_HIP_CHECK(hipEventRecord(h->ev_fast0, h->side));
      launch_blur(h->side, 0, 1);
    }
    for (int l = 1; l < D.nlevels; l++) launch_resize(l);
    VO_STAGE_MARK(1);
    if (overlap) {
      VO_HIP_CHECK(hipEventRecord(h->ev_fork, st));
      VO_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
      launch_blur(h->side, early0 ? 1 : 0, D.nlevels);
      VO_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
    }
    if (early0) {
      launch_fast(D.lv[1].cellBase, h->cells_frame);
      VO_HIP_CHECK(hipStreamWaitEvent(st, h->ev_fast0, 0));
    } else {
      launch_fast(0, h->cells_frame);
    }
  }
  VO_STAGE_MARK(2);
  if (h->oct_small)
    hipLaunchKernelGGL(k_octree<256>, dim3(n_frames, D.nlevels), dim3(256), 0, st, D, h->slots.as<uint32_t>(),
                       h->slots_frame, h->cellcnt.as<int>(), h->cells_frame, h->keydata.as<uint32_t>(),
                       h->keylabel.as<unsigned short>(), h->keys_frame, h->candcnt.as<int>(),
                       h->sel.as<uint32_t>(), h->sel_frame, h->nk.as<int>(), h->err.as<int>());
  else
    hipLaunchKernelGGL(k_octree<kMaxList>, dim3(n_frames, D.nlevels), dim3(256), 0, st, D, h->slots.as<uint32_t>(),
                       h->slots_frame, h->cellcnt.as<int>(), h->cells_frame, h->keydata.as<uint32_t>(),
                       h->keylabel.as<unsigned short>(), h->keys_frame, h->candcnt.as<int>(),
                       h->sel.as<uint32_t>(), h->sel_frame, h->nk.as<int>(), h->err.as<int>());
  VO_STAGE_MARK(3);
  // (stage 3, the per-frame offsets and counts, is computed by k_describe itself: no launch and no event here)
  if (overlap)
    VO_HIP_CHECK(hipStreamWaitEvent(st, h->ev_join, 0));
  else if (!any_fused && !od)
    launch_blur(st, 0, D.nlevels);
  VO_STAGE_MARK(5);
  const int kp_blocks = (std::min(capacity, h->max_kp) + 63) / 64;
  if (kp_blocks == 0 && dcounts) VO_HIP_CHECK(hipMemsetAsync(dcounts, 0, (size_t)n_frames * sizeof(int), st));
  if (kp_blocks > 0 && od) {
    int groups = 0;  // 16-slot groups of the levels' selections
    for (int l = 0; l < D.nlevels; l++) groups += (D.lv[l].capSel + kDwWaves * kDwKpw - 1) / (kDwWaves * kDwKpw);
    auto kd = k_describe_win;
    hipLaunchKernelGGL(kd, dim3(groups * ((n_frames + 7) / 8) * 8), dim3(64 * kDwWaves), 0, st, D, S, h->sel.as<uint32_t>(), h->sel_frame,
                       h->nk.as<int>(), dcounts, capacity, dkp, ddesc, groups, n_frames, h->err.as<int>(), T + h->od_tab_off);
  } else if (kp_blocks > 0) {
    hipLaunchKernelGGL(k_describe<kDescNK>, dim3(kp_blocks * ((n_frames + 7) / 8) * 8), dim3(256), 0, st, D, S, h->sel.as<uint32_t>(),
                       h->sel_frame, h->nk.as<int>(), dcounts, capacity, dkp, ddesc, lv0_not16, kp_blocks, n_frames, h->err.as<int>());
  }
  VO_STAGE_MARK(6);
#undef VO_STAGE_MARK
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

}  // namespace

extern "C" {

int vo_orb_create(vo_orb **out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
  if (!out || nfeatures < 1 || nlevels < 1 || nlevels > kMaxLevels || !(scale_factor > 1.0f) || ini_th < 0 ||
      min_th < 0 || ini_th > 255 || min_th > 255) {
    vo::set_error("vo_orb_create: invalid argument");
    return VO_ERR_INVALID;
  }
  VO_CHECK(vo::ensure_device());
  vo_orb *h = new vo_orb();
  h->nfeatures = nfeatures;
  h->nlevels = nlevels;
  h->ini_th = ini_th;
  h->min_th = min_th;
  h->scale_factor = scale_factor;
  // ORBextractor::ORBextractor, ORBextractor.cpp:414-476.  `scaleFactor` is a double member
  // initialised from the float argument (ORBextractor.h:101).
  const double sf = (double)scale_factor;
  h->scale[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) h->scale[i] = (float)(h->scale[i - 1] * sf);
  for (int i = 0; i < nlevels; i++) h->inv_scale[i] = 1.0f / h->scale[i];
  const float factor = (float)(1.0f / sf);
  float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    h->quota[l] = cv_round_f(desired);
    sum += h->quota[l];
    desired *= factor;
  }
  h->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
  // circular patch rows, :457-475
  const int vmax = cv_floor_f(kHalfPatch * sqrtf(2.f) / 2 + 1);
  const int vmin = (int)ceilf(kHalfPatch * sqrtf(2.f) / 2);
  for (int v = 0; v <= vmax; ++v) h->umax[v] = (int)lrint(sqrt((double)kHalfPatch * kHalfPatch - v * v));
  for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
    h->umax[v] = v0;
    ++v0;
  }
  {
    // disc pixels of IC_Angle (:79-107) as an explicit list; padding entries (0,0) contribute u*I = v*I = 0
    int8_t disc[768 * 2];
    memset(disc, 0, sizeof(disc));
    int nd = 0;
    for (int v = -kHalfPatch; v <= kHalfPatch; v++)
      for (int u = -h->umax[v < 0 ? -v : v]; u <= h->umax[v < 0 ? -v : v]; u++) {
        disc[2 * nd] = (int8_t)u, disc[2 * nd + 1] = (int8_t)v;
        nd++;
      }
    if (nd > 768) {
      vo::set_error("orientation disc has %d pixels (table holds 768)", nd);
      delete h;
      return VO_ERR_INVALID;
    }
    static uint32_t tab[64][36];
    memset(tab, 0, sizeof(tab));
    for (int ln = 0; ln < 64; ln++) {
      for (int i = 0; i < 12; i++) {
        const int u = disc[2 * (ln + 64 * i)], v = disc[2 * (ln + 64 * i) + 1];
        tab[ln][i] = (uint32_t)((v + kHalfPatch) * 64 + (u + kHalfPatch));
        tab[ln][12 + i / 4] |= (uint32_t)(uint8_t)(int8_t)u << (8 * (i % 4));
        tab[ln][15 + i / 4] |= (uint32_t)(uint8_t)(int8_t)v << (8 * (i % 4));
      }
      for (int k = 0; k < 4; k++)
        for (int c = 0; c < 4; c++) {
          const float fv = (float)h_pattern[4 * (64 * k + ln) + c];
          memcpy(&tab[ln][18 + 4 * k + c], &fv, 4);
        }
    }
    static uint32_t tab48[64][12];
    for (int ln = 0; ln < 64; ln++)
      for (int i = 0; i < 12; i++) {
        const int u = disc[2 * (ln + 64 * i)], v = disc[2 * (ln + 64 * i) + 1];
        tab48[ln][i] = (uint32_t)((v + 22) * kDwPitch + (u + 22));
      }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_desc_tab), tab, sizeof(tab)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(c_disc48), tab48, sizeof(tab48)) != hipSuccess) {
      vo::set_error("hipMemcpyToSymbol(orientation / pattern table) failed");
      delete h;
      return VO_ERR_HIP;
    }
  }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    vo::set_error("hipStreamCreate failed");
    delete h;
    return VO_ERR_HIP;
  }
  h->own_stream = true;
  if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork0, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fast0, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    h->side = nullptr;  // no overlap then
  }
  *out = h;
  return VO_OK;
}

void vo_orb_destroy(vo_orb *h) {
  if (!h) return;
  (void)hipStreamSynchronize(h->stream);
  for (vo::DevBuf *b : {&h->tables, &h->pyr, &h->blur, &h->slots, &h->cellcnt, &h->keydata, &h->keylabel,
                        &h->candcnt, &h->sel, &h->nk, &h->err, &h->in_img, &h->out_kp, &h->out_desc,
                        &h->out_cnt})
    b->release();
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_fork0) (void)hipEventDestroy(h->ev_fork0);
  if (h->ev_fast0) (void)hipEventDestroy(h->ev_fast0);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int vo_orb_set_stream(vo_orb *h, void *s) {
  if (!h) return VO_ERR_INVALID;
  (void)hipStreamSynchronize(h->stream);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  h->own_stream = false;
  h->stream = (hipStream_t)s;
  return VO_OK;
}

int vo_orb_set_option(vo_orb *h, int option, int value) {
  if (!h) return VO_ERR_INVALID;
  if (option == VO_ORB_OPT_FUSED_LEVEL_PASS) {
    h->fused = value != 0 ? 1 : 0;
    return VO_OK;
  }
  if (option == VO_ORB_OPT_EARLY_LEVEL0) {
    h->early_level0 = value ? 1 : 0;
    return VO_OK;
  }
  if (option == VO_ORB_OPT_DESCRIBE_BLUR) {
    if (value != 0 && value != 1) {
      vo::set_error("vo_orb_set_option(VO_ORB_OPT_DESCRIBE_BLUR): 0 (on demand, inside the descriptor kernel) or 1 (blurred planes)");
      return VO_ERR_INVALID;
    }
    h->desc_blur = value;
    return VO_OK;
  }
  if (option == VO_ORB_OPT_BLUR_KERNEL) {
    if (value != 0 && value != 1) {
      vo::set_error("vo_orb_set_option(VO_ORB_OPT_BLUR_KERNEL): 0 (matrix cores) or 1 (VALU)");
      return VO_ERR_INVALID;
    }
    h->blur_mfma = value;
    return VO_OK;
  }
  vo::set_error("vo_orb_set_option: unknown option %d", option);
  return VO_ERR_INVALID;
}

int vo_orb_set_stage_hook(vo_orb *h, vo_orb_stage_hook hook, void *user) {
  if (!h) return VO_ERR_INVALID;
  h->hook = hook, h->hook_user = user;
  return VO_OK;
}

int vo_orb_debug_level_pass(vo_orb *h, int width, int height, int level, int out[8]) {
  if (!h || !out || level < 0 || level >= h->nlevels || width <= 0 || height <= 0) return VO_ERR_INVALID;
  VO_CHECK(configure(h, width, height, 1));
  out[0] = h->fused && h->lp_ok[level] ? 1 : 0, out[1] = h->lp_tp[level], out[2] = h->lp_tile_rows[level], out[3] = h->lp_score_rows[level];
  out[4] = h->lp_blocks[level], out[5] = (int)h->lp_lds[level], out[6] = h->lp_list_cap[level], out[7] = 0;
  return VO_OK;
}

int vo_orb_levels(const vo_orb *h) { return h ? h->nlevels : 0; }
float vo_orb_scale_factor(const vo_orb *h) { return h ? h->scale_factor : 0.f; }
int vo_orb_scale_factors(const vo_orb *h, float *s, float *is) {
  if (!h || !s) return VO_ERR_INVALID;
  for (int i = 0; i < h->nlevels; i++) {
    s[i] = h->scale[i];
    if (is) is[i] = h->inv_scale[i];
  }
  return VO_OK;
}
int vo_orb_features_per_level(const vo_orb *h, int *q) {
  if (!h || !q) return VO_ERR_INVALID;
  for (int i = 0; i < h->nlevels; i++) q[i] = h->quota[i];
  return VO_OK;
}
int vo_orb_max_keypoints(const vo_orb *h) {
  if (!h) return 0;
  // after the first call the exact per-level capacities are known (max(quota + 4, 4 * nIni), configure());
  // before it the bound covers root-node counts up to nIni = 8 (images up to ~8.5 : 1)
  if (h->cfg_w > 0) return h->max_kp;
  int s = 0;
  for (int i = 0; i < h->nlevels; i++) s += std::max(h->quota[i] + 4, 32);
  return s;
}

int vo_orb_sync(vo_orb *h) {
  if (!h) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  int e = 0;
  if (h->err.p) {
    VO_HIP_CHECK(hipMemcpy(&e, h->err.p, 4, hipMemcpyDeviceToHost));
    if (e) VO_HIP_CHECK(hipMemsetAsync(h->err.p, 0, 4, h->stream));  // reported once (on the handle's stream: ordered with its kernels)
  }
  if (e) {
    vo::set_error(e == 1   ? "more FAST candidates on one level than the 65535-key scratch holds"
                  : e == 2 ? "oct-tree node list overflow"
                           : "more key-points in a frame than the caller's capacity (see vo_orb_max_keypoints)");
    return VO_ERR_CAPACITY;
  }
  return VO_OK;
}

int vo_orb_extract_batch_dev(vo_orb *h, const uint8_t *dev_images, int n_frames, int width, int height,
                             int stride, size_t frame_stride_bytes, vo_keypoint *dev_keypoints,
                             uint8_t *dev_descriptors, int capacity, int32_t *dev_counts) {
  if (!h || !dev_images || n_frames < 1 || width < 1 || height < 1 || stride < width || !dev_keypoints ||
      !dev_descriptors || capacity < 1) {
    vo::set_error("vo_orb_extract_batch_dev: invalid argument");
    return VO_ERR_INVALID;
  }
  return run_pipeline(h, dev_images, n_frames, width, height, stride, frame_stride_bytes, dev_keypoints,
                      dev_descriptors, capacity, dev_counts);
}

int vo_orb_extract(vo_orb *h, const uint8_t *image, int width, int height, int stride, vo_keypoint *keypoints,
                   uint8_t *descriptors, int capacity, int *n_keypoints) {
  if (!h) return VO_ERR_INVALID;
  if (!image || width <= 0 || height <= 0) return VO_OK;  // `if(_image.empty()) return;` :1054-1055
  if (!keypoints || !descriptors || !n_keypoints || capacity < 1 || stride < width) {
    vo::set_error("vo_orb_extract: invalid argument");
    return VO_ERR_INVALID;
  }
  const int pitch = align_up(width, 64);
  VO_CHECK(h->in_img.reserve((size_t)pitch * height));
  VO_CHECK(h->out_kp.reserve((size_t)capacity * sizeof(vo_keypoint)));
  VO_CHECK(h->out_desc.reserve((size_t)capacity * 32));
  VO_CHECK(h->out_cnt.reserve(64));
  VO_HIP_CHECK(hipMemcpy2DAsync(h->in_img.p, pitch, image, stride, width, height, hipMemcpyHostToDevice, h->stream));
  VO_CHECK(run_pipeline(h, h->in_img.as<uint8_t>(), 1, width, height, pitch, (size_t)pitch * height,
                        h->out_kp.as<vo_keypoint>(), h->out_desc.as<uint8_t>(), capacity, h->out_cnt.as<int32_t>()));
  int n = 0;
  VO_HIP_CHECK(hipMemcpyAsync(&n, h->out_cnt.p, 4, hipMemcpyDeviceToHost, h->stream));
  VO_CHECK(vo_orb_sync(h));
  if (n > 0) {
    VO_HIP_CHECK(hipMemcpy(keypoints, h->out_kp.p, (size_t)n * sizeof(vo_keypoint), hipMemcpyDeviceToHost));
    VO_HIP_CHECK(hipMemcpy(descriptors, h->out_desc.p, (size_t)n * 32, hipMemcpyDeviceToHost));
  }
  *n_keypoints = n;
  return VO_OK;
}

int vo_orb_set_timing(vo_orb *h, int enabled) {
  if (!h) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  h->timing = enabled != 0;
  h->timed_calls = 0;
  return VO_OK;
}

int vo_orb_get_timing(vo_orb *h, double *ms, int *n_calls) {
  if (!h || !ms || !n_calls) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int c = 0; c < h->timed_calls; c++) {
    hipEvent_t *ev = h->ev.data() + (size_t)c * (VO_ORB_STAGES + 1);
    for (int s = 0; s < VO_ORB_STAGES; s++) {
      if (s == 3) continue;  // folded into the descriptor kernel; event 4 is not recorded
      float t = 0;
      VO_HIP_CHECK(hipEventElapsedTime(&t, ev[s == 4 ? 3 : s], ev[s + 1]));
      ms[s] += t;
    }
  }
  *n_calls = h->timed_calls;
  h->timed_calls = 0;
  return VO_OK;
}

int vo_orb_get_level(vo_orb *h, int frame, int level, int blurred, uint8_t *dst, int dst_stride, int *width,
                     int *height) {
  if (!h || frame < 0 || frame >= h->last_frames || level < 0 || level >= h->nlevels) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  const LevelGeom &L = h->dev.lv[level];
  if (width) *width = L.w;
  if (height) *height = L.h;
  if (!dst) return VO_OK;
  const uint8_t *sp;
  int pitch;
  if (blurred) {
    if (!h->blur_valid) {  // on-demand mode: the extraction did not make the planes; make them now (all levels of the last batch)
      launch_blur_levels(h, h->last_src, h->last_frames, h->last_lv0_rows16, h->last_lv0_unaligned, h->stream, 0, h->nlevels);
      VO_HIP_CHECK(hipGetLastError());
      VO_HIP_CHECK(hipStreamSynchronize(h->stream));
      h->blur_valid = true;
    }
    sp = h->last_src.blur + (long long)frame * h->last_src.blur_frame_stride + L.blur_off;
    pitch = L.pitch;
  } else if (level == 0) {
    sp = h->last_src.img0 + (long long)frame * h->last_src.img0_frame_stride;
    pitch = h->last_src.img0_pitch;
  } else {
    sp = h->last_src.pyr + (long long)frame * h->last_src.pyr_frame_stride + L.pyr_off;
    pitch = L.pitch;
  }
  if (blurred) {  // the blurred planes are stored in 16 x 8 tiles: fetch the plane, un-tile on the host
    std::vector<uint8_t> tmp((size_t)pitch * align_up(L.h, 8));
    VO_HIP_CHECK(hipMemcpy(tmp.data(), sp, tmp.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; y++)
      for (int x = 0; x < L.w; x++) dst[(size_t)y * dst_stride + x] = tmp[(size_t)blur_tiled_off(x, y, pitch)];
    return VO_OK;
  }
  VO_HIP_CHECK(hipMemcpy2D(dst, dst_stride, sp, pitch, L.w, L.h, hipMemcpyDeviceToHost));
  return VO_OK;
}

int vo_orb_get_candidates(vo_orb *h, int frame, int level, float *x, float *y, float *response, int capacity,
                          int *n) {
  if (!h || frame < 0 || frame >= h->last_frames || level < 0 || level >= h->nlevels || !n) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  int cnt = 0;
  VO_HIP_CHECK(hipMemcpy(&cnt, h->candcnt.as<int>() + frame * h->nlevels + level, 4, hipMemcpyDeviceToHost));
  *n = cnt;
  const int m = std::min(cnt, capacity);
  if (m <= 0 || !x || !y || !response) return VO_OK;
  std::vector<uint32_t> tmp(m);
  VO_HIP_CHECK(hipMemcpy(tmp.data(),
                         h->keydata.as<uint32_t>() + (long long)frame * h->keys_frame + h->dev.lv[level].candBase,
                         (size_t)m * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < m; i++) {
    x[i] = (float)(tmp[i] & 0xfff);
    y[i] = (float)((tmp[i] >> 12) & 0xfff);
    response[i] = (float)(tmp[i] >> 24);
  }
  return VO_OK;
}

int vo_orb_get_level_counts(vo_orb *h, int frame, int32_t *counts) {
  if (!h || frame < 0 || frame >= h->last_frames || !counts) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(h->stream));
  VO_HIP_CHECK(hipMemcpy(counts, h->nk.as<int>() + frame * h->nlevels, (size_t)h->nlevels * 4, hipMemcpyDeviceToHost));
  return VO_OK;
}

}  // extern "C"

const int *vo::orb_error_flag(const vo_orb *h) { return h ? h->err.as<int>() : nullptr; }
