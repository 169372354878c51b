// guided.hip -- device-resident frame features and the guided (window) matchers on gfx950.
//
// Replaces, on the device:
//   Frame::undistortKeyPoints / findDepth / assignFeaturesToGrid   reference src/frame.cpp:36-133
//   Frame::getFeaturesInArea / KeyFrame::getFeaturesInArea          frame.cpp:199-247, keyframe.cpp:268-312
//   the candidate loops of Matcher::searchByProjection (x4), fuseMapPoints, fuseByPose, searchBySim3
//                                                                   src/matcher.cpp:18-447, 679-865, 1012-1238
//
// A `vo_frames` object keeps the features of a batch of frames in HBM (undistorted key-points, uRight,
// depth, descriptors and the 64 x 48 grid as CSR), so key-points never leave the device between
// vo_orb_extract_batch_dev and the matcher.  Matching is two kernels:
//   k_guided_cand<G> G lanes per query (8 for the narrow tracking windows, 16 for wide ones), 256 / G queries per
//                    workgroup.  The store keeps a second, CELL-ORDERED copy of what a window search reads (16-byte
//                    records x, y, uright, octave | index << 8, and the descriptors in the same order), so the grid
//                    columns of a window are contiguous record runs, walked G records at a time: a record that
//                    passes the radius / octave / stereo gates costs one 32-byte descriptor load and 8 v_xor + v_bcnt,
//                    and lands -- by ballot / mbcnt within the group -- in the query's fixed 32-record slot as
//                    index | distance << 14 | octave << 23 | rotation bin << 27.  Queries with more candidates
//                    continue in a per-frame overflow pool claimed by one atomic per query; an exhausted pool raises
//                    the handle's sticky error flag (VO_ERR_CAPACITY at the next status call), never truncation.
//                    Searches without a claim step (fuse, area-best) finish here with a group-wide arg-min.
//   k_guided_replay  one wavefront per frame replays the order-dependent part of the reference's loop -- "features
//                    claimed earlier in the call are skipped" (:87, :218, :314, :422), a later non-blocked claim
//                    overwrites the assignment (:110-128) -- 64 QUERIES PER STEP, lane = query, its 32 records in
//                    registers: every lane proposes its claim (best and, for the ratio test, runner-up among its
//                    unblocked records) from the blocked[] state the previous steps left in LDS; blocking proposals
//                    are published by ds_min (earliest lane per feature); a lane is stale if an earlier lane of the
//                    step blocks a feature of its list; the lanes in front of the first stale one commit, the rest
//                    re-propose.  Frames with an overflowing window and the Sim3 search (whose :422 quirk indexes
//                    blocked[] by candidate rank) take the four-queries-per-step serial form.  Tail: rotation
//                    histogram and its three-maxima pruning (:128-145, computeThreeMax :1258-1304).
// Match pairs are bit-identical to the sequential reference loop for any input.
//
// Compiled with -ffp-contract=off: the float gates must round like the x86-64 reference build.
#include "vo_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

using namespace vo;

constexpr int kGridCols = 64, kGridRows = 48;  // FRAME_GRID_COLS/ROWS, camera.h:8-9
constexpr int kCells = kGridCols * kGridRows;
constexpr int TH_HIGH = 100, TH_LOW = 50;      // matcher.cpp:11-12
constexpr int HISTO_LENGTH = 30;               // :13
constexpr int kMaxFeat = 16384;                // features per frame (LDS arrays of the replay kernel)

enum GuidedMode {
  kModeFrame = 0,     // searchByProjection(Frame*, Frame*)        :18-148
  kModeLocalMap = 1,  // searchByProjection(Frame*, MapPoints)     :274-353
  kModeKeyFrame = 2,  // searchByProjection(Frame*, KeyFrame*)     :150-272
  kModeFuse = 3,      // fuseMapPoints candidate search            :1064-1106
  kModeArea = 4,      // searchBySim3 / fuseByPose inner search    :756-786, :1196-1213
  kModeSim3 = 5       // searchByProjection(KeyFrame*, Sim3&)      :356-447 (Q-M1)
};

struct FramesDev {
  int cap;  // feature slots per frame
  float *x, *y, *angle, *uright, *depth;
  int *octave;
  uint8_t *desc;
  int *n;              // [frames]
  int *cell_start;     // [frames][kCells + 1]
  unsigned short *cell_items;  // [frames][cap]
  // the same features in CELL ORDER (the order of cell_items): the window of a query is a handful of contiguous runs
  // of these arrays, so the matcher reads its candidates with coalesced loads and one round trip
  uint4 *srec;         // [frames][cap]  x, y, uRight (float bits), octave | index << 8
  uint4 *sdesc;        // [frames][cap][2]
  float xmin, ymin, gw, gh;
};

struct CamDev {
  float fx, fy, cx, cy, bf;
  double k[5];  // k1 k2 p1 p2 k3
  int distorted;
};

// ------------------------------------------------------------------------------------------
// N1  frame post-processing.  One thread per key-point:
//   undistortKeyPoints (frame.cpp:36-70): cv::undistortPoints(mat, mat, K, distCoef, Mat(), K) -- OpenCV 3.x
//     cvUndistortPoints: normalise with 1/fx, 1/fy, five fixed-point iterations of the inverse distortion
//     model in double, re-project with K, round to float; skipped when k1 == 0 (:41-45)
//   findDepth (:108-133): d = depthImg.at<float>(v, u) at the ORIGINAL key-point (float -> int truncation),
//     uRight = undistorted x - bf / d when d > 0
//   grid cell of assignFeaturesToGrid (:72-89): round((x - xMin) * gridPerPixelWidth)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_frame_post(FramesDev F, CamDev C, const vo_keypoint *kps, const uint8_t *desc,
                                                    const int *counts, int kp_capacity, const void *depth, int depth_kind,
                                                    long long depth_frame_stride, int depth_pitch, float inv_depth_scale,
                                                    int img_w, int img_h, int slot0, int *err) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int n = min(counts[f], min(kp_capacity, F.cap));
  if (i == 0) {
    F.n[slot0 + f] = n;
    if (counts[f] > n) atomicOr(err, 2);  // key-points beyond the store's slots are dropped: never silently (sticky flag)
  }
  if (i >= n) return;
  const vo_keypoint kp = kps[(long long)f * kp_capacity + i];
  float ux = kp.x, uy = kp.y;
  if (C.distorted) {
    const double fx = (double)C.fx, fy = (double)C.fy, cx = (double)C.cx, cy = (double)C.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = ((double)kp.x - cx) * ifx, y = ((double)kp.y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((C.k[4] * r2 + C.k[1]) * r2 + C.k[0]) * r2);
      const double deltaX = 2 * C.k[2] * x * y + C.k[3] * (r2 + 2 * x * x);
      const double deltaY = C.k[2] * (r2 + 2 * y * y) + 2 * C.k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1.0);
    ux = (float)(xx * ww);
    uy = (float)(yy * ww);
  }
  const long long o = (long long)(slot0 + f) * F.cap + i;
  float d = -1.f, ur = -1.f;
  if (depth_kind) {
    const int px = min(max((int)kp.x, 0), img_w - 1), py = min(max((int)kp.y, 0), img_h - 1);
    float dv;
    if (depth_kind == 1)
      dv = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(depth) + f * depth_frame_stride + (long long)py * depth_pitch)[px];
    else  // 16-bit raw depth, Mat::convertTo(CV_32F, 1 / depthScale) (visualOdometry.cpp:162-163): float multiply
      dv = (float)reinterpret_cast<const unsigned short *>(reinterpret_cast<const uint8_t *>(depth) + f * depth_frame_stride + (long long)py * depth_pitch)[px] * inv_depth_scale;
    if (dv > 0) {
      d = dv;
      ur = ux - C.bf / dv;
    }
  }
  F.x[o] = ux, F.y[o] = uy, F.angle[o] = kp.angle, F.octave[o] = kp.octave, F.uright[o] = ur, F.depth[o] = d;
  const uint4 *ds = reinterpret_cast<const uint4 *>(desc + ((long long)f * kp_capacity + i) * 32);
  uint4 *dd = reinterpret_cast<uint4 *>(F.desc + o * 32);
  dd[0] = ds[0], dd[1] = ds[1];
}

// assignFeaturesToGrid (frame.cpp:72-89) as CSR: one workgroup per frame.  Cell lists keep feature-index order
// (push_back order): slots are handed out by LDS atomics, then every cell with more than one item is sorted.
// LDS_ITEMS (the frame's slots fit kGridLdsCap): every feature's cell and the item list stay in LDS until the list is
// final -- the insertion sort of a cell was a chain of dependent global loads and stores (25 of the 0.59 ms of a
// single-stream frame, 0.11 ms per 1024 frames: the kernel ran at 7 % of the issue rate).
constexpr int kGridLdsCap = 4096;
template <bool LDS_ITEMS>
__global__ __launch_bounds__(256) void k_frame_grid(FramesDev F, int slot0) {
  __shared__ int cnt[kCells];
  __shared__ int wsum[4];
  __shared__ unsigned short s_items[LDS_ITEMS ? kGridLdsCap : 1];
  __shared__ short s_cell[LDS_ITEMS ? kGridLdsCap : 1];
  const int s = slot0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = F.n[s];
  const float *X = F.x + (long long)s * F.cap, *Y = F.y + (long long)s * F.cap;
  int *start = F.cell_start + (long long)s * (kCells + 1);
  unsigned short *gitems = F.cell_items + (long long)s * F.cap;
  unsigned short *items = LDS_ITEMS ? s_items : gitems;
  for (int c = tid; c < kCells; c += 256) cnt[c] = 0;
  __syncthreads();
  auto cell_of = [&](int i) {
    const int gx = (int)roundf((X[i] - F.xmin) * F.gw), gy = (int)roundf((Y[i] - F.ymin) * F.gh);
    return (gx < 0 || gx >= kGridCols || gy < 0 || gy >= kGridRows) ? -1 : gx * kGridRows + gy;
  };
  for (int i = tid; i < n; i += 256) {
    const int c = cell_of(i);
    if (LDS_ITEMS) s_cell[i] = (short)c;
    if (c >= 0) atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  // exclusive scan of the 3072 counts: 12 per thread
  constexpr int PER = kCells / 256;
  int loc[PER], sum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) loc[k] = cnt[tid * PER + k], sum += loc[k];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < wave; w++) base += wsum[w];
  __syncthreads();
  int first[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    first[k] = base;
    start[tid * PER + k] = base;
    cnt[tid * PER + k] = base;  // fill cursor
    base += loc[k];
  }
  if (tid == 255) start[kCells] = base;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const int c = LDS_ITEMS ? (int)s_cell[i] : cell_of(i);
    if (c >= 0) items[atomicAdd(&cnt[c], 1)] = (unsigned short)i;
  }
  __syncthreads();
  if (!LDS_ITEMS) __threadfence_block();
  // a thread sorts the cells whose counts it scanned (it still holds their extents)
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int s0 = first[k], e = s0 + loc[k];
    for (int a2 = s0 + 1; a2 < e; a2++) {  // insertion sort, lists of a handful of entries
      const unsigned short v = items[a2];
      int b2 = a2 - 1;
      while (b2 >= s0 && items[b2] > v) items[b2 + 1] = items[b2], b2--;
      items[b2 + 1] = v;
    }
  }
  __syncthreads();
  if (!LDS_ITEMS) __threadfence_block();
  // the list, and the cell-ordered copies for the matcher
  const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];  // = start[kCells]
  const long long fo = (long long)s * F.cap;
  for (int t = tid; t < total; t += 256) {
    const int i = items[t];
    if (LDS_ITEMS) gitems[t] = (unsigned short)i;
    F.srec[fo + t] = make_uint4(__float_as_uint(X[i]), __float_as_uint(Y[i]), __float_as_uint(F.uright[fo + i]),
                                (unsigned)(F.octave[fo + i] & 0xff) | ((unsigned)i << 8));
    const uint4 *d = reinterpret_cast<const uint4 *>(F.desc + (fo + i) * 32);
    F.sdesc[2 * (fo + t)] = d[0], F.sdesc[2 * (fo + t) + 1] = d[1];
  }
}

// ------------------------------------------------------------------------------------------
// guided matching
// ------------------------------------------------------------------------------------------
struct Queries {  // device arrays, query q of frame f at f * stride + q
  const uint8_t *flags;
  const float *u, *v, *aux;   // aux: 1/z (frame search) or projected uRight (local map, fuse)
  const int *level;           // last octave (frame search) or predicted level
  const float *angle, *viewcos;
  const uint8_t *desc;
  const int *nq;              // per frame, or NULL: nq_all
  int nq_all, stride;
};

struct GuidedParams {
  int mode;
  float radius, bf, ratio, dist_threshold;
  int direction, check_rot, n_levels, max_dist;
  float sf[16];
  int retry_below;  // with GuidedOut::retry_nq: frames that end with fewer matches are cleared and queued for a second call
};

constexpr float kNarrowRadius = 20.f;  // search radius (before the level scale) up to which a query gets 8 lanes, not 16
constexpr int kSlot = 32;  // candidate records every query owns in the pool; the (rare) rest goes to the overflow area

struct GuidedOut {
  unsigned *pool;           // [frames][stride][kSlot] candidate records: idx | dist << 14 | octave << 23 | rot bin << 27
  unsigned short *rank;     // same shape: position in the raw window list (Sim3 search only)
  unsigned *ovf;            // [frames][ovf_stride] records beyond a query's slot
  unsigned short *ovf_rank;
  uint2 *qrec;              // [frames][stride] (count | observed << 31, overflow offset) of every query
  int *ovf_used;            // [frames]
  int ovf_stride;
  int *best_idx;            // [frames][stride]   (fuse / area searches)
  int *assigned;            // [frames][cap]      (claiming searches), in/out
  const uint8_t *fmask;     // [frames][cap] or NULL: blocked / has-map-point / occupied on entry
  unsigned *pushes;         // [frames][stride] rotation-histogram entries idx | bin << 16
  int *n_matches;           // [frames]
  int *retry_nq;            // [frames] or NULL: the n_per_frame array of a retry call (vo_guided_params::retry_n_per_frame)
  int *err;                 // bit 0: overflow area exhausted (bit 1: the frame store dropped key-points, k_frame_post)
};

__device__ __forceinline__ unsigned row_min_u32(unsigned x) {  // minimum over the 16 lanes of a DPP row, in every lane
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));
  return x;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {  // minimum over the 64 lanes, uniform
  x = row_min_u32(x);
  return min(min((unsigned)__builtin_amdgcn_readlane((int)x, 0), (unsigned)__builtin_amdgcn_readlane((int)x, 16)),
             min((unsigned)__builtin_amdgcn_readlane((int)x, 32), (unsigned)__builtin_amdgcn_readlane((int)x, 48)));
}

template <int G>
__device__ __forceinline__ unsigned group_min_u32(unsigned x) {  // minimum over the G lanes of a query, in every lane
  if (G == 16) return row_min_u32(x);
#pragma unroll
  for (int o = G / 2; o >= 1; o >>= 1) x = min(x, (unsigned)__shfl_xor((int)x, o, G));
  return x;
}

// SIXTEEN LANES per (query, frame): a window holds a dozen items on average, so a 64-lane wavefront per query
// would idle three quarters of its lanes and -- the kernel is a chain of dependent memory round trips -- need four
// times the wavefronts.  Sixteen queries per workgroup, no workgroup barrier.  Round trips per query: its fields
// -> the (start, end) of its grid columns -> the cell-ordered feature records and descriptors (coalesced: the
// window cells of one grid column are one contiguous run) -> the records out.
template <int G>  // lanes per query: 16 (wide windows) or 8 (the tracking searches: a handful of features per window;
                   // measured per 1024 frames, frame / local-map search stage: 16 lanes 0.60 / 0.61 ms, 8: 0.57 / 0.52, 4: 0.67 / 0.73,
                   // one lane per query 0.85 / 0.72 -- fewer lanes save set-up instructions but scatter the record loads)
__global__ __launch_bounds__(256) void k_guided_cand(FramesDev F, Queries Q, GuidedParams P, GuidedOut O, int slot0) {
  constexpr int kMaxCol = 64, QPB = 256 / G;
  __shared__ int s_pre[QPB][kMaxCol], s_base[QPB][kMaxCol];
  const int lane = threadIdx.x & 63, l16 = threadIdx.x & (G - 1), grp = threadIdx.x / G, sub = lane / G;
  const int f = blockIdx.y, q = blockIdx.x * QPB + grp;
  const int nq = Q.nq ? Q.nq[f] : Q.nq_all;
  if (nq < 0) return;        // frame not part of this call (uniform over the workgroup)
  const bool live = q < nq;  // (groups past the end idle through the loops: the wave's ballots need every lane)
  const long long qo = (long long)f * Q.stride + (live ? q : 0);
  const int s = slot0 + f;
  const bool claims = P.mode == kModeFrame || P.mode == kModeLocalMap || P.mode == kModeKeyFrame || P.mode == kModeSim3;
  const unsigned flags = live ? Q.flags[qo] : 0u;
  const bool valid = (flags & 1u) != 0;
  const float u = Q.u[qo], v = Q.v[qo];
  const int lv = min(max(Q.level[qo], 0), 15);
  const float aux = Q.aux ? Q.aux[qo] : 0.f;
  const float q_angle = (P.check_rot && Q.angle) ? Q.angle[qo] : 0.f;
  const uint4 *qd4 = reinterpret_cast<const uint4 *>(Q.desc + qo * 32);
  const uint4 qa = qd4[0], qb = qd4[1];
  // search radius and level range of the routine
  float rs;
  int lmin, lmax;
  if (P.mode == kModeFrame) {
    rs = P.radius * P.sf[lv];  // :66-68
    if (P.direction == 1) lmin = lv, lmax = P.n_levels;       // :70-71
    else if (P.direction == 2) lmin = 0, lmax = lv;           // :72-73
    else lmin = lv - 1, lmax = lv + 1;                        // :74-75
  } else if (P.mode == kModeLocalMap) {
    float r = (Q.viewcos ? Q.viewcos[qo] : 1.f) > 0.998 ? 2.5f : 4.0f;  // :288-291 (float vs double literal compare)
    r *= P.radius;
    rs = r * P.sf[lv];
    lmin = lv - 1, lmax = lv;
  } else if (P.mode == kModeKeyFrame) {
    rs = P.radius * P.sf[lv];
    lmin = lv - 1, lmax = lv + 1;
  } else {  // KeyFrame::getFeaturesInArea has no level filter (keyframe.cpp:268-312); the octave gate follows
    rs = P.radius * P.sf[lv];
    lmin = -(1 << 30), lmax = 1 << 30;
  }
  // window -> grid columns / rows, frame.cpp:205-223
  const int x0 = max(0, (int)floorf((u - F.xmin - rs) * F.gw));
  const int x1 = min(kGridCols - 1, (int)floorf((u - F.xmin + rs) * F.gw));
  const int y0 = max(0, (int)floorf((v - F.ymin - rs) * F.gh));
  const int y1 = min(kGridRows - 1, (int)floorf((v - F.ymin + rs) * F.gh));
  const bool window = valid && !(x0 >= kGridCols || x1 < 0 || y0 >= kGridRows || y1 < 0 || x1 < x0 || y1 < y0);
  const int *start = F.cell_start + (long long)s * (kCells + 1);
  const long long fo = (long long)s * F.cap, mo = (long long)f * F.cap;  // frame-store slot / this call's frame
  // grid column x0 + c: its window cells y0..y1 are one contiguous run of the cell-ordered arrays
  const int ncol = window ? x1 - x0 + 1 : 0;
  int T = 0;
  for (int cb = 0; cb < kMaxCol; cb += G) {
    const int c = cb + l16;
    int cbeg = 0, ccnt = 0;
    if (c < ncol) {
      cbeg = start[(x0 + c) * kGridRows + y0];
      ccnt = start[(x0 + c) * kGridRows + y1 + 1] - cbeg;
    }
    int incl = ccnt;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const int t = __shfl_up(incl, o, G);
      if (l16 >= o) incl += t;
    }
    s_pre[grp][c] = T + incl - ccnt;
    s_base[grp][c] = cbeg;
    T += __shfl(incl, G - 1, G);
    if (__builtin_amdgcn_ballot_w64(cb + G < ncol) == 0ull) break;  // no group of the wave has more columns
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float ur_q = P.mode == kModeFrame ? u - P.bf * aux : aux;  // :92 / trackProj_uR_
  const float pdf = HISTO_LENGTH / 360.0f;

  struct Item { bool inraw, pass; int idx, dist, oct, bin; };
  auto eval = [&](int t) {
    Item it{false, false, 0, 511, 0, 0};
    if (t >= T) return it;
    int lo = 0, hi = ncol - 1;  // last column whose prefix is <= t
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_pre[grp][mid] <= t) lo = mid; else hi = mid - 1;
    }
    const long long pos = fo + s_base[grp][lo] + (t - s_pre[grp][lo]);
    const uint4 r = F.srec[pos];
    const uint4 da = F.sdesc[2 * pos], db = F.sdesc[2 * pos + 1];
    const float fx = __uint_as_float(r.x), fy = __uint_as_float(r.y), fur = __uint_as_float(r.z);
    const int oct = (int)(r.w & 0xffu), idx = (int)(r.w >> 8);
    it.idx = idx, it.oct = oct;
    if (oct < lmin || oct > lmax) return it;
    if (!(fabsf(fx - u) < rs && fabsf(fy - v) < rs)) return it;  // frame.cpp:238-241
    it.inraw = true;
    bool ok = true;
    if (P.mode == kModeFrame || P.mode == kModeLocalMap) {
      if (O.fmask && O.fmask[mo + idx]) ok = false;  // already blocked on entry: stays blocked (:87, :314)
      if (fur > 0 && fabsf(ur_q - fur) > rs) ok = false;  // :90-96, :317-322
    } else if (P.mode == kModeKeyFrame) {
      if (O.fmask && O.fmask[mo + idx]) ok = false;  // :218
    } else {
      if (oct < lv - 1 || oct > lv) ok = false;  // :1077, :769, :425
      if (P.mode == kModeFuse && ok) {  // chi2 gates :1084-1099
        const float ex = u - fx, ey = v - fy;
        const float is = 1.0f / P.sf[min(oct, 15)];
        if (fur >= 0) {
          const float er = aux - fur;
          const float e2 = ex * ex + ey * ey + er * er;
          if (e2 * is * is > 7.815f) ok = false;
        } else {
          const float e2 = ex * ex + ey * ey;
          if (e2 * is * is > 5.991f) ok = false;
        }
      }
    }
    it.pass = ok;
    it.dist = __popc(da.x ^ qa.x) + __popc(da.y ^ qa.y) + __popc(da.z ^ qa.z) + __popc(da.w ^ qa.w) + __popc(db.x ^ qb.x) +
              __popc(db.y ^ qb.y) + __popc(db.z ^ qb.z) + __popc(db.w ^ qb.w);
    if (ok && P.check_rot && (P.mode == kModeFrame || P.mode == kModeKeyFrame)) {  // the bin this pair would vote for (:115-125)
      float rot = q_angle - F.angle[fo + idx];
      if (rot < 0) rot += 360.0f;
      int bin = (int)rintf(rot * pdf);
      if (bin == HISTO_LENGTH) bin = 0;
      it.bin = min(max(bin, 0), 31);
    }
    return it;
  };
  const int Tmax = (int)(~wave_min_u32(~(unsigned)T));  // max over the wave (uniform trip count)
  constexpr unsigned kGroupMask = (1u << G) - 1u;

  if (!claims) {  // independent queries: arg-min in candidate order, strict < keeps the first minimum
    unsigned best = 0xffffffffu, best_idx = 0;
    for (int base = 0; base < Tmax; base += G) {
      const Item it = eval(base + l16);
      const unsigned key = it.pass ? ((unsigned)it.dist << 14) | (unsigned)(base + l16) : 0xffffffffu;  // T <= 16384
      const unsigned m = group_min_u32<G>(key);
      const unsigned widx = group_min_u32<G>(key == m ? (unsigned)it.idx : 0xffffffffu);  // the winner's feature index
      if (m < best) best = m, best_idx = widx;
    }
    const int limit = P.mode == kModeFuse ? TH_LOW : P.max_dist;
    if (live && l16 == 0) O.best_idx[qo] = (best != 0xffffffffu && (int)(best >> 14) <= limit) ? (int)best_idx : -1;
    return;
  }

  // claiming searches: records of the gated candidates, in window order, into the query's pool slot
  unsigned *slot = O.pool + ((long long)f * Q.stride + (live ? q : 0)) * kSlot;
  unsigned short *rslot = O.rank ? O.rank + ((long long)f * Q.stride + (live ? q : 0)) * kSlot : nullptr;
  int written = 0, rawbase = 0;
  for (int base = 0; base < Tmax; base += G) {
    const Item it = eval(base + l16);
    const unsigned pm = (unsigned)(__builtin_amdgcn_ballot_w64(it.pass) >> (G * sub)) & kGroupMask;
    const unsigned rm = (unsigned)(__builtin_amdgcn_ballot_w64(it.inraw) >> (G * sub)) & kGroupMask;
    const unsigned below = (1u << l16) - 1u;
    if (it.pass) {
      const int p = written + __popc(pm & below);
      if (p < kSlot) {
        slot[p] = (unsigned)it.idx | ((unsigned)it.dist << 14) | ((unsigned)(it.oct & 15) << 23) | ((unsigned)it.bin << 27);
        if (rslot) rslot[p] = (unsigned short)min(rawbase + __popc(rm & below), 65535);
      }
    }
    written += __popc(pm);
    rawbase += __popc(rm);
  }
  int ovf_off = 0;
  if (__builtin_amdgcn_ballot_w64(written > kSlot) != 0ull) {  // rare: dense windows.  The tail goes to the overflow area
    if (written > kSlot) {
      if (l16 == 0) ovf_off = atomicAdd(&O.ovf_used[f], written - kSlot);
      ovf_off = __shfl(ovf_off, 0, G);
      if (ovf_off + written - kSlot > O.ovf_stride) {
        if (l16 == 0) atomicOr(O.err, 1);
        written = kSlot;  // truncated: reported through vo_match_guided_status
      }
    }
    unsigned *ov = O.ovf + (long long)f * O.ovf_stride + ovf_off - kSlot;
    unsigned short *orv = O.ovf_rank ? O.ovf_rank + (long long)f * O.ovf_stride + ovf_off - kSlot : nullptr;
    int w2 = 0, r2 = 0;
    for (int base = 0; base < Tmax; base += G) {
      const Item it = eval(base + l16);
      const unsigned pm = (unsigned)(__builtin_amdgcn_ballot_w64(it.pass) >> (G * sub)) & kGroupMask;
      const unsigned rm = (unsigned)(__builtin_amdgcn_ballot_w64(it.inraw) >> (G * sub)) & kGroupMask;
      const unsigned below = (1u << l16) - 1u;
      if (it.pass) {
        const int p = w2 + __popc(pm & below);
        if (p >= kSlot && p < written) {
          ov[p] = (unsigned)it.idx | ((unsigned)it.dist << 14) | ((unsigned)(it.oct & 15) << 23) | ((unsigned)it.bin << 27);
          if (orv) orv[p] = (unsigned short)min(r2 + __popc(rm & below), 65535);
        }
      }
      w2 += __popc(pm);
      r2 += __popc(rm);
    }
  }
  if (live && l16 == 0) O.qrec[qo] = make_uint2((unsigned)written | ((flags >> 1) & 1u) << 31, (unsigned)ovf_off);
}

// Frame / KeyFrame::getFeaturesInArea (frame.cpp:199-247, keyframe.cpp:268-312) as an entry point of its own: the
// feature indices of a window in the reference's order (grid column by column, cells top to bottom, push_back order
// inside a cell).  Sixteen lanes per query walk the cell-ordered records; ballot compaction keeps the order.
__global__ __launch_bounds__(256) void k_features_in_area(FramesDev F, int slot, int nq, const float *qu, const float *qv,
                                                          const float *qr, const int *qlo, const int *qhi, int *out,
                                                          int max_out, int *count) {
  const int lane = threadIdx.x & 63, l16 = threadIdx.x & 15, sub = lane >> 4;
  const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = q < nq;
  const int qq = live ? q : 0;
  const float u = qu[qq], v = qv[qq], rs = qr[qq];
  const int lmin = qlo ? qlo[qq] : -(1 << 30), lmax = qhi ? qhi[qq] : (1 << 30);
  const int x0 = max(0, (int)floorf((u - F.xmin - rs) * F.gw));
  const int x1 = min(kGridCols - 1, (int)floorf((u - F.xmin + rs) * F.gw));
  const int y0 = max(0, (int)floorf((v - F.ymin - rs) * F.gh));
  const int y1 = min(kGridRows - 1, (int)floorf((v - F.ymin + rs) * F.gh));
  const bool window = live && !(x0 >= kGridCols || x1 < 0 || y0 >= kGridRows || y1 < 0);
  const int *start = F.cell_start + (long long)slot * (kCells + 1);
  const long long fo = (long long)slot * F.cap;
  int written = 0;
  const int ncol = window ? x1 - x0 + 1 : 0;
  int ncmax = ncol;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ncmax = max(ncmax, __shfl_xor(ncmax, o));
  for (int c = 0; c < ncmax; c++) {  // uniform trip count; a grid column's window cells are one contiguous run
    int cbeg = 0, cend = 0;
    if (c < ncol && y1 >= y0) cbeg = start[(x0 + c) * kGridRows + y0], cend = start[(x0 + c) * kGridRows + y1 + 1];
    int len = cend - cbeg, lenmax = len;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lenmax = max(lenmax, __shfl_xor(lenmax, o));
    for (int base = 0; base < lenmax; base += 16) {
      const int t = base + l16;
      bool hit = false;
      int idx = 0;
      if (t < len) {
        const uint4 r = F.srec[fo + cbeg + t];
        const float fx = __uint_as_float(r.x), fy = __uint_as_float(r.y);
        const int oct = (int)(r.w & 0xffu);
        idx = (int)(r.w >> 8);
        hit = !(oct < lmin || oct > lmax) && fabsf(fx - u) < rs && fabsf(fy - v) < rs;
      }
      const unsigned m = (unsigned)(__builtin_amdgcn_ballot_w64(hit) >> (16 * sub)) & 0xffffu;
      if (hit) {
        const int p = written + __popc(m & ((1u << l16) - 1u));
        if (p < max_out) out[(long long)q * max_out + p] = idx;
      }
      written += __popc(m);
    }
  }
  if (live && l16 == 0) count[q] = written;
}

// One wavefront per frame: the sequential claim replay.  LDS: blocked[cap] bytes, assigned[cap] u16.  Everything a
// decision needs rides in the records (distance, octave, rotation bin) and in the query record (count, observed flag).
// FOUR consecutive queries are replayed per step, one per 16-lane row (a query has a dozen gated candidates): each
// row finds its best (and runner-up) under the blocked state at the start of the step; a later row whose candidate
// list contains a feature that an earlier row of the same step claims is evaluated again after that claim -- so the
// result is exactly the sequential loop's, while the common step costs two DPP row minima instead of four passes.
// Claims are committed row by row in query order (assignments of a feature overwrite each other in that order).
// LDS hand-off between the lanes of ONE wavefront: its LDS operations execute in order, so all that is needed is
// that the compiler keeps them in order and that the writes have been issued.  (A wavefront-scope fence also waits
// for vmcnt(0), i.e. for every prefetched global load in flight -- that alone made the replay memory-latency bound.)
__device__ __forceinline__ void lds_handoff() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void k_guided_replay(FramesDev F, Queries Q, GuidedParams P, GuidedOut O, int slot0,
                                                      int batched_lds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rp_lds[];
  __shared__ int hist[32];
  const int lane = threadIdx.x, row = lane >> 4, l16 = lane & 15, f = blockIdx.x, s = slot0 + f;
  const int nf = F.n[s];
  const int nq = Q.nq ? Q.nq[f] : Q.nq_all;
  if (nq < 0) return;  // n_per_frame[f] < 0: the frame is not part of this call (nothing of it is read or written)
  uint8_t *blocked = rp_lds;                                                       // [capA]
  const int capA = (F.cap + 15) & ~15;
  unsigned short *asg = reinterpret_cast<unsigned short *>(rp_lds + capA);          // [cap]: query + 1, 0 = none
  int *assigned = O.assigned + (long long)f * F.cap;
  const bool sim3 = P.mode == kModeSim3;
  for (int i = lane; i < nf; i += 64) {
    blocked[i] = O.fmask ? O.fmask[(long long)f * F.cap + i] : 0;
    const int a = sim3 ? -1 : assigned[i];
    asg[i] = (unsigned short)(a + 1);
  }
  if (lane < 32) hist[lane] = 0;
  lds_handoff();
  const uint2 *qrec = O.qrec + (long long)f * Q.stride;
  const unsigned *pool = O.pool + (long long)f * Q.stride * kSlot;
  const unsigned short *rkp = O.rank ? O.rank + (long long)f * Q.stride * kSlot : nullptr;
  const unsigned *ovf = O.ovf + (long long)f * O.ovf_stride;
  const unsigned short *ovr = O.ovf_rank ? O.ovf_rank + (long long)f * O.ovf_stride : nullptr;
  unsigned *pushes = O.pushes + (long long)f * Q.stride;
  const bool rot_on = P.check_rot && (P.mode == kModeFrame || P.mode == kModeKeyFrame);
  const bool want2 = P.mode == kModeLocalMap;  // only the ratio test looks at the runner-up
  int cnt = 0, npush = 0;
  constexpr unsigned kNone = 0xffffffffu;
  // ------------------------------------------------------------------------------------------------------------
  // Batched replay (every mode but Sim3, frames whose windows all fit their slots): 64 queries per step, lane = query.
  // Every lane proposes its claim from the blocked[] state left by the previous steps; a proposal is final unless an
  // EARLIER lane of the same step proposes a blocking claim on the feature it chose (or on its runner-up, where the
  // ratio test looks at it) -- the first such lane and everything behind it is re-proposed after the lanes in front
  // have committed.
  // Conflicts are rare (a handful per frame), so a step is usually one round: ~25 steps instead of ~450 four-query
  // steps, with the same result as the one-query-at-a-time order of matcher.cpp:76-128.
  if (batched_lds && !sim3 && O.ovf_used[f] == 0) {
    int *tmpb = reinterpret_cast<int *>(rp_lds + capA + 2 * (size_t)capA);  // [cap] first blocking lane of the round
    int *tmpw = tmpb + capA;                                                  // [cap] last final claimant + 1
    for (int i = lane; i < capA; i += 64) tmpb[i] = 64, tmpw[i] = 0;
    lds_handoff();
    constexpr int R = kSlot;  // 32 records per query, all in registers
    auto fetch64 = [&](int gb, unsigned (&rc)[R], unsigned &cw) {
      const int q = min(gb + lane, max(nq - 1, 0));
      cw = gb + lane < nq ? qrec[q].x : 0u;
      const uint4 *p4 = reinterpret_cast<const uint4 *>(pool + (long long)q * kSlot);
#pragma unroll
      for (int k = 0; k < R / 4; k++) {
        const uint4 v = p4[k];
        rc[4 * k] = v.x, rc[4 * k + 1] = v.y, rc[4 * k + 2] = v.z, rc[4 * k + 3] = v.w;
      }
    };
    unsigned recN[R], cwN;
    fetch64(0, recN, cwN);
    for (int gb = 0; gb < nq; gb += 64) {
      unsigned rec[R];
#pragma unroll
      for (int k = 0; k < R; k++) rec[k] = recN[k];
      const unsigned cw = cwN;
      fetch64(gb + 64, recN, cwN);  // in flight while this step is replayed
      const int cn = (int)(cw & 0x7fffffffu), ob = (int)(cw >> 31);
      const int cmax = ~(int)wave_min_u32(~(unsigned)cn);  // uniform: chunks of 4 records beyond it are skipped
      const int blocks = (P.mode == kModeFrame || P.mode == kModeLocalMap) ? ob : 1;
      bool unresolved = cn > 0;
      while (__builtin_amdgcn_ballot_w64(unresolved) != 0ull) {  // uniform
        // 1. proposal: best (and runner-up) among the unblocked records, ties to the earlier position
        unsigned d1 = kNone, d2 = kNone, r1 = 0, r2 = 0;
#pragma unroll
        for (int c4 = 0; c4 < R / 4; c4++) {
          if (4 * c4 >= cmax) break;  // uniform
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int pos = 4 * c4 + u;
            const unsigned rcd = rec[pos];
            const int idx = min((int)(rcd & 0x3fffu), capA - 1);  // records past the count hold anything
            const bool valid = unresolved & (pos < cn) & (blocked[idx] == 0);
            const unsigned d = valid ? ((rcd >> 14) & 0x1ffu) : kNone;
            const bool lt1 = d < d1, lt2 = d < d2;
            d2 = lt1 ? d1 : (lt2 ? d : d2), r2 = lt1 ? r1 : (lt2 ? rcd : r2);
            d1 = lt1 ? d : d1, r1 = lt1 ? rcd : r1;
          }
        }
        const int best = (int)d1, bidx = (int)(r1 & 0x3fffu);
        bool accept = unresolved && d1 != kNone;
        if (accept) {
          if (P.mode == kModeFrame) accept = best <= TH_HIGH;
          else if (P.mode == kModeLocalMap) {
            accept = best <= TH_HIGH;
            if (accept && d2 != kNone) {
              const int lv1 = (int)((r1 >> 23) & 0xfu), lv2 = (int)((r2 >> 23) & 0xfu);
              if (lv1 == lv2 && (float)best > P.ratio * (float)(int)d2) accept = false;  // :344
            }
          } else if (P.mode == kModeKeyFrame) accept = (float)best <= P.dist_threshold;  // :238
          else accept = best <= TH_LOW;                                                  // :437
        }
        // 2. blocking proposals, earliest lane per feature
        const bool blocker = accept && blocks != 0;
        if (blocker) atomicMin(&tmpb[bidx], lane);
        lds_handoff();
        // 3. a lane is stale when an earlier lane of the step blocks the feature it chose -- or, where the ratio test
        //    looks at it, its runner-up: blocking any other record of its list leaves (best, runner-up) as they are.
        //    (Testing the whole list instead is also correct but makes half of the lanes stale in the frame search:
        //    ~8 records per query, 63 earlier claims among ~1000 features -- 8 rounds per step instead of 1-2.)
        bool stale = false;
        if (unresolved && d1 != kNone) {
          stale = tmpb[bidx] < lane;
          if (want2 && d2 != kNone) stale |= tmpb[(int)(r2 & 0x3fffu)] < lane;
        }
        const unsigned long long sm = __builtin_amdgcn_ballot_w64(stale);
        const int first_stale = sm ? (int)__builtin_ctzll(sm) : 64;
        const bool fin = unresolved && lane < first_stale, claim = fin && accept;
        // 4. several final lanes may claim one feature (non-blocking claims): the last one in query order owns it
        if (claim) atomicMax(&tmpw[bidx], lane + 1);
        lds_handoff();
        const unsigned long long cm = __builtin_amdgcn_ballot_w64(claim);
        if (claim) {
          if (tmpw[bidx] == lane + 1) {
            asg[bidx] = (unsigned short)(gb + lane + 1);
            blocked[bidx] = (uint8_t)blocks;
          }
          if (rot_on) {  // :115-125, in query order
            const int bin = (int)(r1 >> 27);
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
            pushes[npush + below] = (unsigned)bidx | ((unsigned)bin << 16);
            atomicAdd(&hist[bin], 1);
          }
        }
        const int ncl = (int)__popcll(cm);
        cnt += ncl;
        if (rot_on) npush += ncl;
        lds_handoff();
        if (blocker) tmpb[bidx] = 64;  // scratch back to its idle state
        if (claim) tmpw[bidx] = 0;
        unresolved = unresolved && !fin;
        lds_handoff();
      }
    }
  } else {
  // the records of a group: row r = query g0 + r, lane l16 = records l16 and 16 + l16 of its slot
  struct Group { unsigned cw, oo, rec0, rec1; unsigned short rk0, rk1; };
  auto fetch = [&](int g0) {
    // every load is issued unconditionally (clamped addresses): the record loads must not wait for the count in the
    // query record, or each fetch would cost two dependent round trips; entries past the count are masked at use
    Group g{0u, 0u, 0u, 0u, 0, 0};
    const int q = min(g0 + row, max(nq - 1, 0));
    const uint2 qr = qrec[q];
    const long long sb = (long long)q * kSlot;
    g.rec0 = pool[sb + l16], g.rec1 = pool[sb + 16 + l16];
    if (rkp) g.rk0 = rkp[sb + l16], g.rk1 = rkp[sb + 16 + l16];
    g.cw = g0 + row < nq ? qr.x : 0u, g.oo = qr.y;
    return g;
  };
  // D groups (16 queries) are in flight while D others are replayed: with one wavefront per SIMD nothing else hides
  // the ~1.5 us of a record fetch, and a group takes a fraction of that to replay
  constexpr int D = 4;
  auto replay_group = [&](const Group &g, int g0) {
    const int cn = (int)(g.cw & 0x7fffffffu), ob = (int)(g.cw >> 31);
    const int cmax = ~(int)wave_min_u32(~(unsigned)cn);
    if (cmax == 0) return;  // uniform
    const int nch = (cmax + 15) >> 4;
    unsigned pending;
    {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(cn > 0);
      pending = (unsigned)((m & 1ull) | ((m >> 15) & 2ull) | ((m >> 30) & 4ull) | ((m >> 45) & 8ull));
    }
    while (pending != 0u) {  // uniform
      const bool mine = (pending >> row) & 1u;
      unsigned k1 = kNone, k2 = kNone, r1 = 0, r2 = 0;
      for (int c = 0; c < nch; c++) {
        unsigned rec = c == 0 ? g.rec0 : g.rec1;
        unsigned short rk = c == 0 ? g.rk0 : g.rk1;
        const int pos = 16 * c + l16;
        if (c >= 2) {  // beyond the query's slot: the overflow area (dense windows only)
          rec = pos < cn ? ovf[g.oo + pos - kSlot] : 0u;
          rk = (ovr && pos < cn) ? ovr[g.oo + pos - kSlot] : (unsigned short)0;
        }
        const int idx = rec & 0x3fffu;
        const bool skip = !mine || pos >= cn || (sim3 ? (rk < nf && blocked[rk]) : blocked[idx]);  // :422 indexes by the candidate counter
        unsigned key = skip ? kNone : (((rec >> 14) & 0x1ffu) << 16) | (unsigned)pos;
        const unsigned m1 = row_min_u32(key);
        const unsigned rr1 = (unsigned)__shfl((int)rec, (lane & 48) | (int)(m1 & 15u));
        unsigned m2 = kNone, rr2 = 0;
        if (want2) {
          if (key == m1) key = kNone;
          m2 = row_min_u32(key);
          rr2 = (unsigned)__shfl((int)rec, (lane & 48) | (int)(m2 & 15u));
        }
        if (m1 < k1) {  // merge (k1, k2) with (m1, m2): all keys distinct by position
          if (k1 < m2) k2 = k1, r2 = r1; else k2 = m2, r2 = rr2;
          k1 = m1, r1 = rr1;
        } else if (m1 < k2) {
          k2 = m1, r2 = rr1;
        }
      }
      const int best = (int)(k1 >> 16), bidx = (int)(r1 & 0x3fffu);
      bool accept = k1 != kNone;
      if (accept) {
        if (P.mode == kModeFrame) accept = best <= TH_HIGH;
        else if (P.mode == kModeLocalMap) {
          accept = best <= TH_HIGH;
          if (accept && k2 != kNone) {
            const int second = (int)(k2 >> 16);
            const int lv1 = (int)((r1 >> 23) & 0xfu), lv2 = (int)((r2 >> 23) & 0xfu);
            if (lv1 == lv2 && (float)best > P.ratio * (float)second) accept = false;  // :344
          }
        } else if (P.mode == kModeKeyFrame) accept = (float)best <= P.dist_threshold;  // :238
        else accept = best <= TH_LOW;                                                  // :437
      }
      const int blocks = (P.mode == kModeFrame || P.mode == kModeLocalMap) ? ob : 1;  // what the claim writes into blocked[] (:110-113)
      // a later pending row is stale if an earlier pending row of this step blocks a feature on its list
      bool stale = false;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int a_r = __builtin_amdgcn_readlane((int)(accept && mine), 16 * r);
        const int b_r = __builtin_amdgcn_readlane(bidx, 16 * r), bl_r = __builtin_amdgcn_readlane(blocks, 16 * r);
        if (!(a_r && bl_r)) continue;  // uniform
        if (row > r && mine) {
          const unsigned key0 = sim3 ? g.rk0 : (g.rec0 & 0x3fffu), key1 = sim3 ? g.rk1 : (g.rec1 & 0x3fffu);
          if ((l16 < cn && key0 == (unsigned)b_r) || (16 + l16 < cn && key1 == (unsigned)b_r) || cn > 32) stale = true;
        }
      }
      const unsigned long long sm = __builtin_amdgcn_ballot_w64(stale);
      int first_stale = 4;
      if (sm & 0xffff000000000000ull) first_stale = 3;
      if (sm & 0x0000ffff00000000ull) first_stale = 2;
      if (sm & 0x00000000ffff0000ull) first_stale = 1;
      // commit the rows in front of the first stale one, in query order
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (r >= first_stale || !((pending >> r) & 1u)) continue;  // uniform
        const int a_r = __builtin_amdgcn_readlane((int)accept, 16 * r);
        if (a_r) {
          const int b_r = __builtin_amdgcn_readlane(bidx, 16 * r), bl_r = __builtin_amdgcn_readlane(blocks, 16 * r);
          const int bin = (int)((unsigned)__builtin_amdgcn_readlane((int)r1, 16 * r) >> 27);
          if (lane == 0) {
            asg[b_r] = (unsigned short)(g0 + r + 1);
            blocked[b_r] = (uint8_t)bl_r;
            if (rot_on) {  // :115-125
              pushes[npush] = (unsigned)b_r | ((unsigned)bin << 16);
              hist[bin]++;
            }
          }
        }
        // (plain sums: `cnt++; if (rot_on) npush++;` inside the branch made the compiler keep both counters in scratch
        //  memory and select between their addresses)
        cnt += a_r ? 1 : 0;
        npush += (a_r && rot_on) ? 1 : 0;
        pending &= ~(1u << r);
      }
      lds_handoff();
    }
  };
  Group nxt[D];
#pragma unroll
  for (int d = 0; d < D; d++) nxt[d] = fetch(4 * d);
  for (int gb = 0; gb < nq; gb += 4 * D) {
    Group cur[D];
#pragma unroll
    for (int d = 0; d < D; d++) cur[d] = nxt[d];
#pragma unroll
    for (int d = 0; d < D; d++) nxt[d] = fetch(gb + 4 * D + 4 * d);
#pragma unroll
    for (int d = 0; d < D; d++)
      if (gb + 4 * d < nq) replay_group(cur[d], gb + 4 * d);  // uniform
  }
  }  // serial replay
  lds_handoff();
  if (npush > 0) {  // computeThreeMax (:1258-1304) + pruning (:128-145)
    int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int sz = hist[i];
      if (sz > m1) m3 = m2, i3 = i2, m2 = m1, i2 = i1, m1 = sz, i1 = i;
      else if (sz > m2) m3 = m2, i3 = i2, m2 = sz, i2 = i;
      else if (sz > m3) m3 = sz, i3 = i;
    }
    if (m2 < 0.1f * (float)m1) i2 = i3 = -1;
    else if (m3 < 0.1f * (float)m1) i3 = -1;
    __threadfence_block();
    for (int base = 0; base < npush; base += 64) {
      bool drop = false;
      if (base + lane < npush) {
        const unsigned p = pushes[base + lane];
        const int bin = (int)(p >> 16);
        if (bin != i1 && bin != i2 && bin != i3) {
          drop = true;
          asg[p & 0xffffu] = 0;
        }
      }
      cnt -= __popcll(__builtin_amdgcn_ballot_w64(drop));
    }
    lds_handoff();
  }
  // trackWithMotion's retry (visualOdometry.cpp:241-245) decided where the count is known: a frame with fewer than
  // retry_below matches gives its assignments back (`fill(mappoints_, nullptr)`) and is the only kind of frame the second,
  // wider call looks at
  const bool retry = O.retry_nq && cnt < P.retry_below;
  for (int i = lane; i < nf; i += 64) assigned[i] = retry ? -1 : (int)asg[i] - 1;
  if (lane == 0) {
    O.n_matches[f] = cnt, O.ovf_used[f] = 0;  // (the cursor is back at zero for the next search: guided_launch)
    if (O.retry_nq) O.retry_nq[f] = retry ? nq : -1;
  }
}

// number of accepted queries of the searches without a claim step
__global__ __launch_bounds__(256) void k_guided_count(Queries Q, const int *best_idx, int *n_matches) {
  __shared__ int tot;
  const int f = blockIdx.x;
  const int nq = Q.nq ? Q.nq[f] : Q.nq_all;
  if (nq < 0) return;  // frame not part of this call
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  int c = 0;
  for (int q = threadIdx.x; q < nq; q += 256) c += best_idx[(long long)f * Q.stride + q] >= 0;
  if (c) atomicAdd(&tot, c);
  __syncthreads();
  if (threadIdx.x == 0) n_matches[f] = tot;
}

}  // namespace

// ============================================================================================
// host side
// ============================================================================================
namespace vo {  // csrc/track.hip
void track_scatter_launch(int n_frames, int cap, int stride, const int *fn, int slot0, const int *assigned,
                          const double *qpoints, const uint8_t *qflags, double *fpoint, uint8_t *fhas, uint8_t *fobserved,
                          hipStream_t st);
void track_gather_launch(int n_frames, int cap, const int *fn, int slot0, const float *X, const float *Y, const float *UR,
                         const int *OCT, const double *fpoint, const uint8_t *fhas, const float *sf, double *pts, double *obs,
                         double *isg, int *ranges, int *index, hipStream_t st);
void track_scatter_gather_launch(int n_frames, int cap, int stride, const int *fn, int slot0, const int *assigned,
                                 const double *qpoints, const uint8_t *qflags, double *fpoint, uint8_t *fhas, uint8_t *fobserved,
                                 const float *X, const float *Y, const float *UR, const int *OCT, const float *sf, double *pts,
                                 double *obs, double *isg, int *ranges, int *index, hipStream_t st);
}  // namespace vo

struct vo_frames {
  int max_frames = 0, cap = 0;
  FramesDev D{};
  CamDev cam{};
  float width = 640.f, height = 480.f;
  vo::DevBuf b_x, b_y, b_angle, b_ur, b_depth, b_oct, b_desc, b_n, b_cs, b_ci, b_srec, b_sdesc;
  // matcher scratch (grow-only)
  vo::DevBuf b_pool, b_rank, b_ovf, b_ovfr, b_qrec, b_used, b_best, b_asg, b_push, b_nm, b_err, b_sf;
  size_t pool_stride = 0, replay_lds_attr = 0;
  float sf_host[16] = {0};  // scale factors last uploaded for vo_track_gather_dev
};

namespace {

int frames_alloc(vo_frames *h) {
  const size_t N = (size_t)h->max_frames * h->cap;
  VO_CHECK(h->b_x.reserve(N * 4));
  VO_CHECK(h->b_y.reserve(N * 4));
  VO_CHECK(h->b_angle.reserve(N * 4));
  VO_CHECK(h->b_ur.reserve(N * 4));
  VO_CHECK(h->b_depth.reserve(N * 4));
  VO_CHECK(h->b_oct.reserve(N * 4));
  VO_CHECK(h->b_desc.reserve(N * 32));
  VO_CHECK(h->b_n.reserve((size_t)h->max_frames * 4 + 64));
  VO_CHECK(h->b_cs.reserve((size_t)h->max_frames * (kCells + 1) * 4));
  VO_CHECK(h->b_ci.reserve(N * 2 + 64));
  VO_CHECK(h->b_srec.reserve(N * 16 + 64));
  VO_CHECK(h->b_sdesc.reserve(N * 32 + 64));
  FramesDev &D = h->D;
  D.cap = h->cap;
  D.x = h->b_x.as<float>(), D.y = h->b_y.as<float>(), D.angle = h->b_angle.as<float>();
  D.uright = h->b_ur.as<float>(), D.depth = h->b_depth.as<float>(), D.octave = h->b_oct.as<int>();
  D.desc = h->b_desc.as<uint8_t>(), D.n = h->b_n.as<int>(), D.cell_start = h->b_cs.as<int>();
  D.cell_items = h->b_ci.as<unsigned short>();
  D.srec = h->b_srec.as<uint4>(), D.sdesc = h->b_sdesc.as<uint4>();
  return VO_OK;
}

void frames_set_bounds(vo_frames *h, float xmin, float ymin, float xmax, float ymax) {
  h->D.xmin = xmin, h->D.ymin = ymin;
  h->D.gw = (float)kGridCols / (xmax - xmin);  // camera.cpp:45-46
  h->D.gh = (float)kGridRows / (ymax - ymin);
}

struct GuidedCall {
  int mode;
  float radius, bf, ratio, dist_threshold;
  int direction, check_rot, n_levels, max_dist;
  const float *scale_factors;
  int n_scale;
  int retry_below = 0;
  int *retry_nq = nullptr;
};

// Enqueue the matcher kernels for frames [slot0, slot0 + n_frames) on `st`.  All pointers are device memory.
int guided_launch(vo_frames *h, int slot0, int n_frames, const Queries &Q, const GuidedCall &c, const uint8_t *fmask,
                  int *assigned, int *best_idx, int *n_matches, size_t pool_per_frame, hipStream_t st) {
  GuidedParams P{};
  P.mode = c.mode, P.radius = c.radius, P.bf = c.bf, P.ratio = c.ratio, P.dist_threshold = c.dist_threshold;
  P.direction = c.direction, P.check_rot = c.check_rot, P.n_levels = c.n_levels, P.max_dist = c.max_dist;
  for (int i = 0; i < 16; i++) P.sf[i] = i < c.n_scale ? c.scale_factors[i] : (c.n_scale > 0 ? c.scale_factors[c.n_scale - 1] : 1.f);
  const bool claims = c.mode == kModeFrame || c.mode == kModeLocalMap || c.mode == kModeKeyFrame || c.mode == kModeSim3;
  GuidedOut O{};
  O.best_idx = best_idx, O.assigned = assigned, O.fmask = fmask, O.n_matches = n_matches;
  P.retry_below = c.retry_below, O.retry_nq = c.retry_nq;
  // the overflow flag is sticky (like the extractor's): kernels only ever set it, vo_match_guided_status reads and clears
  // it -- two searches launched back to back on one handle (the tracked path) cannot erase each other's report
  if (!h->b_err.p) {
    VO_CHECK(h->b_err.reserve(64));
    VO_HIP_CHECK(hipMemsetAsync(h->b_err.p, 0, 64, st));
  }
  O.err = h->b_err.as<int>();
  if (claims) {
    // every query owns kSlot records; `pool_per_frame` sizes the overflow area dense windows spill into
    const size_t slots = (size_t)n_frames * Q.stride * kSlot;
    VO_CHECK(h->b_pool.reserve(slots * 4 + 64));
    VO_CHECK(h->b_ovf.reserve((size_t)n_frames * pool_per_frame * 4 + 64));
    if (c.mode == kModeSim3) {
      VO_CHECK(h->b_rank.reserve(slots * 2 + 64));
      VO_CHECK(h->b_ovfr.reserve((size_t)n_frames * pool_per_frame * 2 + 64));
    }
    VO_CHECK(h->b_qrec.reserve((size_t)n_frames * Q.stride * 8 + 64));
    {
      // overflow cursors: zero when allocated, and k_guided_replay leaves a frame's cursor at zero again (one fill kernel
      // fewer per search in a chain where every dispatch costs ~4 us)
      void *before = h->b_used.p;
      VO_CHECK(h->b_used.reserve((size_t)h->max_frames * 4 + 64));
      if (h->b_used.p != before) VO_HIP_CHECK(hipMemsetAsync(h->b_used.p, 0, (size_t)h->max_frames * 4 + 64, st));
    }
    VO_CHECK(h->b_push.reserve((size_t)n_frames * Q.stride * 4 + 64));
    O.pool = h->b_pool.as<unsigned>(), O.ovf = h->b_ovf.as<unsigned>();
    O.rank = c.mode == kModeSim3 ? h->b_rank.as<unsigned short>() : nullptr;
    O.ovf_rank = c.mode == kModeSim3 ? h->b_ovfr.as<unsigned short>() : nullptr;
    O.qrec = h->b_qrec.as<uint2>(), O.ovf_used = h->b_used.as<int>(), O.pushes = h->b_push.as<unsigned>();
    O.ovf_stride = (int)pool_per_frame;
  }
  const int nq_max = Q.nq_all;
  if (nq_max > 0) {
    // narrow windows (the tracking searches: 15 px, 2.5-4 px times the level scale: 1-5 features) -> eight lanes per
    // query; wide ones (relocalisation, fuse by pose: tens of features) -> sixteen
    if (c.radius <= kNarrowRadius)
      hipLaunchKernelGGL(k_guided_cand<8>, dim3((nq_max + 31) / 32, n_frames), dim3(256), 0, st, h->D, Q, P, O, slot0);
    else
      hipLaunchKernelGGL(k_guided_cand<16>, dim3((nq_max + 15) / 16, n_frames), dim3(256), 0, st, h->D, Q, P, O, slot0);
  }
  if (claims) {
    // LDS per frame: blocked (1 B per feature slot), asg (2 B), and the batched replay's two int arrays (8 B) where
    // they fit (up to 14 k features per frame; beyond that the four-query serial replay runs)
    const size_t capA = (size_t)((h->cap + 15) & ~15);
    const int batched = capA * 11 <= 160 * 1024 - 256 ? 1 : 0;
    const size_t lds = capA * 3 + (batched ? capA * 8 : 0);
    if (lds > 64 * 1024 && lds > h->replay_lds_attr) {
      VO_HIP_CHECK(hipFuncSetAttribute((const void *)k_guided_replay, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      h->replay_lds_attr = lds;
    }
    hipLaunchKernelGGL(k_guided_replay, dim3(n_frames), dim3(64), lds, st, h->D, Q, P, O, slot0, batched);
  } else {
    hipLaunchKernelGGL(k_guided_count, dim3(n_frames), dim3(256), 0, st, Q, best_idx, n_matches);
  }
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

}  // namespace

extern "C" {

int vo_frames_create(vo_frames **out, int max_frames, int max_features) {
  if (!out || max_frames < 1 || max_features < 1 || max_features > kMaxFeat) {
    vo::set_error("vo_frames_create: 1 <= max_features <= %d", kMaxFeat);
    return VO_ERR_INVALID;
  }
  VO_CHECK(vo::ensure_device());
  vo_frames *h = new vo_frames();
  h->max_frames = max_frames, h->cap = max_features;
  const int rc = frames_alloc(h);
  if (rc != VO_OK) {
    vo_frames_destroy(h);
    return rc;
  }
  frames_set_bounds(h, 0.f, 0.f, 640.f, 480.f);
  const size_t lds = (size_t)((h->cap + 15) & ~15) + (size_t)h->cap * 2;
  if (lds > 48 * 1024)
    VO_HIP_CHECK(hipFuncSetAttribute((const void *)k_guided_replay, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  *out = h;
  return VO_OK;
}

void vo_frames_destroy(vo_frames *h) {
  if (!h) return;
  for (vo::DevBuf *b : {&h->b_x, &h->b_y, &h->b_angle, &h->b_ur, &h->b_depth, &h->b_oct, &h->b_desc, &h->b_n, &h->b_cs,
                        &h->b_ci, &h->b_srec, &h->b_sdesc, &h->b_pool, &h->b_rank, &h->b_ovf, &h->b_ovfr, &h->b_qrec, &h->b_used, &h->b_best, &h->b_asg, &h->b_push,
                        &h->b_nm, &h->b_err, &h->b_sf})
    b->release();
  delete h;
}

int vo_frames_set_camera(vo_frames *h, const float intrinsics[5], const float dist_coef[5], float width, float height) {
  if (!h || !intrinsics || !(width > 0) || !(height > 0)) return VO_ERR_INVALID;
  h->cam.fx = intrinsics[0], h->cam.fy = intrinsics[1], h->cam.cx = intrinsics[2], h->cam.cy = intrinsics[3];
  h->cam.bf = intrinsics[4];
  for (int i = 0; i < 5; i++) h->cam.k[i] = dist_coef ? (double)dist_coef[i] : 0.0;
  h->cam.distorted = dist_coef && dist_coef[0] != 0.0f;  // frame.cpp:41
  h->width = width, h->height = height;
  frames_set_bounds(h, 0.f, 0.f, width, height);  // camera.cpp:40-46
  return VO_OK;
}

int vo_frames_capacity(const vo_frames *h, int *max_frames, int *max_features) {
  if (!h) return VO_ERR_INVALID;
  if (max_frames) *max_frames = h->max_frames;
  if (max_features) *max_features = h->cap;
  return VO_OK;
}

int vo_frames_build_dev(vo_frames *h, int slot0, int n_frames, const vo_keypoint *dev_keypoints,
                        const uint8_t *dev_descriptors, const int32_t *dev_counts, int capacity, const void *dev_depth,
                        int depth_kind, size_t depth_frame_stride_bytes, int depth_pitch_bytes, float inv_depth_scale,
                        void *hip_stream) {
  if (!h || slot0 < 0 || n_frames < 1 || slot0 + n_frames > h->max_frames || !dev_keypoints || !dev_descriptors ||
      !dev_counts || capacity < 1 || depth_kind < 0 || depth_kind > 2 || (depth_kind && !dev_depth)) {
    vo::set_error("vo_frames_build_dev: invalid argument");
    return VO_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)hip_stream;
  const int nmax = std::min(capacity, h->cap);
  if (!h->b_err.p) {  // the handle's sticky flag (read and cleared by vo_match_guided_status)
    VO_CHECK(h->b_err.reserve(64));
    VO_HIP_CHECK(hipMemsetAsync(h->b_err.p, 0, 64, st));
  }
  hipLaunchKernelGGL(k_frame_post, dim3((nmax + 255) / 256, n_frames), dim3(256), 0, st, h->D, h->cam, dev_keypoints,
                     dev_descriptors, dev_counts, capacity, dev_depth, depth_kind, (long long)depth_frame_stride_bytes,
                     depth_pitch_bytes, inv_depth_scale, (int)h->width, (int)h->height, slot0, h->b_err.as<int>());
  hipLaunchKernelGGL(h->cap <= kGridLdsCap ? k_frame_grid<true> : k_frame_grid<false>, dim3(n_frames), dim3(256), 0, st, h->D, slot0);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_frames_upload(vo_frames *h, int slot, const vo_frame_view *view, const float *depth, void *hip_stream) {
  if (!h || slot < 0 || slot >= h->max_frames || !view || view->n < 0) return VO_ERR_INVALID;
  if (view->n > h->cap) {
    vo::set_error("vo_frames_upload: %d features exceed the handle's %d slots", view->n, h->cap);
    return VO_ERR_CAPACITY;
  }
  hipStream_t st = (hipStream_t)hip_stream;
  const int n = view->n;
  const size_t o = (size_t)slot * h->cap;
  frames_set_bounds(h, view->xmin, view->ymin, view->xmax, view->ymax);
  if (n > 0) {
    if (!view->x || !view->y || !view->octave || !view->desc) return VO_ERR_INVALID;
    VO_CHECK(vo::copy_h2d(h->D.x + o, view->x, (size_t)n * 4, st, "vo_frames_upload"));
    VO_CHECK(vo::copy_h2d(h->D.y + o, view->y, (size_t)n * 4, st, "vo_frames_upload"));
    VO_CHECK(vo::copy_h2d(h->D.octave + o, view->octave, (size_t)n * 4, st, "vo_frames_upload"));
    if (view->angle) VO_CHECK(vo::copy_h2d(h->D.angle + o, view->angle, (size_t)n * 4, st, "vo_frames_upload"));
    else VO_HIP_CHECK(hipMemsetAsync(h->D.angle + o, 0, (size_t)n * 4, st));
    if (view->uright) VO_CHECK(vo::copy_h2d(h->D.uright + o, view->uright, (size_t)n * 4, st, "vo_frames_upload"));
    else VO_HIP_CHECK(hipMemsetAsync(h->D.uright + o, 0xbf, (size_t)n * 4, st));  // 0xbfbfbfbf = -1.498...: "no depth"
    if (depth) VO_CHECK(vo::copy_h2d(h->D.depth + o, depth, (size_t)n * 4, st, "vo_frames_upload"));
    VO_CHECK(vo::copy_h2d(h->D.desc + o * 32, view->desc, (size_t)n * 32, st, "vo_frames_upload"));
  }
  VO_CHECK(vo::copy_h2d(h->D.n + slot, &n, 4, st, "vo_frames_upload"));
  hipLaunchKernelGGL(h->cap <= kGridLdsCap ? k_frame_grid<true> : k_frame_grid<false>, dim3(1), dim3(256), 0, st, h->D, slot);
  VO_HIP_CHECK(hipGetLastError());
  // `n` lives on this stack frame: the copy above must have left it before we return
  return vo::stream_sync(st, "vo_frames_upload");
}

int vo_frames_download(vo_frames *h, int slot, int *n, float *x, float *y, int32_t *octave, float *angle, float *uright,
                       float *depth, uint8_t *desc, int32_t *cell_start, uint16_t *cell_items, void *hip_stream) {
  if (!h || slot < 0 || slot >= h->max_frames || !n) return VO_ERR_INVALID;
  hipStream_t st = (hipStream_t)hip_stream;
  int cnt = 0;
  VO_CHECK(vo::copy_d2h(&cnt, h->D.n + slot, 4, st, "vo_frames_download"));
  VO_CHECK(vo::stream_sync(st, "vo_frames_download"));
  *n = cnt;
  const size_t o = (size_t)slot * h->cap;
  if (x) VO_CHECK(vo::copy_d2h(x, h->D.x + o, (size_t)cnt * 4, st, "vo_frames_download"));
  if (y) VO_CHECK(vo::copy_d2h(y, h->D.y + o, (size_t)cnt * 4, st, "vo_frames_download"));
  if (octave) VO_CHECK(vo::copy_d2h(octave, h->D.octave + o, (size_t)cnt * 4, st, "vo_frames_download"));
  if (angle) VO_CHECK(vo::copy_d2h(angle, h->D.angle + o, (size_t)cnt * 4, st, "vo_frames_download"));
  if (uright) VO_CHECK(vo::copy_d2h(uright, h->D.uright + o, (size_t)cnt * 4, st, "vo_frames_download"));
  if (depth) VO_CHECK(vo::copy_d2h(depth, h->D.depth + o, (size_t)cnt * 4, st, "vo_frames_download"));
  if (desc) VO_CHECK(vo::copy_d2h(desc, h->D.desc + o * 32, (size_t)cnt * 32, st, "vo_frames_download"));
  if (cell_start)
    VO_CHECK(vo::copy_d2h(cell_start, h->D.cell_start + (size_t)slot * (kCells + 1), (size_t)(kCells + 1) * 4, st, "vo_frames_download"));
  if (cell_items) VO_CHECK(vo::copy_d2h(cell_items, h->D.cell_items + o, (size_t)cnt * 2, st, "vo_frames_download"));
  return vo::stream_sync(st, "vo_frames_download");
}

int vo_frames_features_in_area(vo_frames *h, int slot, int n_queries, const float *u, const float *v, const float *radius,
                               const int32_t *min_level, const int32_t *max_level, int32_t *out_idx, int max_out,
                               int32_t *out_count) {
  if (!h || slot < 0 || slot >= h->max_frames || n_queries < 0 || max_out < 0 ||
      (n_queries > 0 && (!u || !v || !radius || !out_count || (max_out > 0 && !out_idx))))
    return VO_ERR_INVALID;
  if (n_queries == 0) return VO_OK;
  thread_local vo::ScratchBuf du, dv, dr, dlo, dhi, dout, dcnt;
  hipStream_t st = vo::thread_stream();
  const char *what = "vo_frames_features_in_area";
  VO_CHECK(vo::upload(du, u, (size_t)n_queries * 4, st, what));
  VO_CHECK(vo::upload(dv, v, (size_t)n_queries * 4, st, what));
  VO_CHECK(vo::upload(dr, radius, (size_t)n_queries * 4, st, what));
  if (min_level) VO_CHECK(vo::upload(dlo, min_level, (size_t)n_queries * 4, st, what));
  if (max_level) VO_CHECK(vo::upload(dhi, max_level, (size_t)n_queries * 4, st, what));
  VO_CHECK(dout.reserve(std::max<size_t>((size_t)n_queries * max_out * 4, 64)));
  VO_CHECK(dcnt.reserve((size_t)n_queries * 4));
  hipLaunchKernelGGL(k_features_in_area, dim3((n_queries + 15) / 16), dim3(256), 0, st, h->D, slot, n_queries, du.as<float>(),
                     dv.as<float>(), dr.as<float>(), min_level ? dlo.as<int>() : nullptr, max_level ? dhi.as<int>() : nullptr,
                     dout.as<int>(), max_out, dcnt.as<int>());
  VO_HIP_CHECK(hipGetLastError());
  if (max_out > 0) VO_CHECK(vo::copy_d2h(out_idx, dout.p, (size_t)n_queries * max_out * 4, st, what));
  VO_CHECK(vo::copy_d2h(out_count, dcnt.p, (size_t)n_queries * 4, st, what));
  return vo::stream_sync(st, what);
}

int vo_match_guided_dev(vo_frames *h, int slot0, int n_frames, const vo_guided_queries *q, const vo_guided_params *p,
                        const uint8_t *dev_feature_mask, int32_t *dev_assigned, int32_t *dev_best_idx,
                        int32_t *dev_n_matches, size_t pool_per_frame, void *hip_stream) {
  if (!h || !q || !p || slot0 < 0 || n_frames < 1 || slot0 + n_frames > h->max_frames || q->n_queries < 0 ||
      q->stride < q->n_queries || !dev_n_matches || p->mode < 0 || p->mode > 5 || !p->scale_factors || p->n_levels < 1 ||
      p->n_levels > 16) {
    vo::set_error("vo_match_guided_dev: invalid argument");
    return VO_ERR_INVALID;
  }
  const bool claims = p->mode == kModeFrame || p->mode == kModeLocalMap || p->mode == kModeKeyFrame || p->mode == kModeSim3;
  if ((claims && !dev_assigned) || (!claims && !dev_best_idx) || q->n_queries > 65534) {
    vo::set_error("vo_match_guided_dev: missing output array or more than 65534 queries per frame");
    return VO_ERR_INVALID;
  }
  Queries Q{};
  Q.flags = q->flags, Q.u = q->u, Q.v = q->v, Q.aux = q->aux, Q.level = q->level, Q.angle = q->angle;
  Q.viewcos = q->viewcos, Q.desc = q->desc, Q.nq = q->n_per_frame, Q.nq_all = q->n_queries, Q.stride = q->stride;
  GuidedCall c{p->mode, p->radius, p->bf, p->ratio, p->dist_threshold, p->direction, p->check_rot, p->n_levels,
               p->max_dist, p->scale_factors, p->n_levels};
  if (p->retry_n_per_frame) {
    if (p->mode != kModeFrame) {
      vo::set_error("vo_match_guided_dev: retry_n_per_frame is for mode 0 (trackWithMotion's retry)");
      return VO_ERR_INVALID;
    }
    c.retry_below = p->retry_below, c.retry_nq = p->retry_n_per_frame;
  }
  if (pool_per_frame == 0) pool_per_frame = (size_t)std::max(q->n_queries, 1) * 16;
  return guided_launch(h, slot0, n_frames, Q, c, dev_feature_mask, dev_assigned, dev_best_idx, dev_n_matches,
                       pool_per_frame, (hipStream_t)hip_stream);
}

int vo_track_scatter_dev(vo_frames *h, int slot0, int n_frames, const int32_t *dev_assigned, const double *dev_query_points,
                         const uint8_t *dev_query_flags, int stride, double *dev_feature_points, uint8_t *dev_feature_has,
                         uint8_t *dev_feature_observed, void *hip_stream) {
  if (!h || slot0 < 0 || n_frames < 1 || slot0 + n_frames > h->max_frames || !dev_assigned || !dev_query_points ||
      !dev_query_flags || stride < 1 || !dev_feature_points || !dev_feature_has)
    return VO_ERR_INVALID;
  vo::track_scatter_launch(n_frames, h->cap, stride, h->D.n, slot0, dev_assigned, dev_query_points, dev_query_flags,
                           dev_feature_points, dev_feature_has, dev_feature_observed, (hipStream_t)hip_stream);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_track_gather_dev(vo_frames *h, int slot0, int n_frames, const double *dev_feature_points,
                        const uint8_t *dev_feature_has, const float *scale_factors, int n_levels, double *dev_points,
                        double *dev_obs, double *dev_inv_sigma, int32_t *dev_ranges, int32_t *dev_index, void *hip_stream) {
  if (!h || slot0 < 0 || n_frames < 1 || slot0 + n_frames > h->max_frames || !dev_feature_points || !dev_feature_has ||
      !scale_factors || n_levels < 1 || n_levels > 16 || !dev_points || !dev_obs || !dev_inv_sigma || !dev_ranges)
    return VO_ERR_INVALID;
  hipStream_t st = (hipStream_t)hip_stream;
  if (!h->b_sf.p) VO_CHECK(h->b_sf.reserve(64));
  if (memcmp(h->sf_host, scale_factors, (size_t)n_levels * 4) != 0) {  // uploaded when it changes (once per extractor)
    memcpy(h->sf_host, scale_factors, (size_t)n_levels * 4);
    VO_HIP_CHECK(hipMemcpyAsync(h->b_sf.p, h->sf_host, 64, hipMemcpyHostToDevice, st));
  }
  vo::track_gather_launch(n_frames, h->cap, h->D.n, slot0, h->D.x, h->D.y, h->D.uright, h->D.octave, dev_feature_points,
                          dev_feature_has, h->b_sf.as<float>(), dev_points, dev_obs, dev_inv_sigma, dev_ranges, dev_index, st);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_track_scatter_gather_dev(vo_frames *h, int slot0, int n_frames, const int32_t *dev_assigned, const double *dev_query_points,
                                const uint8_t *dev_query_flags, int stride, double *dev_feature_points, uint8_t *dev_feature_has,
                                uint8_t *dev_feature_observed, const float *scale_factors, int n_levels, double *dev_points,
                                double *dev_obs, double *dev_inv_sigma, int32_t *dev_ranges, int32_t *dev_index, void *hip_stream) {
  if (!h || slot0 < 0 || n_frames < 1 || slot0 + n_frames > h->max_frames || !dev_assigned || !dev_query_points ||
      !dev_query_flags || stride < 1 || !dev_feature_points || !dev_feature_has || !scale_factors || n_levels < 1 ||
      n_levels > 16 || !dev_points || !dev_obs || !dev_inv_sigma || !dev_ranges)
    return VO_ERR_INVALID;
  hipStream_t st = (hipStream_t)hip_stream;
  if (!h->b_sf.p) VO_CHECK(h->b_sf.reserve(64));
  if (memcmp(h->sf_host, scale_factors, (size_t)n_levels * 4) != 0) {  // uploaded when it changes (once per extractor)
    memcpy(h->sf_host, scale_factors, (size_t)n_levels * 4);
    VO_HIP_CHECK(hipMemcpyAsync(h->b_sf.p, h->sf_host, 64, hipMemcpyHostToDevice, st));
  }
  vo::track_scatter_gather_launch(n_frames, h->cap, stride, h->D.n, slot0, dev_assigned, dev_query_points, dev_query_flags,
                                  dev_feature_points, dev_feature_has, dev_feature_observed, h->D.x, h->D.y, h->D.uright,
                                  h->D.octave, h->b_sf.as<float>(), dev_points, dev_obs, dev_inv_sigma, dev_ranges, dev_index, st);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_match_guided_status(vo_frames *h, void *hip_stream) {
  if (!h) return VO_ERR_INVALID;
  if (!h->b_err.p) return VO_OK;
  int e = 0;
  hipStream_t st = (hipStream_t)hip_stream;
  VO_CHECK(vo::copy_d2h(&e, h->b_err.p, 4, st, "vo_match_guided_status"));
  VO_CHECK(vo::stream_sync(st, "vo_match_guided_status"));
  if (e) {
    VO_HIP_CHECK(hipMemsetAsync(h->b_err.p, 0, 4, st));  // reported once
    vo::set_error(e & 2 ? "frame store: a frame had more key-points than the store's slots per frame (they were dropped)"
                        : "guided matcher: candidate pool overflow in a search since the last status call (raise pool_per_frame)");
    return VO_ERR_CAPACITY;
  }
  return VO_OK;
}

}  // extern "C"

// ---- host-array entry points (the reference's call shape: one frame, everything in host memory) ----------
namespace {

struct HostCtx {  // per host thread: one-slot frame store + staging, grow-only
  vo_frames *fr = nullptr;
  vo::ScratchBuf dq, dout;
  vo::PinnedBuf pin;
};

HostCtx &host_ctx() {
  thread_local HostCtx c;
  return c;
}

struct HostQueries {
  int nq;
  const uint8_t *flags;
  const float *u, *v, *aux;
  const int32_t *level;
  const float *angle, *viewcos;
  const uint8_t *desc;
};

// One frame through the device matcher: upload the frame view and the queries (one staging block), run, fetch
// the result.  mask_in: blocked / has-map-point / occupied [cur.n] or NULL.  assigned in/out or best_idx out.
int guided_host(const vo_frame_view *cur, const HostQueries &hq, const GuidedCall &c, const uint8_t *mask_in,
                int32_t *assigned, int32_t *best_idx, int *n_matches) {
  VO_CHECK(vo::ensure_device());
  HostCtx &C = host_ctx();
  const int nf = cur->n, nq = hq.nq;
  if (nf > kMaxFeat || nq > 65534) {
    vo::set_error("guided matcher: %d features / %d queries exceed %d / 65534", nf, nq, kMaxFeat);
    return VO_ERR_CAPACITY;
  }
  if (!C.fr || C.fr->cap < nf) {
    if (C.fr) vo_frames_destroy(C.fr);
    C.fr = nullptr;
    int cap = 2048;
    while (cap < nf) cap *= 2;
    VO_CHECK(vo_frames_create(&C.fr, 1, cap));
  }
  vo_frames *h = C.fr;
  hipStream_t st = vo::thread_stream();
  // staging layout (4-byte aligned blocks)
  auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t o_x = 0, o_y = o_x + up((size_t)nf * 4), o_oct = o_y + up((size_t)nf * 4), o_ang = o_oct + up((size_t)nf * 4),
               o_ur = o_ang + up((size_t)nf * 4), o_fd = o_ur + up((size_t)nf * 4), o_mask = o_fd + up((size_t)nf * 32),
               o_asg = o_mask + up((size_t)nf), o_qf = o_asg + up((size_t)nf * 4), o_qu = o_qf + up((size_t)nq),
               o_qv = o_qu + up((size_t)nq * 4), o_qa = o_qv + up((size_t)nq * 4), o_ql = o_qa + up((size_t)nq * 4),
               o_qang = o_ql + up((size_t)nq * 4), o_qvc = o_qang + up((size_t)nq * 4), o_qd = o_qvc + up((size_t)nq * 4),
               o_n = o_qd + up((size_t)nq * 32), in_bytes = o_n + 16;
  const size_t r_asg = 0, r_best = r_asg + up((size_t)nf * 4), r_nm = r_best + up((size_t)nq * 4), r_err = r_nm + 16,
               out_bytes = r_err + 16;
  VO_CHECK(C.pin.reserve(std::max(in_bytes, out_bytes)));
  VO_CHECK(C.dq.reserve(in_bytes));
  VO_CHECK(C.dout.reserve(out_bytes));
  uint8_t *sg = C.pin.data();
  memset(sg, 0, in_bytes);
  auto put = [&](size_t off, const void *src, size_t bytes) {
    if (src && bytes) memcpy(sg + off, src, bytes);
  };
  put(o_x, cur->x, (size_t)nf * 4), put(o_y, cur->y, (size_t)nf * 4), put(o_oct, cur->octave, (size_t)nf * 4);
  put(o_ang, cur->angle, (size_t)nf * 4), put(o_fd, cur->desc, (size_t)nf * 32), put(o_mask, mask_in, (size_t)nf);
  if (cur->uright) put(o_ur, cur->uright, (size_t)nf * 4);
  else for (int i = 0; i < nf; i++) reinterpret_cast<float *>(sg + o_ur)[i] = -1.f;
  if (assigned) put(o_asg, assigned, (size_t)nf * 4);
  put(o_qf, hq.flags, (size_t)nq), put(o_qu, hq.u, (size_t)nq * 4), put(o_qv, hq.v, (size_t)nq * 4);
  put(o_qa, hq.aux, (size_t)nq * 4), put(o_ql, hq.level, (size_t)nq * 4), put(o_qang, hq.angle, (size_t)nq * 4);
  put(o_qvc, hq.viewcos, (size_t)nq * 4), put(o_qd, hq.desc, (size_t)nq * 32);
  memcpy(sg + o_n, &nf, 4);
  VO_CHECK(vo::copy_h2d(C.dq.p, sg, in_bytes, st, "guided matcher"));
  uint8_t *d = C.dq.as<uint8_t>(), *r = C.dout.as<uint8_t>();
  // frame slot 0 <- the staged view (device-to-device), then its grid
  FramesDev &D = h->D;
  frames_set_bounds(h, cur->xmin, cur->ymin, cur->xmax, cur->ymax);
  auto d2d = [&](void *dst, size_t off, size_t bytes) -> int {
    if (bytes) VO_HIP_CHECK(hipMemcpyAsync(dst, d + off, bytes, hipMemcpyDeviceToDevice, st));
    return VO_OK;
  };
  VO_CHECK(d2d(D.x, o_x, (size_t)nf * 4));
  VO_CHECK(d2d(D.y, o_y, (size_t)nf * 4));
  VO_CHECK(d2d(D.octave, o_oct, (size_t)nf * 4));
  VO_CHECK(d2d(D.angle, o_ang, (size_t)nf * 4));
  VO_CHECK(d2d(D.uright, o_ur, (size_t)nf * 4));
  VO_CHECK(d2d(D.desc, o_fd, (size_t)nf * 32));
  VO_CHECK(d2d(D.n, o_n, 4));
  hipLaunchKernelGGL(D.cap <= kGridLdsCap ? k_frame_grid<true> : k_frame_grid<false>, dim3(1), dim3(256), 0, st, D, 0);
  Queries Q{};
  Q.flags = d + o_qf, Q.u = reinterpret_cast<const float *>(d + o_qu), Q.v = reinterpret_cast<const float *>(d + o_qv);
  Q.aux = reinterpret_cast<const float *>(d + o_qa), Q.level = reinterpret_cast<const int *>(d + o_ql);
  Q.angle = reinterpret_cast<const float *>(d + o_qang), Q.viewcos = reinterpret_cast<const float *>(d + o_qvc);
  Q.desc = d + o_qd, Q.nq = nullptr, Q.nq_all = nq, Q.stride = std::max(nq, 1);
  int *d_asg = reinterpret_cast<int *>(r + r_asg), *d_best = reinterpret_cast<int *>(r + r_best);
  int *d_nm = reinterpret_cast<int *>(r + r_nm);
  const bool claims = c.mode == kModeFrame || c.mode == kModeLocalMap || c.mode == kModeKeyFrame || c.mode == kModeSim3;
  if (claims) VO_CHECK(d2d(d_asg, o_asg, (size_t)nf * 4));
  // the cap of the frame store may exceed nf: assigned / mask arrays are indexed with the store's stride for
  // frame 0 only, so the staged [nf] arrays serve as they are
  size_t pool = std::max<size_t>((size_t)nq * 16, 4096);
  for (int attempt = 0;; attempt++) {
    VO_CHECK(guided_launch(h, 0, 1, Q, c, mask_in ? d + o_mask : nullptr, d_asg, d_best, d_nm, pool, st));
    VO_HIP_CHECK(hipMemcpyAsync(r + r_err, h->b_err.p, 4, hipMemcpyDeviceToDevice, st));
    VO_CHECK(vo::copy_d2h(sg, r, out_bytes, st, "guided matcher"));
    VO_CHECK(vo::stream_sync(st, "guided matcher"));
    int e;
    memcpy(&e, sg + r_err, 4);
    if (!(e & 1)) break;  // (bit 0: pool overflow)
    VO_HIP_CHECK(hipMemsetAsync(h->b_err.p, 0, 4, st));  // the flag is sticky: this retry loop consumes it
    if (attempt >= 6) {
      vo::set_error("guided matcher: candidate pool overflow");
      return VO_ERR_CAPACITY;
    }
    pool *= 4;  // rare (dense windows): redo with a larger pool
    if (claims) VO_CHECK(d2d(d_asg, o_asg, (size_t)nf * 4));
  }
  if (claims && assigned) memcpy(assigned, sg + r_asg, (size_t)nf * 4);
  if (!claims && best_idx) memcpy(best_idx, sg + r_best, (size_t)nq * 4);
  memcpy(n_matches, sg + r_nm, 4);
  return VO_OK;
}

}  // namespace

extern "C" {

int vo_match_frame_projection(const vo_frame_view *cur, int nq, const uint8_t *q_flags, const float *q_u,
                              const float *q_v, const float *q_invz, const int32_t *q_octave, const float *q_angle,
                              const uint8_t *q_desc, float radius, float bf, int direction, int check_rot, int n_levels,
                              const float *scale_factors, const uint8_t *blocked_in, int32_t *assigned, int *n_matches) {
  if (!cur || nq < 0 || !assigned || !n_matches || !scale_factors || n_levels < 1 || n_levels > 16) return VO_ERR_INVALID;
  *n_matches = 0;
  if (nq == 0 || cur->n == 0) return VO_OK;
  if (!q_flags || !q_u || !q_v || !q_invz || !q_octave || !q_desc || (check_rot && !q_angle)) return VO_ERR_INVALID;
  const HostQueries hq{nq, q_flags, q_u, q_v, q_invz, q_octave, q_angle, nullptr, q_desc};
  const GuidedCall c{kModeFrame, radius, bf, 0.f, 0.f, direction, check_rot, n_levels, 0, scale_factors, n_levels};
  return guided_host(cur, hq, c, blocked_in, assigned, nullptr, n_matches);
}

int vo_match_local_map(const vo_frame_view *cur, int nq, const uint8_t *q_flags, const float *q_u, const float *q_v,
                       const float *q_ur, const int32_t *q_level, const float *q_viewcos, const uint8_t *q_desc,
                       float th_radius, float ratio, const float *scale_factors, const uint8_t *blocked_in,
                       int32_t *assigned, int *n_matches) {
  if (!cur || nq < 0 || !assigned || !n_matches || !scale_factors) return VO_ERR_INVALID;
  *n_matches = 0;
  if (nq == 0 || cur->n == 0) return VO_OK;
  if (!q_flags || !q_u || !q_v || !q_ur || !q_level || !q_viewcos || !q_desc) return VO_ERR_INVALID;
  const HostQueries hq{nq, q_flags, q_u, q_v, q_ur, q_level, nullptr, q_viewcos, q_desc};
  const GuidedCall c{kModeLocalMap, th_radius, 0.f, ratio, 0.f, 0, 0, 16, 0, scale_factors, 16};
  return guided_host(cur, hq, c, blocked_in, assigned, nullptr, n_matches);
}

int vo_match_frame_keyframe(const vo_frame_view *cur, int nq, const uint8_t *q_flags, const float *q_u,
                            const float *q_v, const int32_t *q_level, const float *q_angle, const uint8_t *q_desc,
                            float radius, float dist_threshold, int check_rot, const float *scale_factors,
                            const uint8_t *has_mp_in, int32_t *assigned, int *n_matches) {
  if (!cur || nq < 0 || !assigned || !n_matches || !scale_factors) return VO_ERR_INVALID;
  *n_matches = 0;
  if (nq == 0 || cur->n == 0) return VO_OK;
  if (!q_flags || !q_u || !q_v || !q_level || !q_desc || (check_rot && !q_angle)) return VO_ERR_INVALID;
  const HostQueries hq{nq, q_flags, q_u, q_v, nullptr, q_level, q_angle, nullptr, q_desc};
  const GuidedCall c{kModeKeyFrame, radius, 0.f, 0.f, dist_threshold, 0, check_rot, 16, 0, scale_factors, 16};
  return guided_host(cur, hq, c, has_mp_in, assigned, nullptr, n_matches);
}

int vo_match_fuse(const vo_frame_view *kf, int nq, const uint8_t *q_flags, const float *q_u, const float *q_v,
                  const float *q_ur, const int32_t *q_level, const uint8_t *q_desc, float threshold,
                  const float *scale_factors, int32_t *best_idx, int *n_matches) {
  if (!kf || nq < 0 || !best_idx || !n_matches || !scale_factors) return VO_ERR_INVALID;
  for (int i = 0; i < nq; i++) best_idx[i] = -1;
  *n_matches = 0;
  if (nq == 0 || kf->n == 0) return VO_OK;
  if (!q_flags || !q_u || !q_v || !q_ur || !q_level || !q_desc) return VO_ERR_INVALID;
  const HostQueries hq{nq, q_flags, q_u, q_v, q_ur, q_level, nullptr, nullptr, q_desc};
  const GuidedCall c{kModeFuse, threshold, 0.f, 0.f, 0.f, 0, 0, 16, TH_LOW, scale_factors, 16};
  return guided_host(kf, hq, c, nullptr, nullptr, best_idx, n_matches);
}

int vo_match_area_best(const vo_frame_view *kf, int nq, const uint8_t *q_flags, const float *q_u, const float *q_v,
                       const int32_t *q_level, const uint8_t *q_desc, float th, const float *scale_factors,
                       int max_dist, int32_t *best_idx, int *n_matches) {
  if (!kf || nq < 0 || !best_idx || !n_matches || !scale_factors) return VO_ERR_INVALID;
  for (int i = 0; i < nq; i++) best_idx[i] = -1;
  *n_matches = 0;
  if (nq == 0 || kf->n == 0) return VO_OK;
  if (!q_flags || !q_u || !q_v || !q_level || !q_desc) return VO_ERR_INVALID;
  const HostQueries hq{nq, q_flags, q_u, q_v, nullptr, q_level, nullptr, nullptr, q_desc};
  const GuidedCall c{kModeArea, th, 0.f, 0.f, 0.f, 0, 0, 16, max_dist, scale_factors, 16};
  return guided_host(kf, hq, c, nullptr, nullptr, best_idx, n_matches);
}

int vo_match_sim3_projection(const vo_frame_view *kf, int nq, const uint8_t *q_flags, const float *q_u,
                             const float *q_v, const int32_t *q_level, const uint8_t *q_desc, int th,
                             const float *scale_factors, const uint8_t *occupied, int32_t *assigned, int *n_matches) {
  if (!kf || nq < 0 || !assigned || !n_matches || !scale_factors) return VO_ERR_INVALID;
  for (int k = 0; k < kf->n; k++) assigned[k] = -1;
  *n_matches = 0;
  if (nq == 0 || kf->n == 0) return VO_OK;
  if (!q_flags || !q_u || !q_v || !q_level || !q_desc) return VO_ERR_INVALID;
  const HostQueries hq{nq, q_flags, q_u, q_v, nullptr, q_level, nullptr, nullptr, q_desc};
  const GuidedCall c{kModeSim3, (float)th, 0.f, 0.f, 0.f, 0, 0, 16, TH_LOW, scale_factors, 16};
  return guided_host(kf, hq, c, occupied, assigned, nullptr, n_matches);
}

int vo_match_sim3_mutual(const vo_frame_view *kf1, const vo_frame_view *kf2, const uint8_t *q1_flags, const float *q1_u,
                         const float *q1_v, const int32_t *q1_level, const uint8_t *q1_desc, const uint8_t *q2_flags,
                         const float *q2_u, const float *q2_v, const int32_t *q2_level, const uint8_t *q2_desc, float th,
                         const float *scale_factors1, const float *scale_factors2, int32_t *match12, int *n_matches) {
  if (!kf1 || !kf2 || !match12 || !n_matches) return VO_ERR_INVALID;
  std::vector<int32_t> m1(std::max(1, kf1->n)), m2(std::max(1, kf2->n));
  int n1 = 0, n2 = 0;
  VO_CHECK(vo_match_area_best(kf2, kf1->n, q1_flags, q1_u, q1_v, q1_level, q1_desc, th, scale_factors2, TH_HIGH,
                              m1.data(), &n1));
  VO_CHECK(vo_match_area_best(kf1, kf2->n, q2_flags, q2_u, q2_v, q2_level, q2_desc, th, scale_factors1, TH_HIGH,
                              m2.data(), &n2));
  int found = 0;
  for (int i = 0; i < kf1->n; i++) {
    match12[i] = -1;
    if (m1[i] >= 0 && m2[m1[i]] == i) match12[i] = m1[i], found++;  // :853-864
  }
  *n_matches = found;
  return VO_OK;
}

}  // extern "C"

vo::FrameStoreView vo::frame_store_view(const vo_frames *h) {
  return vo::FrameStoreView{h->cap, h->D.desc, h->D.angle, h->D.n};
}

const int *vo::guided_error_flag(const vo_frames *h) { return h ? h->b_err.as<int>() : nullptr; }
