// tracker.hip -- the tracked-frame pipeline behind the C-ABI (vo_tracker_*): what VisualOdometry::trackWithMotion +
// trackLocalMap do per frame (reference src/visualOdometry.cpp:228-251, 286-300, 745-775, 864-886), for a batch of
// independent camera streams that stay in HBM from the image to the pose:
//
//   ORB extraction (frame.cpp:22)                    vo_orb_extract_batch_dev          extraction stream
//   Frame::Frame post-processing, grid (:27-32)      vo_frames_build_dev               tracking stream from here on
//   searchByProjection(frame, last frame, 15)        k_track_project_prep + vo_match_guided_dev mode 0 (+ the 2 x radius retry)
//   solvePoseOnlySE3                                 k_track_scatter_gather (the matches into the frame's slots, the
//                                                    observation list) + k_pose_only
//   cullingOutliersBeforeLocalMap (:864-886)         k_track_cull: outliers of the solve lose their map point
//   searchLocalMapPoints: Frame::isInFrame (frame.cpp:145-190, with the REFINED pose) + MapPoint::predictScale
//                                                    k_track_in_frame
//   searchByProjection(frame, local points, 3)       vo_match_guided_dev mode 1
//   solvePoseOnlySE3, inlier count (:289-300)        k_track_scatter_gather + k_pose_only + k_track_count (which also
//                                                    writes the frame's record of the result block)
//
// 27 kernel launches per batch (round 4: 33), no host synchronisation in between.  The extraction may run on a stream shared by several
// trackers (vo_tracker_config.extract_stream): batch i + 1's extraction then overlaps batch i's searches and pose
// solves (two events order them).  Round 2 kept this sequence in Python (vo_slam_test_amd/tracking.py) without the
// culling step and with the local-map projections fixed before the first solve; host code is now C++
// throughout, as the reference's callers are.
#include "vo_common.h"

#include <cmath>
#include <new>
#include <vector>

#include "ba_math.h"

namespace {

using namespace vo;
using namespace vo::ba;

// Start of a tracked batch: the feature -> map-point state of every frame is cleared and the solver's pose starts at the
// motion-model pose -- one launch instead of four memsets and a copy (a launch or a copy costs ~5 us of a 0.7 ms frame
// when one camera stream is tracked at a time).
__global__ __launch_bounds__(256) void k_track_prep(int cap, int last_stride, int *assigned, uint8_t *fhas, uint8_t *fobs,
                                                    uint8_t *last_matched, const double *pose0, double *pose) {
  const int f = blockIdx.x, tid = threadIdx.x;
  const long long o = (long long)f * cap;
  for (int i = tid; i < cap; i += 256) assigned[o + i] = -1, fhas[o + i] = 0, fobs[o + i] = 0;
  for (int i = tid; i < last_stride; i += 256) last_matched[(long long)f * last_stride + i] = 0;
  if (tid < 6) pose[6 * f + tid] = pose0[6 * f + tid];
}

// k_track_prep and the projection prologue of searchByProjection(Frame*, Frame*) (k_track_project, csrc/track.hip: the
// same arithmetic, matcher.cpp:41-64) in one launch: the two touch disjoint arrays.  Block (x, f) projects queries
// [256 x, 256 x + 256) of frame f and clears the slice of the per-feature state that the blocks of a frame share out.
__global__ __launch_bounds__(256) void k_track_project_prep(int nq, int stride, const double *Tcw, const double *points,
                                                            const uint8_t *pflags, float fx, float fy, float cx, float cy, int xmin,
                                                            int xmax, int ymin, int ymax, uint8_t *qflags, float *qu, float *qv,
                                                            float *qinvz, int cap, int *assigned, uint8_t *fhas, uint8_t *fobs,
                                                            uint8_t *last_matched, const double *pose0, double *pose) {
  const int f = blockIdx.y, tid = threadIdx.x, q = blockIdx.x * 256 + tid;
  {
    const long long o = (long long)f * cap;
    const int step = (int)gridDim.x * 256;
    for (int i = q; i < cap; i += step) assigned[o + i] = -1, fhas[o + i] = 0, fobs[o + i] = 0;
    for (int i = q; i < stride; i += step) last_matched[(long long)f * stride + i] = 0;
    if (blockIdx.x == 0 && tid < 6) pose[6 * f + tid] = pose0[6 * f + tid];
  }
  if (q >= nq) return;
  const long long o = (long long)f * stride + q;
  const double *T = Tcw + 12 * (long long)f, *p = points + 3 * o;
  const unsigned pf = pflags[o];
  uint8_t out = 0;
  float u = 0.f, v = 0.f, invz = 0.f;
  if (pf & 1u) {
    const double x = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[9];
    const double y = T[3] * p[0] + T[4] * p[1] + T[5] * p[2] + T[10];
    const double zc = T[6] * p[0] + T[7] * p[1] + T[8] * p[2] + T[11];
    const float z = (float)zc;
    if (!(z < 0.0f)) {  // :52-53
      invz = 1.0f / z;
      u = (float)((double)fx * x / zc + (double)cx);  // Camera::camera2pixel, camera.cpp:72-75 (float members widened)
      v = (float)((double)fy * y / zc + (double)cy);
      if (!(u < xmin || u > xmax) && !(v < ymin || v > ymax)) out = (uint8_t)(1u | (pf & 2u));  // :61-64
    }
  }
  qflags[o] = out, qu[o] = u, qv[o] = v, qinvz[o] = invz;
}

// End of a batch: everything vo_tracker_results hands back -- pose, the four counts, the status word per frame, and the
// two stages' sticky overflow flags -- in one block, so that the host needs ONE download (it was eight small ones and
// three synchronisations: ~60 us of a 0.7 ms frame).  Record: [pose 6 doubles | n_tracked, n_inliers, n_matches_last,
// n_matches_local, status, 0] = 72 bytes; the flags follow the records.
struct PackArgs {
  int B;
  const double *pose;
  const int *ninl, *nm_first, *nm, *orb_err, *guided_err;
  uint8_t *out;
};
__device__ __forceinline__ void pack_record(const PackArgs &K, int f, int ntracked, int status) {
  double *pd = reinterpret_cast<double *>(K.out + (size_t)f * 72);
  for (int k = 0; k < 6; k++) pd[k] = K.pose[6 * f + k];
  int *pi = reinterpret_cast<int *>(K.out + (size_t)f * 72 + 48);
  pi[0] = ntracked, pi[1] = K.ninl[f], pi[2] = K.nm_first[f], pi[3] = K.nm[f], pi[4] = status, pi[5] = 0;
  if (f == 0) {
    int *fl = reinterpret_cast<int *>(K.out + (size_t)K.B * 72);
    fl[0] = K.orb_err ? *K.orb_err : 0;
    fl[1] = K.guided_err ? *K.guided_err : 0;
  }
}
// (on its own: the first frame of a sequence, whose counts are memsets)
__global__ __launch_bounds__(256) void k_track_pack(PackArgs K, const int *ntracked, const int *status) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f < K.B) pack_record(K, f, ntracked[f], status[f]);
}

// cullingOutliersBeforeLocalMap (visualOdometry.cpp:864-886) on the pose solver's observation list: an outlier's
// feature loses its map point (`mappoints_[i] = nullptr`, and with it the "holds an observed point" mark the local-map
// search tests at matcher.cpp:314); inliers whose point has observations are counted (the function's return value).
// Also marks the last-frame points that were matched at all: matched points -- kept or culled -- carry
// visualIdxOfFrame_ == frame id (:752, :881) and are skipped by searchLocalMapPoints (:765).
// It also keeps the first search's assignments, pose and inlier count for the caller (VO_TRACKER_*_FIRST) and hands
// `assigned` to the second search cleared -- three copies and a memset less per batch.
struct CullArgs {
  int cap;
  const int *ranges, *index;
  const uint8_t *outlier;
  int *assigned;
  int last_stride;
  uint8_t *fhas, *fobs, *last_matched;
  int *n_observed_inliers, *assigned_first;
  const double *pose;
  double *pose_first;
  const int *ninl;
  int *ninl_first;
};
__device__ __forceinline__ void cull_frame(const CullArgs &C, int f, int *s_cnt) {
  const int tid = threadIdx.x;
  const long long o = (long long)f * C.cap;
  const int start = C.ranges[2 * f], count = C.ranges[2 * f + 1];
  for (int i = tid; i < C.cap; i += 256) {
    const int a = C.assigned[o + i];
    if (a >= 0) C.last_matched[(long long)f * C.last_stride + a] = 1;
    C.assigned_first[o + i] = a;
    C.assigned[o + i] = -1;
  }
  if (tid < 6) C.pose_first[6 * f + tid] = C.pose[6 * f + tid];
  if (tid == 6) C.ninl_first[f] = C.ninl[f];
  int local = 0;
  for (int d = tid; d < count; d += 256) {
    const int i = C.index[start + d];
    if (C.outlier[start + d]) {
      C.fhas[o + i] = 0;
      C.fobs[o + i] = 0;
    } else if (C.fobs[o + i]) {
      local++;
    }
  }
  for (int s = 32; s >= 1; s >>= 1) local += __shfl_xor(local, s);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = local;
  __syncthreads();
  if (tid == 0) C.n_observed_inliers[f] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
__global__ __launch_bounds__(256) void k_track_cull(CullArgs C) {
  __shared__ int s_cnt[4];
  cull_frame(C, blockIdx.x, s_cnt);
}
// Frame::isInFrame (frame.cpp:145-190) and MapPoint::predictScale (mappoint.cpp:182-196) for the local map points of
// every frame, with Tcw = exp(pose) of the first solve (frame->setPose, optimizer_ceres.cpp:311).  Arithmetic as the
// reference has it: the transform in double (Sophus SE3 * Vector3d = Eigen's quaternion rotation), z, u, v, the
// distance and the view cosine narrowed to float where the reference narrows them, logf as the correctly rounded
// float of the double logarithm (the reference calls glibc's logf: DESIGN section 3).
// pflags bit 0: the point exists, is not bad and is not already in the frame; bit 1: it has observations.
struct InFrameArgs {
  int nq, stride;
  const double *pose6, *points, *normals;
  const float *min_dist, *max_dist;
  const uint8_t *pflags;
  const int *link;
  const uint8_t *last_matched;
  int last_stride;
  float fx, fy, cx, cy, bf, xmin, xmax, ymin, ymax, log_sf1;
  int n_levels;
  uint8_t *qflags;
  float *qu, *qv, *qur;
  int *qlevel;
  float *qviewcos;
};
// s_T: q (w, x, y, z), t, Ow of the frame's pose (in_frame_pose), set by one thread before a barrier
__device__ __forceinline__ void in_frame_pose(const InFrameArgs &A, int f, double *s_T) {
  const Se3 T = se3_exp(A.pose6 + 6 * (long long)f);
  // Ow_ = Tcw.inverse().translation() (frame.cpp:103): the conjugate rotation of -t
  const double qc[4] = {T.q[0], -T.q[1], -T.q[2], -T.q[3]}, nt[3] = {-T.t[0], -T.t[1], -T.t[2]};
  double ow[3];
  quat_rotate(qc, nt, ow);
  for (int k = 0; k < 4; k++) s_T[k] = T.q[k];
  for (int k = 0; k < 3; k++) s_T[4 + k] = T.t[k], s_T[7 + k] = ow[k];
}
__device__ __forceinline__ void in_frame_query(const InFrameArgs &A, int f, int q, const double *s_T) {
  const long long o = (long long)f * A.stride + q;
  const unsigned pf = A.pflags[o];
  uint8_t out = 0;
  float u = 0.f, v = 0.f, ur = 0.f, vc = 0.f;
  int level = 0;
  const int lk = A.link ? A.link[o] : -1;
  const bool in_frame_already = lk >= 0 && A.last_matched[(long long)f * A.last_stride + lk] != 0;  // :765
  if ((pf & 1u) && !in_frame_already) {
    const double *p = A.points + 3 * o, *nv = A.normals + 3 * o;
    const double qq[4] = {s_T[0], s_T[1], s_T[2], s_T[3]};
    double rp[3];
    quat_rotate(qq, p, rp);
    const double x = rp[0] + s_T[4], y = rp[1] + s_T[5], zc = rp[2] + s_T[6];
    const float z = (float)zc;
    if (!(z < 0.0f)) {  // :153-154
      u = (float)((double)A.fx * x / zc + (double)A.cx);  // Camera::camera2pixel, camera.cpp:72-75 (float members widened)
      v = (float)((double)A.fy * y / zc + (double)A.cy);
      if (!(u < A.xmin || u > A.xmax) && !(v < A.ymin || v > A.ymax)) {  // :159-164
        const double l0 = p[0] - s_T[7], l1 = p[1] - s_T[8], l2 = p[2] - s_T[9];
        const float dist = (float)sqrt(l0 * l0 + l1 * l1 + l2 * l2);  // :167
        const float mind = 0.8f * A.min_dist[o], maxd = 1.2f * A.max_dist[o];  // mappoint.cpp:391-401
        if (!(dist < mind || dist > maxd)) {
          vc = (float)(l0 * nv[0] + l1 * nv[1] + l2 * nv[2]) / dist;  // :176
          if (!(vc < 0.5f)) {
            out = (uint8_t)(1u | (pf & 2u));
            ur = u - A.bf / z;  // :184
            const float ratio = A.max_dist[o] / dist;  // mappoint.cpp:187
            const float lg = (float)log((double)ratio);
            const int s = (int)ceilf(lg / A.log_sf1);
            level = s < 0 ? 0 : (s >= A.n_levels ? A.n_levels - 1 : s);
          }
        }
      }
    }
  }
  if (!out) u = v = ur = vc = 0.f, level = 0;
  A.qflags[o] = out, A.qu[o] = u, A.qv[o] = v, A.qur[o] = ur, A.qlevel[o] = level, A.qviewcos[o] = vc;
}
__global__ __launch_bounds__(256) void k_track_in_frame(InFrameArgs A) {
  __shared__ double s_T[10];
  const int f = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x == 0) in_frame_pose(A, f, s_T);
  __syncthreads();
  if (q < A.nq) in_frame_query(A, f, q, s_T);
}
// trackLocalMap's inlier count (visualOdometry.cpp:289-300): features that hold a map point with observations and are
// no outlier of the second solve; and the per-frame status word.
__global__ __launch_bounds__(256) void k_track_count(int cap, const int *ranges, const int *index, const uint8_t *outlier,
                                                     const uint8_t *fobs, const int *n_first, const int *n_observed_first,
                                                     const int *ninl_first, int *n_tracked, int *status, int min_matches,
                                                     uint8_t *feature_outlier, PackArgs K) {
  __shared__ int s_cnt[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  const long long o = (long long)f * cap;
  const int start = ranges[2 * f], count = ranges[2 * f + 1];
  // frame_curr_->outliers_[i] after the second solve, per feature (features without a map point: 0)
  for (int i = tid; i < cap; i += 256) feature_outlier[o + i] = 0;
  __syncthreads();
  int local = 0;
  for (int d = tid; d < count; d += 256) {
    if (outlier[start + d]) feature_outlier[o + index[start + d]] = 1;
    if (!outlier[start + d] && fobs[o + index[start + d]]) local++;
  }
  for (int s = 32; s >= 1; s >>= 1) local += __shfl_xor(local, s);
  if ((tid & 63) == 0) s_cnt[tid >> 6] = local;
  __syncthreads();
  if (tid == 0) {
    const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    n_tracked[f] = n;
    int st = 0;
    if (n_first[f] < min_matches) st |= VO_TRACK_FEW_MATCHES;      // visualOdometry.cpp:247-248 (20) / :268-269 (15)
    if (n_observed_first[f] < 10) st |= VO_TRACK_FEW_INLIERS;      // :253
    (void)ninl_first;
    status[f] = st;
    pack_record(K, f, n, st);  // the frame's record of the result block (was a launch of its own)
  }
}

// status word and counts after the first stage alone (vo_tracker_track_first): what trackWithMotion / trackRefKeyFrame
// return -- enough matches, and at least 10 observed inliers after the culling (:247-253, :268-275)
__global__ __launch_bounds__(256) void k_track_first_status(int B, const int *n_first, const int *n_observed_first, int *n_tracked,
                                                            int *status, int min_matches, PackArgs K) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= B) return;
  int st = 0;
  if (n_first[f] < min_matches) st |= VO_TRACK_FEW_MATCHES;
  if (n_observed_first[f] < 10) st |= VO_TRACK_FEW_INLIERS;
  n_tracked[f] = n_observed_first[f];
  status[f] = st;
  pack_record(K, f, n_observed_first[f], st);
}

constexpr int kStages = VO_TRACKER_STAGES;

}  // namespace

struct RefKfHost {
  int n = 0;
  std::vector<uint8_t> valid, desc;
  std::vector<float> angle;
  std::vector<int32_t> node_id, start;
  std::vector<uint32_t> feat;
  vo_bow_view view{};
};

struct vo_tracker {
  vo_tracker_config cfg{};
  vo_orb *orb = nullptr;
  vo_frames *frames = nullptr;
  hipStream_t st = nullptr, est = nullptr;
  bool own_st = false, own_est = false;
  hipEvent_t ev_extract = nullptr, ev_build = nullptr;
  // host-buffer calls: the depth image is first read BEHIND the extraction (vo_frames_build_dev), so its upload goes on a copy
  // stream of its own, issued after the extraction has been enqueued: the key-points are being extracted while the host (pageable
  // memory: the runtime stages it synchronously) and the copy engine move the depth -- two thirds of the bytes of a frame
  hipStream_t cst = nullptr;
  hipEvent_t ev_depth = nullptr;
  const void *pend_depth = nullptr;
  size_t pend_depth_bytes = 0;
  bool have_build = false;
  int B = 0, kcap = 0, cap = 0, n_levels = 0, n_last = 0, n_local = 0, nq_last = 0, nq_local = 0;
  float sf[16] = {0};
  // device state
  DevBuf kps, desc, cnt, images, depth;
  DevBuf q0_flags, q0_u, q0_v, q0_aux, q0_level, q0_angle, q0_desc, p0, pf0, last_matched;
  DevBuf q1_flags, q1_u, q1_v, q1_aux, q1_level, q1_viewcos, q1_desc, p1, nrm1, mind1, maxd1, pf1, link1;
  DevBuf Tcw, pose0, pose, pose_first, resblk, retry_nq, foutl;  // retry_nq: [B] query counts of the retry pass  // resblk: k_track_pack's block (72 bytes per frame + 2 flags)
  DevBuf assigned, assigned_first, nm, nm_first, fpoint, fhas, fobs, pts, obs, isg, ranges, index, outlier, ninl, ninl_first,
      nobs_first, ntracked, status;
  PinnedBuf stage;
  bool have_link = false;
  // a search's matches that the next solve_pose writes into the frame's slots (k_track_scatter_gather)
  struct { const int32_t *assigned = nullptr; const double *qpoints = nullptr; const uint8_t *qflags = nullptr; int stride = 0; } pend;
  int first_min_matches = 20;
  // trackRefKeyFrame's reference key-frame per frame of the batch (host copies: the common-node walk is host work)
  const vo_vocab *ref_vocab = nullptr;
  std::vector<RefKfHost> ref_kf;
  // timing
  bool timing = false;
  // 2 events per stage and timed call, in a list that grows with the calls; they are read (after one synchronisation) only
  // by vo_tracker_get_timing, never inside a track call: a timed call enqueues exactly like an untimed one
  std::vector<hipEvent_t> tev;
  int tissued = 0;   // timed calls whose events are recorded and not yet read
  int tslot = -1;    // event set of the call being enqueued (-1: not timed)
  double tms[kStages] = {0};
  int tcalls = 0;
};

namespace {

int alloc_all(vo_tracker *t) {
  const size_t B = t->B, nl = (size_t)t->n_last, nm = (size_t)t->n_local, cap = t->cap, kc = t->kcap;
  VO_CHECK(t->kps.reserve(B * kc * sizeof(vo_keypoint)));
  VO_CHECK(t->desc.reserve(B * kc * 32));
  VO_CHECK(t->cnt.reserve(B * 4 + 64));
  VO_CHECK(t->q0_flags.reserve(B * nl));
  VO_CHECK(t->q0_u.reserve(B * nl * 4));
  VO_CHECK(t->q0_v.reserve(B * nl * 4));
  VO_CHECK(t->q0_aux.reserve(B * nl * 4));
  VO_CHECK(t->q0_level.reserve(B * nl * 4));
  VO_CHECK(t->q0_angle.reserve(B * nl * 4));
  VO_CHECK(t->q0_desc.reserve(B * nl * 32));
  VO_CHECK(t->p0.reserve(B * nl * 24));
  VO_CHECK(t->pf0.reserve(B * nl));
  VO_CHECK(t->last_matched.reserve(B * nl + 64));
  VO_CHECK(t->q1_flags.reserve(B * nm));
  VO_CHECK(t->q1_u.reserve(B * nm * 4));
  VO_CHECK(t->q1_v.reserve(B * nm * 4));
  VO_CHECK(t->q1_aux.reserve(B * nm * 4));
  VO_CHECK(t->q1_level.reserve(B * nm * 4));
  VO_CHECK(t->q1_viewcos.reserve(B * nm * 4));
  VO_CHECK(t->q1_desc.reserve(B * nm * 32));
  VO_CHECK(t->p1.reserve(B * nm * 24));
  VO_CHECK(t->nrm1.reserve(B * nm * 24));
  VO_CHECK(t->mind1.reserve(B * nm * 4));
  VO_CHECK(t->maxd1.reserve(B * nm * 4));
  VO_CHECK(t->pf1.reserve(B * nm));
  VO_CHECK(t->link1.reserve(B * nm * 4));
  VO_CHECK(t->Tcw.reserve(B * 96 + 64));
  VO_CHECK(t->pose0.reserve(B * 48));
  VO_CHECK(t->pose.reserve(B * 48));  // Tcw, then the intrinsics as doubles
  VO_CHECK(t->pose_first.reserve(B * 48));
  VO_CHECK(t->assigned.reserve(B * cap * 4));
  VO_CHECK(t->assigned_first.reserve(B * cap * 4));
  VO_CHECK(t->nm.reserve(B * 4 + 64));
  VO_CHECK(t->nm_first.reserve(B * 4 + 64));
  VO_CHECK(t->fpoint.reserve(B * cap * 24));
  VO_CHECK(t->fhas.reserve(B * cap));
  VO_CHECK(t->fobs.reserve(B * cap));
  VO_CHECK(t->pts.reserve(B * cap * 24));
  VO_CHECK(t->obs.reserve(B * cap * 24));
  VO_CHECK(t->isg.reserve(B * cap * 8));
  VO_CHECK(t->ranges.reserve(B * 8 + 64));
  VO_CHECK(t->index.reserve(B * cap * 4));
  VO_CHECK(t->outlier.reserve(B * cap));
  VO_CHECK(t->ninl.reserve(B * 4 + 64));
  VO_CHECK(t->ninl_first.reserve(B * 4 + 64));
  VO_CHECK(t->nobs_first.reserve(B * 4 + 64));
  VO_CHECK(t->ntracked.reserve(B * 4 + 64));
  VO_CHECK(t->status.reserve(B * 4 + 64));
  VO_CHECK(t->retry_nq.reserve(B * 4 + 64));
  VO_CHECK(t->foutl.reserve(B * cap + 64));
  return VO_OK;
}

// host [B][n][elem] -> device [B][stride][elem], zero padding behind n
int put_rows(vo_tracker *t, DevBuf &dst, const void *src, int n, int stride, size_t elem, const char *what) {
  const size_t B = t->B;
  VO_HIP_CHECK(hipMemsetAsync(dst.p, 0, B * stride * elem, t->st));
  if (n > 0 && src)
    VO_HIP_CHECK(hipMemcpy2DAsync(dst.p, (size_t)stride * elem, src, (size_t)n * elem, (size_t)n * elem, B,
                                  hipMemcpyHostToDevice, t->st));
  (void)what;
  return VO_OK;
}

struct StageTimer {
  vo_tracker *t;
  int stage;
  hipStream_t s;
  StageTimer(vo_tracker *t_, int stage_, hipStream_t s_) : t(t_), stage(stage_), s(s_) {
    if (t->tslot >= 0) (void)hipEventRecord(t->tev[(size_t)t->tslot * 2 * kStages + 2 * stage], s);
  }
  ~StageTimer() {
    if (t->tslot >= 0) (void)hipEventRecord(t->tev[(size_t)t->tslot * 2 * kStages + 2 * stage + 1], s);
  }
};

int collect_timing(vo_tracker *t) {
  if (t->tissued == 0) return VO_OK;
  VO_HIP_CHECK(hipStreamSynchronize(t->st));
  if (t->est != t->st) VO_HIP_CHECK(hipStreamSynchronize(t->est));
  for (int c = 0; c < t->tissued; c++) {
    for (int s = 0; s < kStages; s++) {
      float ms = 0.f;
      const size_t e = (size_t)c * 2 * kStages + 2 * s;
      if (hipEventElapsedTime(&ms, t->tev[e], t->tev[e + 1]) == hipSuccess) t->tms[s] += ms;
    }
    t->tcalls++;
  }
  t->tissued = 0;
  return VO_OK;
}

// the event set of the call about to be enqueued (created on first use)
int begin_timed_call(vo_tracker *t) {
  t->tslot = -1;
  if (!t->timing) return VO_OK;
  const size_t need = (size_t)(t->tissued + 1) * 2 * kStages;
  while (t->tev.size() < need) {
    hipEvent_t e;
    VO_HIP_CHECK(hipEventCreate(&e));
    t->tev.push_back(e);
  }
  t->tslot = t->tissued;
  return VO_OK;
}

int solve_pose(vo_tracker *t) {
  if (t->pend.assigned) {
    VO_CHECK(vo_track_scatter_gather_dev(t->frames, 0, t->B, t->pend.assigned, t->pend.qpoints, t->pend.qflags, t->pend.stride,
                                         t->fpoint.as<double>(), t->fhas.as<uint8_t>(), t->fobs.as<uint8_t>(), t->sf, t->n_levels,
                                         t->pts.as<double>(), t->obs.as<double>(), t->isg.as<double>(), t->ranges.as<int32_t>(),
                                         t->index.as<int32_t>(), t->st));
    t->pend.assigned = nullptr;
  } else {
    VO_CHECK(vo_track_gather_dev(t->frames, 0, t->B, t->fpoint.as<double>(), t->fhas.as<uint8_t>(), t->sf, t->n_levels,
                                 t->pts.as<double>(), t->obs.as<double>(), t->isg.as<double>(), t->ranges.as<int32_t>(),
                                 t->index.as<int32_t>(), t->st));
  }
  return vo_pose_only_solve_ranges_dev(t->B, t->ranges.as<int32_t>(), t->pts.as<double>(), t->obs.as<double>(),
                                       t->isg.as<double>(), t->Tcw.as<double>() + (size_t)t->B * 12, t->pose.as<double>(),
                                       t->outlier.as<uint8_t>(), t->ninl.as<int32_t>(), nullptr, t->st);
}

// What a call runs: the front (extraction + Frame::Frame), one of the two first stages -- trackWithMotion's projection
// search (with its 2 x radius retry) or trackRefKeyFrame's vocabulary-node search --, and the local-map stage.
enum : unsigned { kRunFront = 1u, kRunMotion = 2u, kRunRefKeyFrame = 4u, kRunLocal = 8u };

int stage_front(vo_tracker *t, const uint8_t *dev_images, int img_pitch, size_t img_frame_stride, const void *dev_depth,
                int depth_kind, size_t depth_frame_stride, int depth_pitch, const void *host_depth) {
  const int B = t->B;
  hipStream_t st = t->st, est = t->est;
  const vo_tracker_config &c = t->cfg;
  // ---- extraction (its stream may be shared with other trackers)
  if (est != st && t->have_build) VO_HIP_CHECK(hipStreamWaitEvent(est, t->ev_build, 0));  // last batch's key-points consumed
  {
    StageTimer tm(t, 0, est);
    VO_CHECK(vo_orb_extract_batch_dev(t->orb, dev_images, B, c.width, c.height, img_pitch, img_frame_stride,
                                      t->kps.as<vo_keypoint>(), t->desc.as<uint8_t>(), t->kcap, t->cnt.as<int32_t>()));
  }
  VO_HIP_CHECK(hipEventRecord(t->ev_extract, est));
  if (est != st) VO_HIP_CHECK(hipStreamWaitEvent(st, t->ev_extract, 0));
  if (host_depth) {  // (see vo_tracker::cst)
    if (t->have_build) VO_HIP_CHECK(hipStreamWaitEvent(t->cst, t->ev_build, 0));  // the last call's depth has been consumed
    VO_HIP_CHECK(hipMemcpyAsync(t->depth.p, host_depth, t->pend_depth_bytes, hipMemcpyHostToDevice, t->cst));
    VO_HIP_CHECK(hipEventRecord(t->ev_depth, t->cst));
    VO_HIP_CHECK(hipStreamWaitEvent(st, t->ev_depth, 0));
  }
  // ---- Frame::Frame post-processing
  {
    StageTimer tm(t, 1, st);
    VO_CHECK(vo_frames_build_dev(t->frames, 0, B, t->kps.as<vo_keypoint>(), t->desc.as<uint8_t>(), t->cnt.as<int32_t>(),
                                 t->kcap, dev_depth, depth_kind, depth_frame_stride, depth_pitch, c.inv_depth_scale, st));
  }
  VO_HIP_CHECK(hipEventRecord(t->ev_build, st));
  t->have_build = true;
  return VO_OK;
}

// everything k_track_count / k_track_first_status / k_track_pack need to write the result block
int pack_args(vo_tracker *t, PackArgs &K) {
  VO_CHECK(t->resblk.reserve((size_t)t->B * 72 + 64));
  K = PackArgs{t->B, t->pose.as<double>(), t->ninl.as<int>(), t->nm_first.as<int>(), t->nm.as<int>(), vo::orb_error_flag(t->orb),
               vo::guided_error_flag(t->frames), t->resblk.as<uint8_t>()};
  return VO_OK;
}
int launch_pack(vo_tracker *t) {
  PackArgs K;
  VO_CHECK(pack_args(t, K));
  hipLaunchKernelGGL(k_track_pack, dim3((t->B + 255) / 256), dim3(256), 0, t->st, K, t->ntracked.as<int>(), t->status.as<int>());
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

// trackWithMotion's search (visualOdometry.cpp:238-245): projection, searchByProjection(frame, last frame, radius), the
// retry at 2 x radius for frames with fewer than 20 matches, the map points of the matches into the frame's slots
int stage_motion(vo_tracker *t, const vo_tracker_params &P) {
  const int B = t->B;
  hipStream_t st = t->st;
  const vo_tracker_config &c = t->cfg;
  vo_guided_queries q{};
  vo_guided_params gp{};
  gp.n_levels = t->n_levels, gp.scale_factors = t->sf;
  StageTimer tm(t, 2, st);
  const float cam4[4] = {c.intrinsics[0], c.intrinsics[1], c.intrinsics[2], c.intrinsics[3]};
  // (nq_last > 0 here: run_pipeline takes the first-frame route otherwise)
  hipLaunchKernelGGL(k_track_project_prep, dim3((t->nq_last + 255) / 256, B), dim3(256), 0, st, t->nq_last, t->n_last,
                     t->Tcw.as<double>(), t->p0.as<double>(), t->pf0.as<uint8_t>(), cam4[0], cam4[1], cam4[2], cam4[3], 0, c.width, 0,
                     c.height, t->q0_flags.as<uint8_t>(), t->q0_u.as<float>(), t->q0_v.as<float>(), t->q0_aux.as<float>(), t->cap,
                     t->assigned.as<int>(), t->fhas.as<uint8_t>(), t->fobs.as<uint8_t>(), t->last_matched.as<uint8_t>(),
                     t->pose0.as<double>(), t->pose.as<double>());
  VO_HIP_CHECK(hipGetLastError());
  q.n_queries = t->nq_last, q.stride = t->n_last, q.flags = t->q0_flags.as<uint8_t>(), q.u = t->q0_u.as<float>();
  q.v = t->q0_v.as<float>(), q.aux = t->q0_aux.as<float>(), q.level = t->q0_level.as<int32_t>();
  q.angle = t->q0_angle.as<float>(), q.desc = t->q0_desc.as<uint8_t>();
  gp.mode = 0, gp.radius = P.radius, gp.bf = c.intrinsics[4], gp.direction = P.direction, gp.check_rot = 1;
  // `if (match_num < 20) { fill(mappoints_, nullptr); match_num = searchByProjection(..., 2*radius); }` (:241-245): the
  // first search's replay decides per frame (it knows the count), clears the assignments of the frames that need the
  // second look and writes their query counts; the second call leaves every other frame out (n_per_frame < 0) -- two
  // short dispatches (a candidate grid whose workgroups return at once, a replay) even when no frame needs them: ~8 us
  // per batch inside bench.py's `match_last_frame` stage
  int *rq = t->retry_nq.as<int>();
  if (!P.no_retry) gp.retry_below = 20, gp.retry_n_per_frame = rq;
  VO_CHECK(vo_match_guided_dev(t->frames, 0, B, &q, &gp, nullptr, t->assigned.as<int32_t>(), nullptr,
                               t->nm_first.as<int32_t>(), 0, st));
  if (!P.no_retry) {
    q.n_per_frame = rq;
    gp.radius = 2.f * P.radius, gp.retry_n_per_frame = nullptr;
    VO_CHECK(vo_match_guided_dev(t->frames, 0, B, &q, &gp, nullptr, t->assigned.as<int32_t>(), nullptr,
                                 t->nm_first.as<int32_t>(), 0, st));
  }
  // the matches go into the frame's slots in the launch that gathers the pose problem (solve_pose)
  t->pend.assigned = t->assigned.as<int32_t>(), t->pend.qpoints = t->p0.as<double>(), t->pend.qflags = t->q0_flags.as<uint8_t>();
  t->pend.stride = t->n_last;
  return VO_OK;
}

// trackRefKeyFrame's search (visualOdometry.cpp:256-270): computeBow of the frames, searchByBoW(reference key-frame,
// frame) with Matcher(0.7); the key-frame's map points of the matches into the frame's slots, the pose starts at
// frame_last_->Tcw_ (vo_tracker_set_ref_keyframe).  The common-node walk is host work between two device steps
// (vo::bow_search_resident synchronises): this is the route of a frame that trackWithMotion has already given up on.
int stage_ref_keyframe(vo_tracker *t, const vo_tracker_params &P) {
  const int B = t->B;
  hipStream_t st = t->st;
  StageTimer tm(t, 2, st);
  if (!t->ref_vocab || (int)t->ref_kf.size() != B) {
    vo::set_error("vo_tracker_track_ref_keyframe: no reference key-frame (vo_tracker_set_ref_keyframe)");
    return VO_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_track_prep, dim3(B), dim3(256), 0, st, t->cap, t->n_last, t->assigned.as<int>(), t->fhas.as<uint8_t>(),
                     t->fobs.as<uint8_t>(), t->last_matched.as<uint8_t>(), t->pose0.as<double>(), t->pose.as<double>());
  std::vector<vo::RefKeyFrame> kfs((size_t)B);
  for (int f = 0; f < B; f++) {
    const RefKfHost &k = t->ref_kf[f];
    kfs[f] = vo::RefKeyFrame{k.n, k.valid.data(), k.desc.data(), k.angle.data(), &k.view};
  }
  VO_CHECK(vo::bow_search_resident(t->ref_vocab, t->frames, 0, B, kfs.data(), P.ref_ratio > 0.f ? P.ref_ratio : 0.7f, 1, 3,
                                   t->assigned.as<int32_t>(), t->cap, t->nm_first.as<int32_t>(), st));
  // the matches go into the frame's slots in the launch that gathers the pose problem (solve_pose)
  t->pend.assigned = t->assigned.as<int32_t>(), t->pend.qpoints = t->p0.as<double>(), t->pend.qflags = t->q0_flags.as<uint8_t>();
  t->pend.stride = t->n_last;
  return VO_OK;
}

InFrameArgs in_frame_args(vo_tracker *t) {
  const vo_tracker_config &c = t->cfg;
  return InFrameArgs{t->nq_local, t->n_local, t->pose.as<double>(), t->p1.as<double>(), t->nrm1.as<double>(), t->mind1.as<float>(),
                     t->maxd1.as<float>(), t->pf1.as<uint8_t>(), t->have_link ? t->link1.as<int>() : (const int *)nullptr,
                     t->last_matched.as<uint8_t>(), t->n_last, c.intrinsics[0], c.intrinsics[1], c.intrinsics[2], c.intrinsics[3],
                     c.intrinsics[4], 0.f, (float)c.width, 0.f, (float)c.height, (float)log((double)t->sf[1]), t->n_levels,
                     t->q1_flags.as<uint8_t>(), t->q1_u.as<float>(), t->q1_v.as<float>(), t->q1_aux.as<float>(), t->q1_level.as<int>(),
                     t->q1_viewcos.as<float>()};
}

// solvePoseOnlySE3 + cullingOutliersBeforeLocalMap (:249-250 / :271-272)
// (Round 5 tried the culling and the isInFrame pass of the local-map stage in ONE launch, one workgroup per frame with a barrier
//  between the two: 172 us per 1024 frames against 24 + 50 for the two launches -- a frame's ~1900 queries walked by 256 threads
//  in 8 dependent trips leave one wavefront per SIMD, where the separate kernel runs 8 workgroups per frame side by side.)
int stage_solve_cull(vo_tracker *t) {
  hipStream_t st = t->st;
  StageTimer tm(t, 3, st);
  VO_CHECK(solve_pose(t));
  const CullArgs C{t->cap, t->ranges.as<int>(), t->index.as<int>(), t->outlier.as<uint8_t>(), t->assigned.as<int>(), t->n_last,
                   t->fhas.as<uint8_t>(), t->fobs.as<uint8_t>(), t->last_matched.as<uint8_t>(), t->nobs_first.as<int>(),
                   t->assigned_first.as<int>(), t->pose.as<double>(), t->pose_first.as<double>(), t->ninl.as<int>(),
                   t->ninl_first.as<int>()};
  hipLaunchKernelGGL(k_track_cull, dim3(t->B), dim3(256), 0, st, C);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

// trackLocalMap from searchLocalMapPoints on (:745-775, :289-300)
int stage_local(vo_tracker *t, const vo_tracker_params &P) {
  const int B = t->B;
  hipStream_t st = t->st;
  const vo_tracker_config &c = t->cfg;
  vo_guided_queries q{};
  vo_guided_params gp{};
  gp.n_levels = t->n_levels, gp.scale_factors = t->sf;
  // ---- searchLocalMapPoints: isInFrame with the refined pose, then the search; occupied = holds an observed point
  {
    StageTimer tm(t, 4, st);
    if (t->nq_local > 0) {
      hipLaunchKernelGGL(k_track_in_frame, dim3((t->nq_local + 255) / 256, B), dim3(256), 0, st, in_frame_args(t));
      // (`assigned` was cleared by k_track_cull)
      q.n_queries = t->nq_local, q.stride = t->n_local, q.flags = t->q1_flags.as<uint8_t>(), q.u = t->q1_u.as<float>();
      q.v = t->q1_v.as<float>(), q.aux = t->q1_aux.as<float>(), q.level = t->q1_level.as<int32_t>();
      q.viewcos = t->q1_viewcos.as<float>(), q.desc = t->q1_desc.as<uint8_t>();
      gp.mode = 1, gp.radius = P.th_radius, gp.ratio = P.ratio, gp.bf = 0.f, gp.direction = 0, gp.check_rot = 0;
      VO_CHECK(vo_match_guided_dev(t->frames, 0, B, &q, &gp, t->fobs.as<uint8_t>(), t->assigned.as<int32_t>(), nullptr,
                                   t->nm.as<int32_t>(), 0, st));
      t->pend.assigned = t->assigned.as<int32_t>(), t->pend.qpoints = t->p1.as<double>(), t->pend.qflags = t->q1_flags.as<uint8_t>();
      t->pend.stride = t->n_local;
    } else {
      VO_HIP_CHECK(hipMemsetAsync(t->nm.p, 0, (size_t)B * 4, st));
    }
  }
  // ---- second solvePoseOnlySE3 and the inlier count of trackLocalMap
  {
    StageTimer tm(t, 5, st);
    VO_CHECK(solve_pose(t));
    PackArgs K;
    VO_CHECK(pack_args(t, K));
    hipLaunchKernelGGL(k_track_count, dim3(B), dim3(256), 0, st, t->cap, t->ranges.as<int>(), t->index.as<int>(),
                       t->outlier.as<uint8_t>(), t->fobs.as<uint8_t>(), t->nm_first.as<int>(), t->nobs_first.as<int>(),
                       t->ninl_first.as<int>(), t->ntracked.as<int>(), t->status.as<int>(), t->first_min_matches,
                       t->foutl.as<uint8_t>(), K);
    VO_HIP_CHECK(hipGetLastError());
  }
  return VO_OK;
}

int run_pipeline(vo_tracker *t, const uint8_t *dev_images, int img_pitch, size_t img_frame_stride, const void *dev_depth,
                 int depth_kind, size_t depth_frame_stride, int depth_pitch, const vo_tracker_params *prm, unsigned run) {
  // (the host depth image of THIS call, if any: taken off the handle before anything can fail, so that a call that ends early never
  //  leaves a host pointer behind for the next one)
  const void *host_depth = t->pend_depth;
  t->pend_depth = nullptr;
  VO_CHECK(begin_timed_call(t));
  t->pend.assigned = nullptr;  // (a call that failed half-way must not leave its matches to the next one)
  if (!(run == (kRunFront | kRunMotion | kRunLocal))) t->tslot = -1;  // the six stage timers describe the one-call tracked frame
  const int B = t->B;
  hipStream_t st = t->st;
  vo_tracker_params P;
  if (prm) {
    P = *prm;
  } else {
    P.radius = 15.f, P.th_radius = 3.f, P.ratio = 0.8f, P.direction = 0, P.no_retry = 0, P.ref_ratio = 0.7f;
  }
  if (run & kRunFront)
    VO_CHECK(stage_front(t, dev_images, img_pitch, img_frame_stride, dev_depth, depth_kind, depth_frame_stride, depth_pitch, host_depth));
  const size_t capB = (size_t)B * t->cap;
  if ((run & kRunMotion) && t->nq_last == 0) {
    // no last frame (the first frame of a sequence, visualOdometry.cpp:170-214): Frame construction only; the pose is
    // the one handed in, every count zero
    VO_HIP_CHECK(hipMemcpyAsync(t->pose.p, t->pose0.p, (size_t)B * 48, hipMemcpyDeviceToDevice, st));
    VO_HIP_CHECK(hipMemcpyAsync(t->pose_first.p, t->pose0.p, (size_t)B * 48, hipMemcpyDeviceToDevice, st));
    VO_HIP_CHECK(hipMemsetAsync(t->assigned.p, 0xff, capB * 4, st));
    VO_HIP_CHECK(hipMemsetAsync(t->assigned_first.p, 0xff, capB * 4, st));
    VO_HIP_CHECK(hipMemsetAsync(t->fhas.p, 0, capB, st));
    for (DevBuf *b : {&t->nm, &t->nm_first, &t->ninl, &t->ninl_first, &t->nobs_first, &t->ntracked, &t->status})
      VO_HIP_CHECK(hipMemsetAsync(b->p, 0, (size_t)B * 4, st));
    VO_CHECK(launch_pack(t));
    t->tslot = -1;  // Frame construction only: not a timed tracked frame
    return VO_OK;
  }
  if (run & kRunMotion) {
    t->first_min_matches = 20;  // :247
    VO_CHECK(stage_motion(t, P));
  } else if (run & kRunRefKeyFrame) {
    t->first_min_matches = 15;  // :268
    VO_CHECK(stage_ref_keyframe(t, P));
  }
  if (run & (kRunMotion | kRunRefKeyFrame)) VO_CHECK(stage_solve_cull(t));
  // (the result block is written by the last kernel of either route: k_track_count / k_track_first_status)
  if (run & kRunLocal) {
    VO_CHECK(stage_local(t, P));
  } else {
    // first stage only: the status word and the counts of trackWithMotion / trackRefKeyFrame
    VO_HIP_CHECK(hipMemsetAsync(t->nm.p, 0, (size_t)B * 4, st));
    PackArgs K;
    VO_CHECK(pack_args(t, K));
    hipLaunchKernelGGL(k_track_first_status, dim3((B + 255) / 256), dim3(256), 0, st, B, t->nm_first.as<int>(),
                       t->nobs_first.as<int>(), t->ntracked.as<int>(), t->status.as<int>(), t->first_min_matches, K);
    VO_HIP_CHECK(hipGetLastError());
  }
  if (t->tslot >= 0) t->tissued++, t->tslot = -1;
  return VO_OK;
}

}  // namespace

extern "C" {

int vo_tracker_create(vo_tracker **out, const vo_tracker_config *cfg) {
  if (!out || !cfg || cfg->batch < 1 || cfg->width < 64 || cfg->height < 64 || cfg->max_last < 1 || cfg->max_local < 0)
    return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  vo_tracker *t = new (std::nothrow) vo_tracker();
  if (!t) {
    vo::set_error("vo_tracker_create: out of host memory");
    return VO_ERR_HIP;
  }
  t->cfg = *cfg;
  t->B = cfg->batch;
  int rc = vo_orb_create(&t->orb, cfg->nfeatures > 0 ? cfg->nfeatures : 1000, cfg->scale_factor > 1.f ? cfg->scale_factor : 1.2f,
                         cfg->nlevels > 0 ? cfg->nlevels : 8, cfg->ini_th_fast > 0 ? cfg->ini_th_fast : 20,
                         cfg->min_th_fast > 0 ? cfg->min_th_fast : 7);
  if (rc != VO_OK) {
    delete t;
    return rc;
  }
  auto fail = [&](int code) {
    vo_tracker_destroy(t);
    return code;
  };
  if (cfg->stream) {
    t->st = (hipStream_t)cfg->stream;
  } else {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&t->st, hipStreamNonBlocking, hi) != hipSuccess) return fail(VO_ERR_HIP);
    t->own_st = true;
  }
  if (cfg->extract_stream) {
    t->est = (hipStream_t)cfg->extract_stream;
  } else if (cfg->single_stream) {
    t->est = t->st;
  } else {
    if (hipStreamCreateWithFlags(&t->est, hipStreamNonBlocking) != hipSuccess) return fail(VO_ERR_HIP);
    t->own_est = true;
  }
  if ((rc = vo_orb_set_stream(t->orb, t->est)) != VO_OK) return fail(rc);
  if (hipEventCreateWithFlags(&t->ev_extract, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&t->ev_build, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&t->ev_depth, hipEventDisableTiming) != hipSuccess ||
      hipStreamCreateWithFlags(&t->cst, hipStreamNonBlocking) != hipSuccess)
    return fail(VO_ERR_HIP);
  t->n_levels = vo_orb_levels(t->orb);
  if ((rc = vo_orb_scale_factors(t->orb, t->sf, nullptr)) != VO_OK) return fail(rc);
  t->kcap = vo_orb_max_keypoints(t->orb);
  t->cap = cfg->max_features > 0 ? cfg->max_features : std::max(256, (t->kcap + 63) / 64 * 64);
  t->n_last = cfg->max_last, t->n_local = std::max(cfg->max_local, 1);
  if ((rc = vo_frames_create(&t->frames, t->B, t->cap)) != VO_OK) return fail(rc);
  if ((rc = vo_frames_set_camera(t->frames, cfg->intrinsics, cfg->has_distortion ? cfg->dist_coef : nullptr, (float)cfg->width,
                                 (float)cfg->height)) != VO_OK)
    return fail(rc);
  if ((rc = alloc_all(t)) != VO_OK) return fail(rc);
  // the intrinsics as doubles behind the Tcw block (the pose solver reads them from device memory)
  {
    double cam5d[5];
    for (int i = 0; i < 5; i++) cam5d[i] = (double)cfg->intrinsics[i];
    if (hipMemcpy(t->Tcw.as<double>() + (size_t)t->B * 12, cam5d, sizeof(cam5d), hipMemcpyHostToDevice) != hipSuccess)
      return fail(VO_ERR_HIP);
  }
  *out = t;
  return VO_OK;
}

void vo_tracker_destroy(vo_tracker *t) {
  if (!t) return;
  if (t->st) (void)hipStreamSynchronize(t->st);
  if (t->est && t->est != t->st) (void)hipStreamSynchronize(t->est);
  if (t->cst) (void)hipStreamSynchronize(t->cst);
  for (hipEvent_t e : t->tev)
    if (e) (void)hipEventDestroy(e);
  if (t->ev_extract) (void)hipEventDestroy(t->ev_extract);
  if (t->ev_build) (void)hipEventDestroy(t->ev_build);
  if (t->ev_depth) (void)hipEventDestroy(t->ev_depth);
  if (t->frames) vo_frames_destroy(t->frames);
  if (t->orb) vo_orb_destroy(t->orb);
  for (DevBuf *b : {&t->kps, &t->desc, &t->cnt, &t->images, &t->depth, &t->q0_flags, &t->q0_u, &t->q0_v, &t->q0_aux, &t->q0_level,
                    &t->q0_angle, &t->q0_desc, &t->p0, &t->pf0, &t->last_matched, &t->q1_flags, &t->q1_u, &t->q1_v, &t->q1_aux,
                    &t->q1_level, &t->q1_viewcos, &t->q1_desc, &t->p1, &t->nrm1, &t->mind1, &t->maxd1, &t->pf1, &t->link1, &t->Tcw,
                    &t->pose0, &t->pose, &t->pose_first, &t->resblk, &t->assigned, &t->assigned_first, &t->nm, &t->nm_first, &t->fpoint,
                    &t->fhas, &t->fobs, &t->pts, &t->obs, &t->isg, &t->ranges, &t->index, &t->outlier, &t->ninl, &t->ninl_first,
                    &t->nobs_first, &t->ntracked, &t->status, &t->retry_nq, &t->foutl})
    b->release();
  if (t->own_st && t->st) (void)hipStreamDestroy(t->st);
  if (t->own_est && t->est) (void)hipStreamDestroy(t->est);
  if (t->cst) (void)hipStreamDestroy(t->cst);
  delete t;
}

int vo_tracker_info(const vo_tracker *t, int *batch, int *max_features, int *max_keypoints, int *n_levels) {
  if (!t) return VO_ERR_INVALID;
  if (batch) *batch = t->B;
  if (max_features) *max_features = t->cap;
  if (max_keypoints) *max_keypoints = t->kcap;
  if (n_levels) *n_levels = t->n_levels;
  return VO_OK;
}

vo_orb *vo_tracker_extractor(vo_tracker *t) { return t ? t->orb : nullptr; }
vo_frames *vo_tracker_frames(vo_tracker *t) { return t ? t->frames : nullptr; }
void *vo_tracker_stream(vo_tracker *t) { return t ? (void *)t->st : nullptr; }

int vo_tracker_set_last_frame(vo_tracker *t, int n, const double *Tcw12, const double *points, const uint8_t *flags,
                              const int32_t *octave, const float *angle, const uint8_t *desc) {
  if (!t || n < 0 || n > t->n_last || !Tcw12 || (n > 0 && (!points || !flags || !octave || !angle || !desc)))
    return VO_ERR_INVALID;
  const int B = t->B;
  // pose6 = log(Tcw) on the host (the solver's start value; Frame::setPose keeps the SE3 itself)
  std::vector<double> p6((size_t)B * 6);
  for (int f = 0; f < B; f++) VO_CHECK(vo_se3_log(Tcw12 + 12 * f, Tcw12 + 12 * f + 9, p6.data() + 6 * f));
  VO_CHECK(t->stage.reserve((size_t)B * 144));
  memcpy(t->stage.data(), Tcw12, (size_t)B * 96);
  memcpy(t->stage.data() + (size_t)B * 96, p6.data(), (size_t)B * 48);
  VO_HIP_CHECK(hipMemcpyAsync(t->Tcw.p, t->stage.data(), (size_t)B * 96, hipMemcpyHostToDevice, t->st));
  VO_HIP_CHECK(hipMemcpyAsync(t->pose0.p, t->stage.data() + (size_t)B * 96, (size_t)B * 48, hipMemcpyHostToDevice, t->st));
  VO_CHECK(put_rows(t, t->p0, points, n, t->n_last, 24, "points"));
  VO_CHECK(put_rows(t, t->pf0, flags, n, t->n_last, 1, "flags"));
  VO_CHECK(put_rows(t, t->q0_level, octave, n, t->n_last, 4, "octave"));
  VO_CHECK(put_rows(t, t->q0_angle, angle, n, t->n_last, 4, "angle"));
  VO_CHECK(put_rows(t, t->q0_desc, desc, n, t->n_last, 32, "desc"));
  t->nq_last = n;
  VO_HIP_CHECK(hipStreamSynchronize(t->st));  // the caller's arrays and the staging block are free again
  return VO_OK;
}

int vo_tracker_set_local_map(vo_tracker *t, int n, const double *points, const double *normals, const float *min_distance,
                             const float *max_distance, const uint8_t *flags, const int32_t *link, const uint8_t *desc) {
  if (!t || n < 0 || n > t->n_local || (n > 0 && (!points || !normals || !min_distance || !max_distance || !flags || !desc)))
    return VO_ERR_INVALID;
  VO_CHECK(put_rows(t, t->p1, points, n, t->n_local, 24, "points"));
  VO_CHECK(put_rows(t, t->nrm1, normals, n, t->n_local, 24, "normals"));
  VO_CHECK(put_rows(t, t->mind1, min_distance, n, t->n_local, 4, "min_distance"));
  VO_CHECK(put_rows(t, t->maxd1, max_distance, n, t->n_local, 4, "max_distance"));
  VO_CHECK(put_rows(t, t->pf1, flags, n, t->n_local, 1, "flags"));
  VO_CHECK(put_rows(t, t->q1_desc, desc, n, t->n_local, 32, "desc"));
  t->have_link = link != nullptr;
  if (link) VO_CHECK(put_rows(t, t->link1, link, n, t->n_local, 4, "link"));
  t->nq_local = n;
  VO_HIP_CHECK(hipStreamSynchronize(t->st));
  return VO_OK;
}

int vo_tracker_track_dev(vo_tracker *t, const uint8_t *dev_images, int image_pitch, size_t image_frame_stride,
                         const void *dev_depth, int depth_kind, size_t depth_frame_stride, int depth_pitch,
                         const vo_tracker_params *params) {
  if (!t || !dev_images || image_pitch < t->cfg.width || depth_kind < 0 || depth_kind > 2 || (depth_kind && !dev_depth))
    return VO_ERR_INVALID;
  return run_pipeline(t, dev_images, image_pitch, image_frame_stride, dev_depth, depth_kind, depth_frame_stride, depth_pitch,
                      params, kRunFront | kRunMotion | kRunLocal);
}

static int upload_host_frames(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind, size_t *npx_out, size_t *dsz_out) {
  const size_t B = t->B, npx = (size_t)t->cfg.width * t->cfg.height, dsz = depth_kind == 1 ? 4 : depth_kind == 2 ? 2 : 0;
  VO_CHECK(t->images.reserve(B * npx));
  if (dsz) VO_CHECK(t->depth.reserve(B * npx * dsz));
  // uploads on the extraction stream (the depth is first read behind the extraction, which the tracking stream waits for)
  if (t->est != t->st && t->have_build) VO_HIP_CHECK(hipStreamWaitEvent(t->est, t->ev_build, 0));
  VO_HIP_CHECK(hipMemcpyAsync(t->images.p, images, B * npx, hipMemcpyHostToDevice, t->est));
  // (the depth follows behind the extraction's launches, on the copy stream: stage_front)
  t->pend_depth = dsz ? depth : nullptr, t->pend_depth_bytes = B * npx * dsz;
  *npx_out = npx, *dsz_out = dsz;
  return VO_OK;
}

static int track_host(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind, const vo_tracker_params *params,
                      unsigned run) {
  if (!t || !images || depth_kind < 0 || depth_kind > 2 || (depth_kind && !depth)) return VO_ERR_INVALID;
  size_t npx = 0, dsz = 0;
  VO_CHECK(upload_host_frames(t, images, depth, depth_kind, &npx, &dsz));
  return run_pipeline(t, t->images.as<uint8_t>(), t->cfg.width, npx, dsz ? t->depth.p : nullptr, depth_kind, npx * dsz,
                      t->cfg.width * (int)dsz, params, run);
}

int vo_tracker_track_first(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind, const vo_tracker_params *params) {
  return track_host(t, images, depth, depth_kind, params, kRunFront | kRunMotion);
}

int vo_tracker_track_first_dev(vo_tracker *t, const uint8_t *dev_images, int image_pitch, size_t image_frame_stride,
                               const void *dev_depth, int depth_kind, size_t depth_frame_stride, int depth_pitch,
                               const vo_tracker_params *params) {
  if (!t || !dev_images || image_pitch < t->cfg.width || depth_kind < 0 || depth_kind > 2 || (depth_kind && !dev_depth))
    return VO_ERR_INVALID;
  return run_pipeline(t, dev_images, image_pitch, image_frame_stride, dev_depth, depth_kind, depth_frame_stride, depth_pitch,
                      params, kRunFront | kRunMotion);
}

int vo_tracker_track_local_map(vo_tracker *t, const vo_tracker_params *params) {
  if (!t) return VO_ERR_INVALID;
  if (!t->resblk.p) {
    vo::set_error("vo_tracker_track_local_map: no first stage has run (vo_tracker_track_first / _ref_keyframe_first)");
    return VO_ERR_INVALID;
  }
  return run_pipeline(t, nullptr, 0, 0, nullptr, 0, 0, 0, params, kRunLocal);
}

int vo_tracker_track_ref_keyframe(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind,
                                  const vo_tracker_params *params, int first_stage_only) {
  return track_host(t, images, depth, depth_kind, params, kRunFront | kRunRefKeyFrame | (first_stage_only ? 0u : kRunLocal));
}

int vo_tracker_track_ref_keyframe_dev(vo_tracker *t, const uint8_t *dev_images, int image_pitch, size_t image_frame_stride,
                                      const void *dev_depth, int depth_kind, size_t depth_frame_stride, int depth_pitch,
                                      const vo_tracker_params *params, int first_stage_only) {
  if (!t || !dev_images || image_pitch < t->cfg.width || depth_kind < 0 || depth_kind > 2 || (depth_kind && !dev_depth))
    return VO_ERR_INVALID;
  return run_pipeline(t, dev_images, image_pitch, image_frame_stride, dev_depth, depth_kind, depth_frame_stride, depth_pitch,
                      params, kRunFront | kRunRefKeyFrame | (first_stage_only ? 0u : kRunLocal));
}

int vo_tracker_set_ref_keyframe(vo_tracker *t, const vo_vocab *vocab, int n, const double *Tcw12, const double *points,
                                const uint8_t *flags, const float *angle, const uint8_t *desc, const vo_bow_view *const *nodes) {
  if (!t || !vocab || n < 0 || n > t->n_last || !Tcw12 || !nodes || (n > 0 && (!points || !flags || !angle || !desc)))
    return VO_ERR_INVALID;
  const int B = t->B;
  // every view is validated BEFORE anything is enqueued or any tracker state changes: a DBoW3::FeatureVector as CSR
  // has start[0] == 0, non-decreasing offsets, every feature index below n; an empty view may carry null arrays
  for (int f = 0; f < B; f++) {
    const vo_bow_view *v = nodes[f];
    if (!v || v->n_nodes < 0 || (v->n_nodes > 0 && (!v->node_id || !v->start || !v->feat))) {
      vo::set_error("vo_tracker_set_ref_keyframe: frame %d has no usable vocabulary-node view", f);
      return VO_ERR_INVALID;
    }
    if (v->n_nodes == 0) continue;
    if (v->start[0] != 0) {
      vo::set_error("vo_tracker_set_ref_keyframe: frame %d: start[0] = %d", f, v->start[0]);
      return VO_ERR_INVALID;
    }
    for (int j = 0; j < v->n_nodes; j++)
      if (v->start[j + 1] < v->start[j]) {
        vo::set_error("vo_tracker_set_ref_keyframe: frame %d: node offsets decrease at node %d", f, j);
        return VO_ERR_INVALID;
      }
    const int nf = v->start[v->n_nodes];
    for (int i = 0; i < nf; i++)
      if ((int)v->feat[i] < 0 || (int)v->feat[i] >= n) {
        vo::set_error("vo_tracker_set_ref_keyframe: frame %d names feature %u of %d", f, v->feat[i], n);
        return VO_ERR_INVALID;
      }
  }
  std::vector<double> p6((size_t)B * 6);
  for (int f = 0; f < B; f++) VO_CHECK(vo_se3_log(Tcw12 + 12 * f, Tcw12 + 12 * f + 9, p6.data() + 6 * f));
  VO_CHECK(t->stage.reserve((size_t)B * 144));
  memcpy(t->stage.data(), Tcw12, (size_t)B * 96);
  memcpy(t->stage.data() + (size_t)B * 96, p6.data(), (size_t)B * 48);
  VO_HIP_CHECK(hipMemcpyAsync(t->Tcw.p, t->stage.data(), (size_t)B * 96, hipMemcpyHostToDevice, t->st));
  VO_HIP_CHECK(hipMemcpyAsync(t->pose0.p, t->stage.data() + (size_t)B * 96, (size_t)B * 48, hipMemcpyHostToDevice, t->st));
  // the key-frame's map points take the place of the last frame's list: the tail of the pipeline (scatter, culling, the
  // `link` test of the local-map stage) indexes them exactly as it indexes frame_last_->mappoints_
  VO_CHECK(put_rows(t, t->p0, points, n, t->n_last, 24, "points"));
  VO_CHECK(put_rows(t, t->pf0, flags, n, t->n_last, 1, "flags"));
  VO_CHECK(put_rows(t, t->q0_flags, flags, n, t->n_last, 1, "flags"));
  std::vector<RefKfHost> kfs((size_t)B);
  for (int f = 0; f < B; f++) {
    RefKfHost &k = kfs[f];
    const vo_bow_view *v = nodes[f];
    k.n = n;
    k.valid.resize((size_t)n), k.angle.assign(angle + (size_t)f * n, angle + (size_t)(f + 1) * n);
    for (int i = 0; i < n; i++) k.valid[i] = flags[(size_t)f * n + i] & 1;
    k.desc.assign(desc + (size_t)f * n * 32, desc + (size_t)(f + 1) * n * 32);
    if (v->n_nodes > 0) {
      k.node_id.assign(v->node_id, v->node_id + v->n_nodes);
      k.start.assign(v->start, v->start + v->n_nodes + 1);
      k.feat.assign(v->feat, v->feat + v->start[v->n_nodes]);
    } else {
      k.start.assign(1, 0);
    }
    k.view.n_nodes = v->n_nodes, k.view.node_id = reinterpret_cast<const uint32_t *>(k.node_id.data()), k.view.start = k.start.data(),
    k.view.feat = k.feat.data();
  }
  VO_HIP_CHECK(hipStreamSynchronize(t->st));  // (the staging block may be reused by the next call)
  t->ref_vocab = vocab;  // only now: a call that failed above leaves the tracker's reference key-frame as it was
  t->ref_kf.swap(kfs);
  // the views point into the vectors' heap blocks, which the swap moved along with their owners
  t->nq_last = n;
  return VO_OK;
}

int vo_tracker_track(vo_tracker *t, const uint8_t *images, const void *depth, int depth_kind, const vo_tracker_params *params) {
  return track_host(t, images, depth, depth_kind, params, kRunFront | kRunMotion | kRunLocal);
}

int vo_tracker_results(vo_tracker *t, double *poses6, double *Tcw12, int32_t *n_tracked, int32_t *n_inliers,
                       int32_t *n_matches_last, int32_t *n_matches_local, int32_t *status) {
  if (!t) return VO_ERR_INVALID;
  const size_t B = t->B;
  if (!t->resblk.p) {  // nothing has been tracked yet
    vo::set_error("vo_tracker_results: no batch has been tracked");
    return VO_ERR_INVALID;
  }
  VO_CHECK(t->stage.reserve(B * 72 + 64));
  uint8_t *h = t->stage.data();
  VO_HIP_CHECK(hipMemcpyAsync(h, t->resblk.p, B * 72 + 8, hipMemcpyDeviceToHost, t->st));  // one download: k_track_pack's block
  VO_HIP_CHECK(hipStreamSynchronize(t->st));
  int32_t *dst[5] = {n_tracked, n_inliers, n_matches_last, n_matches_local, status};
  for (size_t f = 0; f < B; f++) {
    const double *p6 = reinterpret_cast<const double *>(h + f * 72);
    const int32_t *pi = reinterpret_cast<const int32_t *>(h + f * 72 + 48);
    if (poses6) memcpy(poses6 + 6 * f, p6, 48);
    if (Tcw12) VO_CHECK(vo_se3_exp(p6, Tcw12 + 12 * f, Tcw12 + 12 * f + 9));
    for (int k = 0; k < 5; k++)
      if (dst[k]) dst[k][f] = pi[k];
  }
  const int32_t *flags = reinterpret_cast<const int32_t *>(h + B * 72);
  if (flags[0] == 0 && flags[1] == 0) return VO_OK;
  // a sticky error flag of a stage is up (dropped key-points, exhausted candidate pools): report and clear it
  VO_CHECK(vo_orb_sync(t->orb));
  VO_CHECK(vo_match_guided_status(t->frames, t->st));
  return VO_OK;
}

int vo_tracker_get(vo_tracker *t, int what, void *dst, size_t dst_bytes) {
  if (!t || !dst) return VO_ERR_INVALID;
  const size_t B = t->B, cap = t->cap;
  const DevBuf *b = nullptr;
  size_t bytes = 0;
  switch (what) {
    case VO_TRACKER_ASSIGNED_LAST: b = &t->assigned_first, bytes = B * cap * 4; break;
    case VO_TRACKER_ASSIGNED_LOCAL: b = &t->assigned, bytes = B * cap * 4; break;
    case VO_TRACKER_POSE_FIRST: b = &t->pose_first, bytes = B * 48; break;
    case VO_TRACKER_INLIERS_FIRST: b = &t->ninl_first, bytes = B * 4; break;
    case VO_TRACKER_OBSERVED_INLIERS_FIRST: b = &t->nobs_first, bytes = B * 4; break;
    case VO_TRACKER_FEATURE_HAS_POINT: b = &t->fhas, bytes = B * cap; break;
    case VO_TRACKER_FEATURE_POINTS: b = &t->fpoint, bytes = B * cap * 24; break;
    case VO_TRACKER_LOCAL_FLAGS: b = &t->q1_flags, bytes = B * (size_t)t->n_local; break;
    case VO_TRACKER_LOCAL_U: b = &t->q1_u, bytes = B * (size_t)t->n_local * 4; break;
    case VO_TRACKER_LOCAL_V: b = &t->q1_v, bytes = B * (size_t)t->n_local * 4; break;
    case VO_TRACKER_LOCAL_UR: b = &t->q1_aux, bytes = B * (size_t)t->n_local * 4; break;
    case VO_TRACKER_LOCAL_LEVEL: b = &t->q1_level, bytes = B * (size_t)t->n_local * 4; break;
    case VO_TRACKER_LOCAL_VIEWCOS: b = &t->q1_viewcos, bytes = B * (size_t)t->n_local * 4; break;
    case VO_TRACKER_KEYPOINT_COUNTS: b = &t->cnt, bytes = B * 4; break;
    case VO_TRACKER_FEATURE_OUTLIER: b = &t->foutl, bytes = B * cap; break;
    default: return VO_ERR_INVALID;
  }
  if (dst_bytes < bytes) {
    vo::set_error("vo_tracker_get(%d): destination holds %zu bytes, %zu needed", what, dst_bytes, bytes);
    return VO_ERR_CAPACITY;
  }
  VO_HIP_CHECK(hipMemcpyAsync(dst, b->p, bytes, hipMemcpyDeviceToHost, t->st));
  VO_HIP_CHECK(hipStreamSynchronize(t->st));
  return VO_OK;
}

int vo_frames_construct(vo_frames *h, int slot, vo_orb *orb, const uint8_t *image, int width, int height, int stride,
                        const void *depth, int depth_kind, int depth_pitch_bytes, float inv_depth_scale,
                        vo_keypoint *keypoints, int capacity, int *n_keypoints) {
  if (!h || !orb || !image || width < 1 || height < 1 || stride < width || depth_kind < 0 || depth_kind > 2 ||
      (depth_kind && (!depth || depth_pitch_bytes < width * (depth_kind == 1 ? 4 : 2))) || !n_keypoints || capacity < 0)
    return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  hipStream_t st = vo::thread_stream();
  thread_local vo::ScratchBuf d_img, d_dep, d_kp, d_desc, d_cnt;
  thread_local vo::PinnedBuf stage;
  const int kcap = vo_orb_max_keypoints(orb);
  const size_t img_bytes = (size_t)stride * height, dep_bytes = depth_kind ? (size_t)depth_pitch_bytes * height : 0;
  VO_CHECK(d_img.reserve(img_bytes));
  VO_CHECK(d_dep.reserve(std::max<size_t>(dep_bytes, 64)));
  VO_CHECK(d_kp.reserve((size_t)kcap * sizeof(vo_keypoint)));
  VO_CHECK(d_desc.reserve((size_t)kcap * 32));
  VO_CHECK(d_cnt.reserve(64));
  VO_CHECK(vo_orb_set_stream(orb, st));
  VO_CHECK(vo::copy_h2d(d_img.p, image, img_bytes, st, "vo_frames_construct"));
  if (dep_bytes) VO_CHECK(vo::copy_h2d(d_dep.p, depth, dep_bytes, st, "vo_frames_construct"));
  VO_CHECK(vo_orb_extract_batch_dev(orb, d_img.as<uint8_t>(), 1, width, height, stride, img_bytes, d_kp.as<vo_keypoint>(),
                                    d_desc.as<uint8_t>(), kcap, d_cnt.as<int32_t>()));
  VO_CHECK(vo_frames_build_dev(h, slot, 1, d_kp.as<vo_keypoint>(), d_desc.as<uint8_t>(), d_cnt.as<int32_t>(), kcap,
                               dep_bytes ? d_dep.p : nullptr, depth_kind, dep_bytes, depth_pitch_bytes, inv_depth_scale, st));
  VO_CHECK(stage.reserve((size_t)kcap * sizeof(vo_keypoint) + 64));
  VO_CHECK(vo::copy_d2h(stage.data(), d_cnt.p, 4, st, "vo_frames_construct"));
  VO_CHECK(vo::copy_d2h(stage.data() + 64, d_kp.p, (size_t)kcap * sizeof(vo_keypoint), st, "vo_frames_construct"));
  VO_CHECK(vo::stream_sync(st, "vo_frames_construct"));
  VO_CHECK(vo_orb_sync(orb));
  int n = 0;
  memcpy(&n, stage.data(), 4);
  if (n > capacity) {
    vo::set_error("vo_frames_construct: %d key-points, capacity %d", n, capacity);
    return VO_ERR_CAPACITY;
  }
  if (keypoints && n > 0) memcpy(keypoints, stage.data() + 64, (size_t)n * sizeof(vo_keypoint));
  *n_keypoints = n;
  return VO_OK;
}

int vo_tracker_sync(vo_tracker *t) {
  if (!t) return VO_ERR_INVALID;
  VO_HIP_CHECK(hipStreamSynchronize(t->st));
  return VO_OK;
}

int vo_tracker_set_timing(vo_tracker *t, int enabled) {
  if (!t) return VO_ERR_INVALID;
  VO_CHECK(collect_timing(t));  // (synchronises only if timed calls are outstanding)
  t->timing = enabled != 0;
  t->tissued = 0, t->tslot = -1;
  for (double &m : t->tms) m = 0;
  t->tcalls = 0;
  return VO_OK;
}

int vo_tracker_get_timing(vo_tracker *t, double *ms, int *n_calls) {
  if (!t || !ms) return VO_ERR_INVALID;
  VO_CHECK(collect_timing(t));
  for (int s = 0; s < kStages; s++) ms[s] = t->tms[s], t->tms[s] = 0;
  if (n_calls) *n_calls = t->tcalls;
  t->tcalls = 0;
  return VO_OK;
}

}  // extern "C"
