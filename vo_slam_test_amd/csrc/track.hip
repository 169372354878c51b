// track.hip -- the per-frame glue between the device matcher and the device pose solver, so that a tracked
// frame never leaves HBM between vo_orb_extract_batch_dev and the pose:
//   k_track_project  the projection prologue of Matcher::searchByProjection(Frame*, Frame*) (reference
//                    src/matcher.cpp:41-64): last frame's map points through the current pose estimate
//   k_track_scatter  `frame_curr->mappoints_[idx] = mp` (matcher.cpp:110, :347 via the shim's write-back)
//   k_track_gather   the observation gather of Optimizer::solvePoseOnlySE3 (src/optimizer_ceres.cpp:181-202)
// Batched over frames; every frame is one independent tracking problem (replicas across GPUs).
#include "vo_common.h"

namespace {

// matcher.cpp:41-64.  Tcw: rotation row-major (9) + translation (3) per frame; point flags bit 0 = the last
// frame holds a map point here and it is no outlier (:48-49), bit 1 = observe_cnt_ > 0.
__global__ __launch_bounds__(256) void k_track_project(int nq, int stride, const double *Tcw, const double *points,
                                                       const uint8_t *pflags, float fx, float fy, float cx, float cy,
                                                       int xmin, int xmax, int ymin, int ymax, uint8_t *qflags, float *qu,
                                                       float *qv, float *qinvz) {
  const int f = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
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

// mappoints_[k] = the query that claimed feature k
__global__ __launch_bounds__(256) void k_track_scatter(int cap, int stride, const int *fn, int slot0, const int *assigned,
                                                       const double *qpoints, const uint8_t *qflags, double *fpoint,
                                                       uint8_t *fhas, uint8_t *fobserved) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= fn[slot0 + f]) return;
  const long long o = (long long)f * cap + i;
  const int a = assigned[o];
  if (a < 0) return;
  const long long qo = (long long)f * stride + a;
  fpoint[3 * o] = qpoints[3 * qo], fpoint[3 * o + 1] = qpoints[3 * qo + 1], fpoint[3 * o + 2] = qpoints[3 * qo + 2];
  fhas[o] = 1;
  if (fobserved) fobserved[o] = (qflags[qo] >> 1) & 1u;
}

// optimizer_ceres.cpp:181-202: one workgroup per frame compacts the features that hold a map point, in feature
// order, into the observation arrays of the pose solver (frame f owns slots [f * cap, f * cap + count)).
__global__ __launch_bounds__(256) void k_track_gather(int cap, const int *fn, int slot0, const float *X, const float *Y,
                                                      const float *UR, const int *OCT, const double *fpoint,
                                                      const uint8_t *fhas, const float *sf, double *pts, double *obs,
                                                      double *isg, int *ranges, int *index) {
  __shared__ int wsum[4];
  __shared__ int s_base;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = fn[slot0 + f];
  const long long so = (long long)(slot0 + f) * cap, o = (long long)f * cap;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int b = 0; b < n; b += 256) {
    const int i = b + tid;
    const bool has = i < n && fhas[o + i];
    const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
    const int within = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int pre = s_base;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    if (has) {
      const long long d = o + pre + within;
      pts[3 * d] = fpoint[3 * (o + i)], pts[3 * d + 1] = fpoint[3 * (o + i) + 1], pts[3 * d + 2] = fpoint[3 * (o + i) + 2];
      obs[3 * d] = (double)X[so + i], obs[3 * d + 1] = (double)Y[so + i], obs[3 * d + 2] = (double)UR[so + i];
      isg[d] = 1.0 / (double)sf[OCT[so + i]];  // :190
      if (index) index[d] = i;
    }
    __syncthreads();
    if (tid == 0) s_base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) ranges[2 * f] = (int)o, ranges[2 * f + 1] = s_base;
}

// k_track_scatter followed by k_track_gather in ONE launch (the tracker's use: a search's matches go into the frame's
// slots right before the pose solve that reads them): a feature that a query claimed takes the query's point -- written
// to its slot exactly as the scatter does -- and enters the observation list with it; a feature that held a point before
// enters with that one.  Same outputs, bit for bit, as the two kernels one after the other.
__global__ __launch_bounds__(256) void k_track_scatter_gather(int cap, int stride, const int *fn, int slot0, const int *assigned,
                                                              const double *qpoints, const uint8_t *qflags, double *fpoint,
                                                              uint8_t *fhas, uint8_t *fobserved, const float *X, const float *Y,
                                                              const float *UR, const int *OCT, const float *sf, double *pts,
                                                              double *obs, double *isg, int *ranges, int *index) {
  __shared__ int wsum[4];
  __shared__ int s_base;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = fn[slot0 + f];
  const long long so = (long long)(slot0 + f) * cap, o = (long long)f * cap;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int b = 0; b < n; b += 256) {
    const int i = b + tid;
    bool has = false;
    double p0 = 0, p1 = 0, p2 = 0;
    if (i < n) {
      const int a = assigned[o + i];
      if (a >= 0) {
        const long long qo = (long long)f * stride + a;
        p0 = qpoints[3 * qo], p1 = qpoints[3 * qo + 1], p2 = qpoints[3 * qo + 2];
        fpoint[3 * (o + i)] = p0, fpoint[3 * (o + i) + 1] = p1, fpoint[3 * (o + i) + 2] = p2;
        fhas[o + i] = 1;
        if (fobserved) fobserved[o + i] = (qflags[qo] >> 1) & 1u;
        has = true;
      } else if (fhas[o + i]) {
        p0 = fpoint[3 * (o + i)], p1 = fpoint[3 * (o + i) + 1], p2 = fpoint[3 * (o + i) + 2];
        has = true;
      }
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
    const int within = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int pre = s_base;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    if (has) {
      const long long d = o + pre + within;
      pts[3 * d] = p0, pts[3 * d + 1] = p1, pts[3 * d + 2] = p2;
      obs[3 * d] = (double)X[so + i], obs[3 * d + 1] = (double)Y[so + i], obs[3 * d + 2] = (double)UR[so + i];
      isg[d] = 1.0 / (double)sf[OCT[so + i]];  // :190
      if (index) index[d] = i;
    }
    __syncthreads();
    if (tid == 0) s_base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) ranges[2 * f] = (int)o, ranges[2 * f + 1] = s_base;
}

}  // namespace

extern "C" {

int vo_track_project_dev(int n_frames, int n_queries, int stride, const double *dev_Tcw, const double *dev_points,
                         const uint8_t *dev_point_flags, const float cam4[4], int xmin, int xmax, int ymin, int ymax,
                         uint8_t *dev_q_flags, float *dev_u, float *dev_v, float *dev_invz, void *hip_stream) {
  if (n_frames < 1 || n_queries < 0 || stride < n_queries || !dev_Tcw || !dev_points || !dev_point_flags || !cam4 ||
      !dev_q_flags || !dev_u || !dev_v || !dev_invz)
    return VO_ERR_INVALID;
  if (n_queries == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  hipLaunchKernelGGL(k_track_project, dim3((n_queries + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)hip_stream,
                     n_queries, stride, dev_Tcw, dev_points, dev_point_flags, cam4[0], cam4[1], cam4[2], cam4[3], xmin, xmax,
                     ymin, ymax, dev_q_flags, dev_u, dev_v, dev_invz);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

}  // extern "C"

// the two entry points below read the frame store: defined in guided.hip next to vo_frames
namespace vo {
void track_scatter_launch(int n_frames, int cap, int stride, const int *fn, int slot0, const int *assigned,
                          const double *qpoints, const uint8_t *qflags, double *fpoint, uint8_t *fhas, uint8_t *fobserved,
                          hipStream_t st) {
  hipLaunchKernelGGL(k_track_scatter, dim3((cap + 255) / 256, n_frames), dim3(256), 0, st, cap, stride, fn, slot0, assigned,
                     qpoints, qflags, fpoint, fhas, fobserved);
}
void track_gather_launch(int n_frames, int cap, const int *fn, int slot0, const float *X, const float *Y, const float *UR,
                         const int *OCT, const double *fpoint, const uint8_t *fhas, const float *sf, double *pts, double *obs,
                         double *isg, int *ranges, int *index, hipStream_t st) {
  hipLaunchKernelGGL(k_track_gather, dim3(n_frames), dim3(256), 0, st, cap, fn, slot0, X, Y, UR, OCT, fpoint, fhas, sf, pts,
                     obs, isg, ranges, index);
}
void track_scatter_gather_launch(int n_frames, int cap, int stride, const int *fn, int slot0, const int *assigned,
                                 const double *qpoints, const uint8_t *qflags, double *fpoint, uint8_t *fhas, uint8_t *fobserved,
                                 const float *X, const float *Y, const float *UR, const int *OCT, const float *sf, double *pts,
                                 double *obs, double *isg, int *ranges, int *index, hipStream_t st) {
  hipLaunchKernelGGL(k_track_scatter_gather, dim3(n_frames), dim3(256), 0, st, cap, stride, fn, slot0, assigned, qpoints, qflags,
                     fpoint, fhas, fobserved, X, Y, UR, OCT, sf, pts, obs, isg, ranges, index);
}
}  // namespace vo
