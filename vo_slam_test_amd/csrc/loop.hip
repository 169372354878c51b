// loop.hip -- small batched kernels around the loop-closing and local-mapping callers of the hot path:
//   k_sim3_ransac   Sim3Solver::iterate's hypotheses (reference src/sim3Solver.cpp:98-280): Horn's closed form per
//                   sample triplet + checkInliers over all correspondences, one workgroup per hypothesis
//   k_triangulate   the 4 x 4 linear triangulation of LocalMapping::createNewMapPoints (src/localMapping.cpp:234-251)
//   k_bow_score     Map::score, the L1 similarity of two BoW vectors (src/map.cpp:335-376), one candidate per thread
//   k_rgb2gray      cv::cvtColor(CV_RGB2GRAY / CV_BGR2GRAY) of VisualOdometry::createFrame (src/visualOdometry.cpp:146-159)
// Compiled with -ffp-contract=off (float gates that must round like the x86-64 reference build).
#include "vo_common.h"

#include <vector>

namespace {

// symmetric 4 x 4 eigen-decomposition, cyclic Jacobi (Eigen::EigenSolver of a symmetric matrix / cv::SVD of a 4 x 4:
// same vectors up to sign and rounding)
__device__ void sym4_eigen(double A[4][4], double V[4][4], double w[4]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) V[i][j] = i == j;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 4; p++)
      for (int q = p + 1; q < 4; q++) off += A[p][q] * A[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < 4; p++)
      for (int q = p + 1; q < 4; q++) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; k++) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq, A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; k++) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk, A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq, V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 4; i++) w[i] = A[i][i];
}

// Sim3Solver::computeSim3 (:179-252)
__device__ void sim3_horn(const double P1[9], const double P2[9], int fix_scale, double R[9], double t[3], double &s) {
  double O1[3] = {0, 0, 0}, O2[3] = {0, 0, 0}, Pr1[3][3], Pr2[3][3], M[3][3];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) O1[k] += P1[3 * i + k], O2[k] += P2[3 * i + k];
  for (int k = 0; k < 3; k++) O1[k] /= 3.0, O2[k] /= 3.0;
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) Pr1[k][i] = P1[3 * i + k] - O1[k], Pr2[k][i] = P2[3 * i + k] - O2[k];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      M[a][b] = 0;
      for (int i = 0; i < 3; i++) M[a][b] += Pr2[a][i] * Pr1[b][i];
    }
  double N[4][4], V[4][4], w[4];
  N[0][0] = M[0][0] + M[1][1] + M[2][2];
  N[0][1] = N[1][0] = M[1][2] - M[2][1];
  N[0][2] = N[2][0] = M[2][0] - M[0][2];
  N[0][3] = N[3][0] = M[0][1] - M[1][0];
  N[1][1] = M[0][0] - M[1][1] - M[2][2];
  N[1][2] = N[2][1] = M[0][1] + M[1][0];
  N[1][3] = N[3][1] = M[2][0] + M[0][2];
  N[2][2] = -M[0][0] + M[1][1] - M[2][2];
  N[2][3] = N[3][2] = M[1][2] + M[2][1];
  N[3][3] = -M[0][0] - M[1][1] + M[2][2];
  sym4_eigen(N, V, w);
  int best = 0;
  for (int i = 1; i < 4; i++)
    if (w[i] > w[best]) best = i;
  double q0 = V[0][best], q1 = V[1][best], q2 = V[2][best], q3 = V[3][best];  // (w, x, y, z)
  const double nq = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  q0 /= nq, q1 /= nq, q2 /= nq, q3 /= nq;
  R[0] = 1 - 2 * (q2 * q2 + q3 * q3), R[1] = 2 * (q1 * q2 - q0 * q3), R[2] = 2 * (q1 * q3 + q0 * q2);
  R[3] = 2 * (q1 * q2 + q0 * q3), R[4] = 1 - 2 * (q1 * q1 + q3 * q3), R[5] = 2 * (q2 * q3 - q0 * q1);
  R[6] = 2 * (q1 * q3 - q0 * q2), R[7] = 2 * (q2 * q3 + q0 * q1), R[8] = 1 - 2 * (q1 * q1 + q2 * q2);
  double sc = 1.0;
  if (!fix_scale) {
    double nom = 0, den = 0;
    for (int i = 0; i < 3; i++)
      for (int a = 0; a < 3; a++) {
        const double p3 = R[3 * a] * Pr2[0][i] + R[3 * a + 1] * Pr2[1][i] + R[3 * a + 2] * Pr2[2][i];
        nom += Pr1[a][i] * p3, den += p3 * p3;
      }
    sc = nom / den;
  }
  s = sc;
  for (int a = 0; a < 3; a++) t[a] = O1[a] - sc * (R[3 * a] * O2[0] + R[3 * a + 1] * O2[1] + R[3 * a + 2] * O2[2]);
}

__device__ __forceinline__ void sim3_project(const double *R, const double *t, double s, const double *p, float fx, float fy,
                                             float cx, float cy, double &u, double &v) {  // Sim3Solver::project :290-313
  const double x = s * (R[0] * p[0] + R[1] * p[1] + R[2] * p[2]) + t[0];
  const double y = s * (R[3] * p[0] + R[4] * p[1] + R[5] * p[2]) + t[1];
  const double z = s * (R[6] * p[0] + R[7] * p[1] + R[8] * p[2]) + t[2];
  const double invz = 1.0 / z;
  u = (double)((float)(x * invz) * fx + cx), v = (double)((float)(y * invz) * fy + cy);
}

__global__ __launch_bounds__(256) void k_sim3_ransac(int n, const double *pc1, const double *pc2, const double *px1,
                                                     const double *px2, const int *me1, const int *me2, float fx, float fy,
                                                     float cx, float cy, const int *triplets, int fix_scale, int *counts,
                                                     uint8_t *flags, double *sims) {
  __shared__ double T[26];  // R (9) t (3) s | Ri (9) ti (3) si
  __shared__ int s_cnt;
  const int k = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    double P1[9], P2[9], R[9], t[3], s;
    for (int i = 0; i < 3; i++) {
      const int idx = triplets[3 * k + i];
      for (int a = 0; a < 3; a++) P1[3 * i + a] = pc1[3 * idx + a], P2[3 * i + a] = pc2[3 * idx + a];
    }
    sim3_horn(P1, P2, fix_scale, R, t, s);
    const double si = 1.0 / s;
    for (int a = 0; a < 9; a++) T[a] = R[a];
    for (int a = 0; a < 3; a++) T[9 + a] = t[a];
    T[12] = s;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) T[13 + 3 * a + b] = R[3 * b + a];
    for (int a = 0; a < 3; a++) T[22 + a] = -si * (T[13 + 3 * a] * t[0] + T[13 + 3 * a + 1] * t[1] + T[13 + 3 * a + 2] * t[2]);
    T[25] = si;
    for (int a = 0; a < 13; a++) sims[13LL * k + a] = T[a];
    s_cnt = 0;
  }
  __syncthreads();
  int local = 0;
  for (int i = tid; i < n; i += 256) {
    double u1, v1, u2, v2;
    sim3_project(T, T + 9, T[12], pc2 + 3 * i, fx, fy, cx, cy, u1, v1);
    sim3_project(T + 13, T + 22, T[25], pc1 + 3 * i, fx, fy, cx, cy, u2, v2);
    const double d1x = px1[2 * i] - u1, d1y = px1[2 * i + 1] - v1, d2x = px2[2 * i] - u2, d2y = px2[2 * i + 1] - v2;
    const float e1 = (float)(d1x * d1x + d1y * d1y), e2 = (float)(d2x * d2x + d2y * d2y);
    const bool in = e1 < (float)me1[i] && e2 < (float)me2[i];  // :268
    flags[(long long)k * n + i] = in ? 1 : 0;
    local += in;
  }
  if (local) atomicAdd(&s_cnt, local);
  __syncthreads();
  if (tid == 0) counts[k] = s_cnt;
}

__global__ __launch_bounds__(256) void k_triangulate(int n, const float *xn1, const float *xn2, const float *T1, const float *T2,
                                                     long long t2_stride, float *out, uint8_t *ok) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *t2 = T2 + (long long)i * t2_stride;
  float A[4][4];
  for (int c = 0; c < 4; c++) {
    A[0][c] = xn1[2 * i] * T1[8 + c] - T1[c];
    A[1][c] = xn1[2 * i + 1] * T1[8 + c] - T1[4 + c];
    A[2][c] = xn2[2 * i] * t2[8 + c] - t2[c];
    A[3][c] = xn2[2 * i + 1] * t2[8 + c] - t2[4 + c];
  }
  double G[4][4], V[4][4], w[4];
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) {
      G[a][b] = 0;
      for (int r = 0; r < 4; r++) G[a][b] += (double)A[r][a] * (double)A[r][b];
    }
  sym4_eigen(G, V, w);
  int best = 0;
  for (int q = 1; q < 4; q++)
    if (w[q] < w[best]) best = q;
  const float x3 = (float)V[3][best];
  const bool good = !(fabsf(x3) < 1e-8f);  // :245-246
  ok[i] = good;
  for (int a = 0; a < 3; a++) out[3 * i + a] = good ? (float)V[a][best] / x3 : 0.f;
}

// Map::score (map.cpp:335-376): L1 score of the query BoW vector against candidate c's (both ascending word ids)
__global__ __launch_bounds__(256) void k_bow_score(int nq, const int *qw, const double *qv, int n_cand, const int *start,
                                                   const int *cw, const double *cv, double *score) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n_cand) return;
  int i = 0, j = start[c];
  const int je = start[c + 1];
  double s = 0;
  while (i < nq && j < je) {
    const int a = qw[i], b = cw[j];
    if (a == b) {
      s += fabs(qv[i] - cv[j]) - fabs(qv[i]) - fabs(cv[j]);
      i++, j++;
    } else if (a < b) {
      i++;  // (lower_bound jumps of the reference visit the same matches)
    } else {
      j++;
    }
  }
  score[c] = -s / 2.0;
}

__global__ __launch_bounds__(256) void k_rgb2gray(const uint8_t *src, long long n_px, int channels, int first_is_red, uint8_t *dst) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_px) return;
  const int c0 = src[i * channels], c1 = src[i * channels + 1], c2 = src[i * channels + 2];
  const int r = first_is_red ? c0 : c2, b = first_is_red ? c2 : c0;
  dst[i] = (uint8_t)((r * 4899 + c1 * 9617 + b * 1868 + (1 << 13)) >> 14);
}

}  // namespace

extern "C" {

int vo_sim3_ransac_eval(int n, const double *cam1_points, const double *cam2_points, const double *pixels1,
                        const double *pixels2, const int32_t *max_err1, const int32_t *max_err2, const float cam4[4],
                        int n_hypotheses, const int32_t *triplets, int fix_scale, int32_t *counts, uint8_t *inlier_flags,
                        double *sims) {
  // all six correspondence arrays NULL: the correspondences this host thread uploaded with its previous call stay in place
  // (Sim3Solver::iterate evaluates one hypothesis per trip to keep rand() in step with the reference: five trips per call
  // of the loop closer, one upload -- ADVICE r4)
  const bool resident = !cam1_points && !cam2_points && !pixels1 && !pixels2 && !max_err1 && !max_err2;
  if (n < 0 || n_hypotheses < 0 || !cam4 || (n_hypotheses > 0 && (!triplets || !counts || !sims)) ||
      (n > 0 && !resident && (!cam1_points || !cam2_points || !pixels1 || !pixels2 || !max_err1 || !max_err2)))
    return VO_ERR_INVALID;
  if (n_hypotheses == 0) return VO_OK;
  for (int k = 0; k < 3 * n_hypotheses; k++)
    if (triplets[k] < 0 || triplets[k] >= n) {
      vo::set_error("vo_sim3_ransac_eval: sample index %d out of range", triplets[k]);
      return VO_ERR_INVALID;
    }
  VO_CHECK(vo::ensure_device());
  thread_local vo::ScratchBuf d1, d2, p1, p2, e1, e2, tr, cn, fl, sm;
  thread_local int resident_n = -1;
  hipStream_t st = vo::thread_stream();
  const char *W = "vo_sim3_ransac_eval";
  if (resident && n > 0) {
    // (the buffers themselves are checked, not only the count: vo_release_thread_scratch() on this thread frees them, and a
    //  failed upload leaves them short -- a launch with null or short pointers is a GPU memory fault, not an error code)
    const bool held = d1.p && d2.p && p1.p && p2.p && e1.p && e2.p && d1.bytes >= (size_t)n * 24 && d2.bytes >= (size_t)n * 24 &&
                      p1.bytes >= (size_t)n * 16 && p2.bytes >= (size_t)n * 16 && e1.bytes >= (size_t)n * 4 && e2.bytes >= (size_t)n * 4;
    if (resident_n != n || !held) {
      vo::set_error("vo_sim3_ransac_eval: no correspondences of this size are resident for this thread (%d requested, %d held%s)", n,
                    resident_n, held ? "" : ", buffers released");
      resident_n = -1;
      return VO_ERR_INVALID;
    }
  } else {
    resident_n = -1;
    VO_CHECK(vo::upload(d1, cam1_points, (size_t)n * 24, st, W));
    VO_CHECK(vo::upload(d2, cam2_points, (size_t)n * 24, st, W));
    VO_CHECK(vo::upload(p1, pixels1, (size_t)n * 16, st, W));
    VO_CHECK(vo::upload(p2, pixels2, (size_t)n * 16, st, W));
    VO_CHECK(vo::upload(e1, max_err1, (size_t)n * 4, st, W));
    VO_CHECK(vo::upload(e2, max_err2, (size_t)n * 4, st, W));
    resident_n = n;
  }
  VO_CHECK(vo::upload(tr, triplets, (size_t)n_hypotheses * 12, st, W));
  VO_CHECK(cn.reserve((size_t)n_hypotheses * 4));
  VO_CHECK(fl.reserve(std::max<size_t>((size_t)n_hypotheses * n, 64)));
  VO_CHECK(sm.reserve((size_t)n_hypotheses * 13 * 8));
  hipLaunchKernelGGL(k_sim3_ransac, dim3(n_hypotheses), dim3(256), 0, st, n, d1.as<double>(), d2.as<double>(), p1.as<double>(),
                     p2.as<double>(), e1.as<int>(), e2.as<int>(), cam4[0], cam4[1], cam4[2], cam4[3], tr.as<int>(), fix_scale,
                     cn.as<int>(), fl.as<uint8_t>(), sm.as<double>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(counts, cn.p, (size_t)n_hypotheses * 4, st, W));
  if (inlier_flags && n > 0) VO_CHECK(vo::copy_d2h(inlier_flags, fl.p, (size_t)n_hypotheses * n, st, W));
  VO_CHECK(vo::copy_d2h(sims, sm.p, (size_t)n_hypotheses * 13 * 8, st, W));
  return vo::stream_sync(st, W);
}

int vo_triangulate(int n, const float *xn1, const float *xn2, const float Tcw1[12], const float *Tcw2, int per_pair_pose2,
                   float *points, uint8_t *ok) {
  if (n < 0 || (n > 0 && (!xn1 || !xn2 || !Tcw1 || !Tcw2 || !points || !ok))) return VO_ERR_INVALID;
  if (n == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  thread_local vo::ScratchBuf a, b, t1, t2, o, k;
  hipStream_t st = vo::thread_stream();
  const char *W = "vo_triangulate";
  VO_CHECK(vo::upload(a, xn1, (size_t)n * 8, st, W));
  VO_CHECK(vo::upload(b, xn2, (size_t)n * 8, st, W));
  VO_CHECK(vo::upload(t1, Tcw1, 48, st, W));
  VO_CHECK(vo::upload(t2, Tcw2, per_pair_pose2 ? (size_t)n * 48 : 48, st, W));
  VO_CHECK(o.reserve((size_t)n * 12));
  VO_CHECK(k.reserve((size_t)n + 64));
  hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256), dim3(256), 0, st, n, a.as<float>(), b.as<float>(), t1.as<float>(),
                     t2.as<float>(), per_pair_pose2 ? 12LL : 0LL, o.as<float>(), k.as<uint8_t>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(points, o.p, (size_t)n * 12, st, W));
  VO_CHECK(vo::copy_d2h(ok, k.p, (size_t)n, st, W));
  return vo::stream_sync(st, W);
}

int vo_bow_score(int n_query, const int32_t *query_words, const double *query_values, int n_candidates,
                 const int32_t *cand_start, const int32_t *cand_words, const double *cand_values, double *scores) {
  if (n_query < 0 || n_candidates < 0 || (n_candidates > 0 && (!cand_start || !scores)) ||
      (n_query > 0 && (!query_words || !query_values)))
    return VO_ERR_INVALID;
  if (n_candidates == 0) return VO_OK;
  const int total = cand_start[n_candidates];
  if (total < 0 || (total > 0 && (!cand_words || !cand_values))) return VO_ERR_INVALID;
  VO_CHECK(vo::ensure_device());
  thread_local vo::ScratchBuf qw, qv, cs, cw, cv, sc;
  hipStream_t st = vo::thread_stream();
  const char *W = "vo_bow_score";
  VO_CHECK(vo::upload(qw, query_words, (size_t)n_query * 4, st, W));
  VO_CHECK(vo::upload(qv, query_values, (size_t)n_query * 8, st, W));
  VO_CHECK(vo::upload(cs, cand_start, (size_t)(n_candidates + 1) * 4, st, W));
  VO_CHECK(vo::upload(cw, cand_words, (size_t)total * 4, st, W));
  VO_CHECK(vo::upload(cv, cand_values, (size_t)total * 8, st, W));
  VO_CHECK(sc.reserve((size_t)n_candidates * 8));
  hipLaunchKernelGGL(k_bow_score, dim3((n_candidates + 255) / 256), dim3(256), 0, st, n_query, qw.as<int>(), qv.as<double>(),
                     n_candidates, cs.as<int>(), cw.as<int>(), cv.as<double>(), sc.as<double>());
  VO_HIP_CHECK(hipGetLastError());
  VO_CHECK(vo::copy_d2h(scores, sc.p, (size_t)n_candidates * 8, st, W));
  return vo::stream_sync(st, W);
}

int vo_rgb_to_gray_dev(const uint8_t *dev_src, long long n_pixels, int channels, int first_is_red, uint8_t *dev_dst,
                       void *hip_stream) {
  if (!dev_src || !dev_dst || n_pixels < 0 || (channels != 3 && channels != 4)) return VO_ERR_INVALID;
  if (n_pixels == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  hipLaunchKernelGGL(k_rgb2gray, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, dev_src, n_pixels,
                     channels, first_is_red, dev_dst);
  VO_HIP_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_rgb_to_gray(const uint8_t *src, long long n_pixels, int channels, int first_is_red, uint8_t *dst) {
  if (!src || !dst || n_pixels < 0 || (channels != 3 && channels != 4)) return VO_ERR_INVALID;
  if (n_pixels == 0) return VO_OK;
  VO_CHECK(vo::ensure_device());
  thread_local vo::ScratchBuf a, b;
  hipStream_t st = vo::thread_stream();
  VO_CHECK(vo::upload(a, src, (size_t)n_pixels * channels, st, "vo_rgb_to_gray"));
  VO_CHECK(b.reserve((size_t)n_pixels));
  VO_CHECK(vo_rgb_to_gray_dev(a.as<uint8_t>(), n_pixels, channels, first_is_red, b.as<uint8_t>(), st));
  VO_CHECK(vo::copy_d2h(dst, b.p, (size_t)n_pixels, st, "vo_rgb_to_gray"));
  return vo::stream_sync(st, "vo_rgb_to_gray");
}

}  // extern "C"
