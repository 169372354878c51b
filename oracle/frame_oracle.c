/*
 * frame_oracle.c -- CPU restatement of the frame post-processing and map-point descriptor selection
 * of guisongchen/vo_slam_test.  TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the
 * reference has no vectors for these routines; cv::undistortPoints is restated from OpenCV 3.x
 * (cvUndistortPoints: fixed 5 iterations, double arithmetic).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* Frame::undistortKeyPoints, frame.cpp:36-70 = cv::undistortPoints(mat, mat, K, distCoef, Mat(), K)
 * on CV_32FC2 points.  OpenCV 3.x cvUndistortPoints: x = (u - cx) * (1/fx), y likewise; 5 iterations of
 * x = (x0 - deltaX) * icdist; re-projection with RR = K * I; results rounded to float.  K and distCoef
 * are float matrices converted to double (camera.cpp:22-38).  k = k1 k2 p1 p2 k3.
 * distCoef[0] == 0 copies the key-points unchanged (:41-45). */
void orc_undistort_points(int n, const float *x, const float *y, const float intr[4], const float dist[5],
                          float *ux, float *uy) {
  if (dist == NULL || dist[0] == 0.0f) {
    for (int i = 0; i < n; i++) ux[i] = x[i], uy[i] = y[i];
    return;
  }
  const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const double ifx = 1. / fx, ify = 1. / fy;
  double k[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 5; i++) k[i] = dist[i];
  for (int i = 0; i < n; i++) {
    double xx = ((double)x[i] - cx) * ifx, yy = ((double)y[i] - cy) * ify;
    const double x0 = xx, y0 = yy;
    for (int j = 0; j < 5; j++) {
      const double r2 = xx * xx + yy * yy;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      const double deltaX = 2 * k[2] * xx * yy + k[3] * (r2 + 2 * xx * xx) + k[8] * r2 + k[9] * r2 * r2;
      const double deltaY = k[2] * (r2 + 2 * yy * yy) + 2 * k[3] * xx * yy + k[10] * r2 + k[11] * r2 * r2;
      xx = (x0 - deltaX) * icdist;
      yy = (y0 - deltaY) * icdist;
    }
    const double px = fx * xx + 0 * yy + cx, py = 0 * xx + fy * yy + cy, ww = 1. / (0 * xx + 0 * yy + 1.0);
    ux[i] = (float)(px * ww);
    uy[i] = (float)(py * ww);
  }
}

/* Frame::findDepth, frame.cpp:108-133: depth read at the ORIGINAL key-point with float -> int
 * truncation (Mat::at<float>(v, u) takes ints), uRight from the UNDISTORTED x.  depth: float metres
 * [h][stride] (the Mat after convertTo, visualOdometry.cpp:162-163). */
void orc_find_depth(int n, const float *x, const float *y, const float *ux, const float *depth_img, int w, int h,
                    int stride, float bf, float *uright, float *depth) {
  for (int i = 0; i < n; i++) {
    uright[i] = -1, depth[i] = -1;
    int u = (int)x[i], v = (int)y[i];
    if (u < 0) u = 0;
    if (u > w - 1) u = w - 1;
    if (v < 0) v = 0;
    if (v > h - 1) v = h - 1;
    const float d = depth_img[(size_t)v * stride + u];
    if (d > 0) {
      depth[i] = d;
      uright[i] = ux[i] - bf / d;
    }
  }
}

/* Mat::convertTo(CV_32F, alpha) for CV_16U (cvtScale_<ushort, float, float>): float multiply */
void orc_depth_to_float(const uint16_t *raw, int n, float inv_scale, float *out) {
  for (int i = 0; i < n; i++) out[i] = (float)raw[i] * inv_scale;
}

/* MapPoint::computeDescriptor, mappoint.cpp:118-179: index of the descriptor with the least median
 * distance to all n (itself included); -1 when n == 0. */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int orc_median_descriptor(const uint8_t *desc, int n) {
  if (n <= 0) return -1;
  int *dist = (int *)malloc(sizeof(int) * (size_t)n * n), *row = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) {
    dist[i * n + i] = 0;
    for (int j = i + 1; j < n; j++) {
      const int d = orc_hamming256(desc + (size_t)i * 32, desc + (size_t)j * 32);
      dist[i * n + j] = d, dist[j * n + i] = d;
    }
  }
  int bestMid = 256, bestIdx = 0;
  for (int i = 0; i < n; i++) {
    memcpy(row, dist + (size_t)i * n, sizeof(int) * n);
    qsort(row, n, sizeof(int), cmp_int);
    const int mid = row[(int)(0.5 * (n - 1))];
    if (mid < bestMid) bestMid = mid, bestIdx = i;
  }
  free(dist), free(row);
  return bestIdx;
}

/* ---- N3: Sim3Solver (reference src/sim3Solver.cpp) ------------------------------------------------------- */

/* symmetric 4 x 4 eigen-decomposition by cyclic Jacobi rotations (the reference calls Eigen::EigenSolver on the
 * symmetric matrix N, :211-213: same eigenvectors up to sign and rounding) */
static void sym4_eigen(double A[4][4], double V[4][4], double w[4]) {
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

/* Sim3Solver::computeSim3, sim3Solver.cpp:179-252 (Horn 1987).  P1, P2: three points each, point i = P[3 i .. 3 i + 2].
 * Out: R12 row-major, t12, s12 (1 when fix_scale). */
void orc_sim3_horn(const double P1[9], const double P2[9], int fix_scale, double R[9], double t[3], double *s) {
  double O1[3] = {0, 0, 0}, O2[3] = {0, 0, 0}, Pr1[3][3], Pr2[3][3], M[3][3];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) O1[k] += P1[3 * i + k], O2[k] += P2[3 * i + k];
  for (int k = 0; k < 3; k++) O1[k] /= 3.0, O2[k] /= 3.0;
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) Pr1[k][i] = P1[3 * i + k] - O1[k], Pr2[k][i] = P2[3 * i + k] - O2[k]; /* 3 x 3, column = point */
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      M[a][b] = 0;
      for (int i = 0; i < 3; i++) M[a][b] += Pr2[a][i] * Pr1[b][i]; /* M = Pr2 Pr1^T */
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
  double q0 = V[0][best], q1 = V[1][best], q2 = V[2][best], q3 = V[3][best]; /* (w, x, y, z) */
  const double nq = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  q0 /= nq, q1 /= nq, q2 /= nq, q3 /= nq;
  /* Eigen::Quaterniond::toRotationMatrix */
  R[0] = 1 - 2 * (q2 * q2 + q3 * q3), R[1] = 2 * (q1 * q2 - q0 * q3), R[2] = 2 * (q1 * q3 + q0 * q2);
  R[3] = 2 * (q1 * q2 + q0 * q3), R[4] = 1 - 2 * (q1 * q1 + q3 * q3), R[5] = 2 * (q2 * q3 - q0 * q1);
  R[6] = 2 * (q1 * q3 - q0 * q2), R[7] = 2 * (q2 * q3 + q0 * q1), R[8] = 1 - 2 * (q1 * q1 + q2 * q2);
  double sc = 1.0;
  if (!fix_scale) { /* :236-242 */
    double nom = 0, den = 0;
    for (int i = 0; i < 3; i++)
      for (int a = 0; a < 3; a++) {
        const double p3 = R[3 * a] * Pr2[0][i] + R[3 * a + 1] * Pr2[1][i] + R[3 * a + 2] * Pr2[2][i];
        nom += Pr1[a][i] * p3, den += p3 * p3;
      }
    sc = nom / den;
  }
  *s = sc;
  for (int a = 0; a < 3; a++) t[a] = O1[a] - sc * (R[3 * a] * O2[0] + R[3 * a + 1] * O2[1] + R[3 * a + 2] * O2[2]);
}

/* Sim3Solver::project (:290-313): float pixel arithmetic on a double camera-frame point */
static void sim3_project(const double R[9], const double t[3], double s, const double p[3], const float cam[4], double uv[2]) {
  const double x = s * (R[0] * p[0] + R[1] * p[1] + R[2] * p[2]) + t[0];
  const double y = s * (R[3] * p[0] + R[4] * p[1] + R[5] * p[2]) + t[1];
  const double z = s * (R[6] * p[0] + R[7] * p[1] + R[8] * p[2]) + t[2];
  const double invz = 1.0 / z;
  const float u = (float)(x * invz) * cam[0] + cam[2], v = (float)(y * invz) * cam[1] + cam[3];
  uv[0] = u, uv[1] = v;
}

/* One RANSAC hypothesis per sample triplet (:119-137) + checkInliers (:254-280) over all n correspondences.
 * pc1 / pc2: camera-frame points, px1 / px2: their pixels (Camera::camera2pixel, doubles), maxerr1 / maxerr2: the
 * reference's INTEGER thresholds (vector<int>, 9.210 sigma^2 truncated; sim3Solver.h).  Out per hypothesis k:
 * counts[k], flags[k * n ..], sims[k * 13 ..] = R12 (9), t12 (3), s12.  The sequential pick (first hypothesis whose
 * count beats the threshold, :141-160) is the caller's loop. */
void orc_sim3_ransac_eval(int n, const double *pc1, const double *pc2, const double *px1, const double *px2,
                          const int32_t *maxerr1, const int32_t *maxerr2, const float cam[4], int K, const int32_t *triplets,
                          int fix_scale, int32_t *counts, uint8_t *flags, double *sims) {
  for (int k = 0; k < K; k++) {
    double P1[9], P2[9], R[9], t[3], s;
    for (int i = 0; i < 3; i++)
      for (int a = 0; a < 3; a++) P1[3 * i + a] = pc1[3 * triplets[3 * k + i] + a], P2[3 * i + a] = pc2[3 * triplets[3 * k + i] + a];
    orc_sim3_horn(P1, P2, fix_scale, R, t, &s);
    /* T21 = T12^-1 (:250): s^-1 R^T, -s^-1 R^T t */
    double Ri[9], ti[3];
    const double si = 1.0 / s;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Ri[3 * a + b] = R[3 * b + a];
    for (int a = 0; a < 3; a++) ti[a] = -si * (Ri[3 * a] * t[0] + Ri[3 * a + 1] * t[1] + Ri[3 * a + 2] * t[2]);
    int cnt = 0;
    for (int i = 0; i < n; i++) {
      double a1[2], a2[2];
      sim3_project(R, t, s, pc2 + 3 * i, cam, a1);    /* points of frame 2 into image 1 */
      sim3_project(Ri, ti, si, pc1 + 3 * i, cam, a2); /* points of frame 1 into image 2 */
      const double d1x = px1[2 * i] - a1[0], d1y = px1[2 * i + 1] - a1[1];
      const double d2x = px2[2 * i] - a2[0], d2y = px2[2 * i + 1] - a2[1];
      const float err1 = (float)(d1x * d1x + d1y * d1y), err2 = (float)(d2x * d2x + d2y * d2y);
      const int in = err1 < (float)maxerr1[i] && err2 < (float)maxerr2[i];
      flags[(size_t)k * n + i] = (uint8_t)in;
      cnt += in;
    }
    counts[k] = cnt;
    memcpy(sims + 13 * (size_t)k, R, 72), memcpy(sims + 13 * (size_t)k + 9, t, 24), sims[13 * (size_t)k + 12] = s;
  }
}

/* ---- N4: linear triangulation of localMapping.cpp:234-251: A (4 x 4, float) from the two normalised
 * observations and the two 3 x 4 float poses, x = right singular vector of the smallest singular value
 * (cv::SVD::compute(A, w, u, vt, MODIFY_A | FULL_UV); vt.row(3)), divided by its last entry.  The SVD is restated as
 * the eigen-decomposition of A^T A in double (OpenCV's float one-sided Jacobi agrees to float rounding; parity is
 * stated as a tolerance, not bit-exact).  Returns 0 when |x[3]| < 1e-8 (:245-246 `continue`). */
int orc_triangulate(const float xn1[2], const float xn2[2], const float T1[12], const float T2[12], float out[3]) {
  float A[4][4];
  for (int c = 0; c < 4; c++) {
    A[0][c] = xn1[0] * T1[8 + c] - T1[c];
    A[1][c] = xn1[1] * T1[8 + c] - T1[4 + c];
    A[2][c] = xn2[0] * T2[8 + c] - T2[c];
    A[3][c] = xn2[1] * T2[8 + c] - T2[4 + c];
  }
  double G[4][4], V[4][4], w[4];
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) {
      G[a][b] = 0;
      for (int r = 0; r < 4; r++) G[a][b] += (double)A[r][a] * (double)A[r][b];
    }
  sym4_eigen(G, V, w);
  int best = 0;
  for (int i = 1; i < 4; i++)
    if (w[i] < w[best]) best = i;
  const float x3 = (float)V[3][best];
  if (fabsf(x3) < 1e-8f) return 0;
  for (int a = 0; a < 3; a++) out[a] = (float)V[a][best] / x3;
  return 1;
}

/* cv::cvtColor(CV_RGB2GRAY / CV_BGR2GRAY) for 8-bit images, OpenCV 3.x fixed point (visualOdometry.cpp:146-159):
 * Y = (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14.  `first_is_red` = the code was CV_RGB2GRAY. */
void orc_rgb_to_gray(const uint8_t *src, int n_px, int channels, int first_is_red, uint8_t *dst) {
  for (int i = 0; i < n_px; i++) {
    const int c0 = src[(size_t)i * channels], c1 = src[(size_t)i * channels + 1], c2 = src[(size_t)i * channels + 2];
    const int r = first_is_red ? c0 : c2, b = first_is_red ? c2 : c0;
    dst[i] = (uint8_t)((r * 4899 + c1 * 9617 + b * 1868 + (1 << 13)) >> 14);
  }
}

/* Frame::isInFrame, frame.cpp:145-190, with MapPoint::predictScale, mappoint.cpp:182-196, and the distance thresholds
 * of mappoint.cpp:391-401, for n map points against the pose Tcw = exp(pose6) (Frame::setPose, frame.cpp:100-105: Ow_ =
 * Tcw_.inverse().translation()).  valid[i] bit 0: the point takes part (exists, not bad, visualIdxOfFrame_ != frame id,
 * visualOdometry.cpp:762-766), bit 1: it has observations (copied to the output flag).  Outputs = trackInLocalMap_,
 * trackProj_u_ / _v_ / _uR_, trackScaleLevel_, viewCos_.  `log` on a float argument is std::log(float) = logf through
 * `using namespace std` (common_include.h:13); restated as the double logarithm rounded to float (glibc's logf is not
 * correctly rounded everywhere; the quotient only matters when it falls within an ulp of an integer). */
void orc_is_in_frame(int n, const double pose6[6], const double *pts, const double *normals, const float *min_dist,
                     const float *max_dist, const uint8_t *valid, const float intr5[5], float xmin, float xmax, float ymin,
                     float ymax, float scale_factor_1, int n_levels, uint8_t *flags, float *u_out, float *v_out,
                     float *ur_out, int32_t *level_out, float *viewcos_out) {
  double q[4], t[3];
  orc_se3_exp(pose6, q, t);
  const double qc[4] = {q[0], -q[1], -q[2], -q[3]}, zero[3] = {0, 0, 0}, nt[3] = {-t[0], -t[1], -t[2]};
  double ow[3];
  orc_se3_apply(qc, zero, nt, ow); /* SE3::inverse: (q^-1, q^-1 * (-t)) */
  const float fx = intr5[0], fy = intr5[1], cx = intr5[2], cy = intr5[3], bf = intr5[4];
  const float log_sf1 = (float)log((double)scale_factor_1);
  for (int i = 0; i < n; i++) {
    flags[i] = 0, u_out[i] = v_out[i] = ur_out[i] = viewcos_out[i] = 0.f, level_out[i] = 0;
    if (!(valid[i] & 1)) continue;
    double pc[3];
    orc_se3_apply(q, t, pts + 3 * i, pc);
    const float z = (float)pc[2];
    if (z < 0.0f) continue;
    const float u = (float)((double)fx * pc[0] / pc[2] + (double)cx); /* camera.cpp:72-75, float members */
    if (u < xmin || u > xmax) continue;
    const float v = (float)((double)fy * pc[1] / pc[2] + (double)cy);
    if (v < ymin || v > ymax) continue;
    const double l0 = pts[3 * i] - ow[0], l1 = pts[3 * i + 1] - ow[1], l2 = pts[3 * i + 2] - ow[2];
    const float dist = (float)sqrt(l0 * l0 + l1 * l1 + l2 * l2);
    const float mind = 0.8f * min_dist[i], maxd = 1.2f * max_dist[i];
    if (dist < mind || dist > maxd) continue;
    const float viewcos = (float)(l0 * normals[3 * i] + l1 * normals[3 * i + 1] + l2 * normals[3 * i + 2]) / dist;
    if (viewcos < 0.5f) continue;
    const float ratio = max_dist[i] / dist;
    int s = (int)ceilf((float)log((double)ratio) / log_sf1);
    if (s < 0) s = 0;
    else if (s >= n_levels) s = n_levels - 1;
    flags[i] = (uint8_t)(1 | (valid[i] & 2));
    u_out[i] = u, v_out[i] = v, ur_out[i] = u - bf / z, level_out[i] = s, viewcos_out[i] = viewcos;
  }
}
