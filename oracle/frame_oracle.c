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
