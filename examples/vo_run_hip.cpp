// vo_run_hip.cpp -- the harness contract of the reference's test/vo_run.cpp (:24-58 associate.txt, :105-159 frame loop
// with a steady_clock around every tracked frame, median / mean report, :163-232 trajectory dump) on top of libvo_hip.so:
// a TUM-layout sequence directory in, a camera trajectory file and the tracking-time report out.  Host code is plain
// C++ against include/vo_hip.h -- no OpenCV, no Ceres, no DBoW3: images are decoded by vo_png_read, converted by
// vo_rgb_to_gray, tracked by vo_tracker (batch 1: Frame construction, searchByProjection against the last frame,
// solvePoseOnlySE3, culling, local-map stage, solvePoseOnlySE3 in one call).
//
// What stands in for the parts of the reference that are out of scope (map, local mapping, loop closing): the map a
// frame is tracked against is the last frame's own features back-projected with their depth through the last pose --
// what VisualOdometry::updateLastFrame creates as temporary points (visualOdometry.cpp:404-464), here flagged as observed
// points so that the inlier bookkeeping of trackWithMotion applies -- the local map is empty, the motion model is the
// last relative motion (:232).  A frame with fewer than 20 matches or fewer than 10 inliers counts as lost and keeps
// the predicted pose (:247-253).
//
//   g++ -O2 -std=c++17 examples/vo_run_hip.cpp -Iinclude -Lvo_slam_test_amd -lvo_hip -Wl,-rpath,$PWD/vo_slam_test_amd -o vo_run_hip
//   ./vo_run_hip <sequence_dir/> <camera_trajectory.txt> [max_frames] [fx fy cx cy bf depth_scale]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vo_hip.h"

#define VO_TRY(x)                                                                \
  do {                                                                           \
    int s_ = (x);                                                                \
    if (s_ != VO_OK) {                                                           \
      fprintf(stderr, "%s: status %d: %s\n", #x, s_, vo_last_error());           \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

namespace {
// Tcw as 12 doubles: rotation row-major, translation
void compose(const double A[12], const double B[12], double C[12]) {  // C = A * B
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    C[9 + i] = A[3 * i] * B[9] + A[3 * i + 1] * B[10] + A[3 * i + 2] * B[11] + A[9 + i];
  }
}
void inverse(const double A[12], double C[12]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * j + i];
  for (int i = 0; i < 3; i++) C[9 + i] = -(C[3 * i] * A[9] + C[3 * i + 1] * A[10] + C[3 * i + 2] * A[11]);
}
// Eigen's Quaterniond(R).coeffs() = x y z w (the branch structure of Eigen's quaternion-from-matrix)
void quat_xyzw(const double R[9], double q[4]) {
  const double t = R[0] + R[4] + R[8];
  if (t > 0) {
    double s = std::sqrt(t + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[7] - R[5]) * s, q[1] = (R[2] - R[6]) * s, q[2] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * s;
    s = 0.5 / s;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
  }
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <sequence_dir/> <camera_trajectory.txt> [max_frames] [fx fy cx cy bf depth_scale]\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1], out_path = argv[2];
  const int max_frames = argc > 3 ? atoi(argv[3]) : 1 << 30;
  // config/example.yaml:20-31,40 (TUM fr1)
  float cam5[5] = {517.306408f, 516.469215f, 318.643040f, 255.313989f, 40.0f};
  float depth_scale = 5000.0f;
  if (argc >= 10) {
    for (int i = 0; i < 5; i++) cam5[i] = (float)atof(argv[4 + i]);
    depth_scale = (float)atof(argv[9]);
  }
  vo_dataset *ds = nullptr;
  VO_TRY(vo_dataset_open(&ds, dir.c_str(), max_frames));
  const int n_img = vo_dataset_size(ds);
  if (n_img < 1) {
    fprintf(stderr, "empty sequence\n");
    return 1;
  }
  const char *rt, *rp, *dt, *dp;
  VO_TRY(vo_dataset_entry(ds, 0, &rt, &rp, &dt, &dp));
  int W = 0, H = 0, ch = 0, bits = 0;
  VO_TRY(vo_png_info(rp, &W, &H, &ch, &bits));

  vo_tracker_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.batch = 1, cfg.width = W, cfg.height = H;
  memcpy(cfg.intrinsics, cam5, sizeof(cam5));
  cfg.inv_depth_scale = 1.0f / depth_scale;
  cfg.max_last = 2048, cfg.max_local = 1, cfg.single_stream = 1;
  vo_tracker *trk = nullptr;
  VO_TRY(vo_tracker_create(&trk, &cfg));
  int cap = 0;
  VO_TRY(vo_tracker_info(trk, nullptr, &cap, nullptr, nullptr));

  std::vector<uint8_t> color((size_t)W * H * 4), gray((size_t)W * H), desc((size_t)cap * 32), flags;
  std::vector<uint16_t> depth((size_t)W * H);
  std::vector<float> x(cap), y(cap), angle(cap), ur(cap), dep(cap);
  std::vector<int32_t> oct(cap);
  std::vector<double> pts;
  double Tcw_last[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, Tcl[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  int n_last = 0, lost = 0;
  std::vector<double> costs, traj;
  std::vector<std::string> stamps;
  for (int i = 0; i < n_img; i++) {
    VO_TRY(vo_dataset_entry(ds, i, &rt, &rp, &dt, &dp));
    int w2, h2, c2, b2;
    if (vo_png_info(rp, &w2, &h2, &c2, &b2) != VO_OK || vo_png_info(dp, &w2, &h2, &c2, &b2) != VO_OK) {
      printf("no more image info.\n");  // vo_run.cpp:111-115
      break;
    }
    VO_TRY(vo_png_read(rp, 1, color.data(), color.size()));  // cv::imread(path, 1): BGR
    VO_TRY(vo_png_read(dp, 0, depth.data(), depth.size() * 2));  // cv::imread(path, -1): 16-bit
    const auto t1 = std::chrono::steady_clock::now();
    VO_TRY(vo_rgb_to_gray(color.data(), (long long)W * H, 3, 0, gray.data()));  // visualOdometry.cpp:146-159
    double Tpred[12];
    compose(Tcl, Tcw_last, Tpred);  // frame_curr_->setPose(Tcl_ * frame_last_->Tcw_), :232
    VO_TRY(vo_tracker_set_last_frame(trk, n_last, Tpred, pts.data(), flags.data(), oct.data(), angle.data(), desc.data()));
    VO_TRY(vo_tracker_track(trk, gray.data(), depth.data(), 2, nullptr));
    double pose6[6], Tcw[12];
    int32_t n_tracked = 0, n_inl = 0, n_m0 = 0, n_m1 = 0, status = 0;
    VO_TRY(vo_tracker_results(trk, pose6, Tcw, &n_tracked, &n_inl, &n_m0, &n_m1, &status));
    const auto t2 = std::chrono::steady_clock::now();
    const bool ok = i == 0 || (status == 0 && n_inl >= 10);
    if (!ok) {
      lost++;
      memcpy(Tcw, Tpred, sizeof(Tcw));
    } else {
      costs.push_back(std::chrono::duration<double>(t2 - t1).count());
    }
    // the map the next frame is tracked against: this frame's features with depth, through Twc
    int n = 0;
    VO_TRY(vo_frames_download(vo_tracker_frames(trk), 0, &n, x.data(), y.data(), oct.data(), angle.data(), ur.data(), dep.data(),
                              desc.data(), nullptr, nullptr, vo_tracker_stream(trk)));
    double Twc[12];
    inverse(Tcw, Twc);
    pts.assign((size_t)3 * n, 0.0), flags.assign(n, 0);
    for (int k = 0; k < n; k++) {
      if (!(dep[k] > 0)) continue;
      const double z = dep[k], xc = ((double)x[k] - cam5[2]) * z / cam5[0], yc = ((double)y[k] - cam5[3]) * z / cam5[1];
      for (int r = 0; r < 3; r++) pts[3 * k + r] = Twc[3 * r] * xc + Twc[3 * r + 1] * yc + Twc[3 * r + 2] * z + Twc[9 + r];
      flags[k] = 3;
    }
    n_last = n;
    if (ok && i > 0) {
      double Twl[12];
      inverse(Tcw_last, Twl);
      compose(Tcw, Twl, Tcl);  // Tcl_ = Tcw * Twl
    }
    memcpy(Tcw_last, Tcw, sizeof(Tcw));
    if (ok) {
      double q[4];
      quat_xyzw(Twc, q);
      stamps.push_back(rt);
      traj.insert(traj.end(), {Twc[9], Twc[10], Twc[11], q[0], q[1], q[2], q[3]});
    }
    printf("frame %d: %d key-points, %d / %d matches, %d inliers, status %d\n", i, n, n_m0, n_m1, n_inl, status);
  }
  const int tracked = (int)costs.size();
  printf("total tracked number: %d; total lost times: %d\n", tracked, lost);
  if (tracked > 0) {
    double median = 0, mean = 0;
    VO_TRY(vo_tracking_time_stats(costs.data(), tracked, &median, &mean));
    printf("median tracking time: %g\n", median);
    printf("mean tracking time: %g\n", mean);
  }
  printf("start saving camera trajectory...\n");
  std::vector<const char *> ts;
  for (const std::string &s : stamps) ts.push_back(s.c_str());
  VO_TRY(vo_trajectory_write(out_path.c_str(), (int)ts.size(), ts.data(), traj.data()));
  printf("camera trajectory saved !!!\n");
  vo_tracker_destroy(trk);
  vo_dataset_close(ds);
  return 0;
}
