// vo_run_hip.cpp -- the harness contract of the reference's test/vo_run.cpp (:24-58 associate.txt, :105-159 frame loop
// with a steady_clock around every tracked frame, median / mean report, :163-232 trajectory dump) on top of libvo_hip.so:
// a TUM-layout sequence directory in, a camera trajectory file and the tracking-time report out.  Host code is plain
// C++ against include/vo_hip.h -- no OpenCV, no Ceres, no DBoW3: images are decoded by vo_png_read, converted by
// vo_rgb_to_gray, tracked by vo_tracker (batch 1), bundle-adjusted by vo_ba_local_ba.
//
// What a frame goes through is BASELINE config 0 ("tracking + local BA"), the hot path of the reference's loop
// (visualOdometry.cpp:105-159 -> trackWithMotion :224-255, trackLocalMap :282-305, localMapping.cpp:38):
//   1. trackWithMotion: Frame construction, searchByProjection against the last frame's map points (with the 2 x radius
//      retry), solvePoseOnlySE3, cullingOutliersBeforeLocalMap                      vo_tracker_track_first
//   2. updateLocalKeyFrames / updateLocalMapPoints: the local map is DERIVED BETWEEN the two stages, as the reference
//      does (:286-291) -- here by the scripted map below                            host code of this file
//   3. trackLocalMap: isInFrame, searchByProjection against the local map points, solvePoseOnlySE3, inlier count
//                                                                                   vo_tracker_track_local_map
//   4. every `ba_every`-th frame: Optimizer::solveLocalBAPoseAndPoint over the last `window` frames
//                                                                                   vo_ba_create / vo_ba_local_ba
// What stands in for the parts of the reference that are out of scope (key-frame policy, map-point culling, covisibility
// graph, loop closing -- control plane): a SCRIPTED map.  Every frame is a key-frame; a feature with depth that ends up
// without a map point creates one (position through the frame's pose, descriptor of the feature, normal / distance range
// as MapPoint::updateNormalAndDepth computes them, mappoint.cpp:66-115); a matched inlier feature becomes an observation
// of its point; the local map of a frame = the points observed by the last `window` frames; the local BA runs on a fixed
// schedule over those frames (oldest one fixed) and its erased edges drop the observation.  tests/test_gpu_harness.py runs
// the same script on the CPU oracle and compares the poses.
//
//   g++ -O2 -std=c++17 examples/vo_run_hip.cpp -Iinclude -Lvo_slam_test_amd -lvo_hip -Wl,-rpath,$PWD/vo_slam_test_amd -o vo_run_hip
//   ./vo_run_hip <sequence_dir/> <camera_trajectory.txt> [max_frames] [fx fy cx cy bf depth_scale] [pose_dump.txt]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
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
constexpr int kWindow = 10;   // frames whose points form the local map and whose poses the local BA refines
constexpr int kBaEvery = 5;   // local BA after every 5th frame
constexpr int kMaxLast = 2048, kMaxLocal = 16384;

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

struct FrameRec {       // a frame of the sequence: pose and what Frame::Frame leaves of its features
  double Tcw[12];
  int n = 0;
  std::vector<float> ux, uy, ur, dep, angle;
  std::vector<int32_t> oct;
  std::vector<uint8_t> desc;
  std::vector<int> mp;  // map point per feature or -1
};
struct MapPoint {
  double p[3];
  uint8_t desc[32];
  double nsum[3];       // sum of the unit vectors camera centre -> point over its observations (normalVector_ * n)
  int ncnt = 0;         // ... and their number
  double ref_c[3];      // camera centre of the reference (creating) frame
  int level = 0, last_frame = -1;
  std::vector<std::pair<int, int>> obs;  // (frame, feature)
};

void cam_centre(const double Tcw[12], double c[3]) {
  for (int r = 0; r < 3; r++) c[r] = -(Tcw[r] * Tcw[9] + Tcw[3 + r] * Tcw[10] + Tcw[6 + r] * Tcw[11]);
}
void add_normal(MapPoint &m, const double c[3]) {
  const double d[3] = {m.p[0] - c[0], m.p[1] - c[1], m.p[2] - c[2]};
  const double nn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  for (int r = 0; r < 3; r++) m.nsum[r] += d[r] / nn;
  m.ncnt++;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <sequence_dir/> <camera_trajectory.txt> [max_frames] [fx fy cx cy bf depth_scale] [pose_dump.txt]\n", argv[0]);
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
  const char *dump_path = argc >= 11 ? argv[10] : nullptr;
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
  cfg.max_last = kMaxLast, cfg.max_local = kMaxLocal, cfg.single_stream = 1;
  vo_tracker *trk = nullptr;
  VO_TRY(vo_tracker_create(&trk, &cfg));
  int cap = 0, n_levels = 0;
  VO_TRY(vo_tracker_info(trk, nullptr, &cap, nullptr, &n_levels));
  float sf[16] = {0};
  VO_TRY(vo_orb_scale_factors(vo_tracker_extractor(trk), sf, nullptr));
  const double cam5d[5] = {cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};

  std::vector<uint8_t> color((size_t)W * H * 4), gray((size_t)W * H);
  std::vector<uint16_t> depth((size_t)W * H);
  std::vector<FrameRec> frames;
  std::vector<MapPoint> map;
  double Tcl[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  int lost = 0, n_ba = 0;
  std::vector<double> costs, traj;
  std::vector<std::string> stamps;
  std::vector<int> traj_frame;
  std::vector<int32_t> asg_first(cap), asg_local(cap);
  std::vector<uint8_t> fhas(cap), foutl(cap);
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
    FrameRec fr;
    // ---- stage 1: trackWithMotion against the last frame's map points
    double Tpred[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    std::vector<int> last_ids;
    {
      std::vector<double> pts;
      std::vector<uint8_t> flags, ldesc;
      std::vector<int32_t> loct;
      std::vector<float> lang;
      if (i > 0) {
        const FrameRec &L = frames[i - 1];
        compose(Tcl, L.Tcw, Tpred);  // frame_curr_->setPose(Tcl_ * frame_last_->Tcw_), :232
        for (int k = 0; k < L.n && (int)last_ids.size() < kMaxLast; k++) {
          if (L.mp[k] < 0) continue;
          const MapPoint &m = map[L.mp[k]];
          last_ids.push_back(L.mp[k]);
          pts.insert(pts.end(), m.p, m.p + 3);
          flags.push_back(3);
          loct.push_back(L.oct[k]), lang.push_back(L.angle[k]);
          ldesc.insert(ldesc.end(), m.desc, m.desc + 32);
        }
      }
      VO_TRY(vo_tracker_set_last_frame(trk, (int)last_ids.size(), Tpred, pts.data(), flags.data(), loct.data(), lang.data(), ldesc.data()));
      VO_TRY(vo_tracker_set_local_map(trk, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
      VO_TRY(vo_tracker_track_first(trk, gray.data(), depth.data(), 2, nullptr));
    }
    double pose6[6], Tcw[12];
    int32_t n_tracked = 0, n_inl = 0, n_m0 = 0, n_m1 = 0, status = 0;
    VO_TRY(vo_tracker_results(trk, pose6, Tcw, &n_tracked, &n_inl, &n_m0, &n_m1, &status));
    bool ok = i == 0 || status == 0;  // :247, :253
    // the frame's features (Frame::Frame's output)
    fr.ux.resize(cap), fr.uy.resize(cap), fr.ur.resize(cap), fr.dep.resize(cap), fr.angle.resize(cap), fr.oct.resize(cap);
    fr.desc.resize((size_t)cap * 32);
    VO_TRY(vo_frames_download(vo_tracker_frames(trk), 0, &fr.n, fr.ux.data(), fr.uy.data(), fr.oct.data(), fr.angle.data(), fr.ur.data(),
                              fr.dep.data(), fr.desc.data(), nullptr, nullptr, vo_tracker_stream(trk)));
    fr.mp.assign(fr.n, -1);
    // ---- stage 2: the local map derived from the window (updateLocalKeyFrames / updateLocalMapPoints, :286-291), then
    // trackLocalMap
    std::vector<int> local_ids;
    double paused = 0;  // host-side map bookkeeping of THIS harness inside the t1 .. t2 window (it scans the ever-growing
                        // scripted map; the reference keeps its local map incrementally): not tracking time
    if (ok && i > 0) {
      const auto p1 = std::chrono::steady_clock::now();
      std::vector<int> pos_in_last(map.size(), -1);
      for (size_t q = 0; q < last_ids.size(); q++) pos_in_last[last_ids[q]] = (int)q;
      std::vector<double> lp, ln;
      std::vector<float> lmin, lmax;
      std::vector<uint8_t> lflags, ldesc;
      std::vector<int32_t> link;
      for (size_t m = 0; m < map.size() && (int)local_ids.size() < kMaxLocal; m++) {
        const MapPoint &M = map[m];
        if (M.obs.empty() || M.last_frame < i - kWindow) continue;
        local_ids.push_back((int)m);
        lp.insert(lp.end(), M.p, M.p + 3);
        const double cnt = (double)M.ncnt;
        for (int r = 0; r < 3; r++) ln.push_back(M.nsum[r] / cnt);  // normalVector_ = normal / n
        const double d[3] = {M.p[0] - M.ref_c[0], M.p[1] - M.ref_c[1], M.p[2] - M.ref_c[2]};
        const float dist = (float)std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float mx = dist * sf[M.level];            // maxDistance_ = dist * levelScaledFactor
        lmax.push_back(mx), lmin.push_back(mx / sf[n_levels - 1]);
        lflags.push_back(3);
        link.push_back(pos_in_last[m]);
        ldesc.insert(ldesc.end(), M.desc, M.desc + 32);
      }
      paused = std::chrono::duration<double>(std::chrono::steady_clock::now() - p1).count();
      VO_TRY(vo_tracker_set_local_map(trk, (int)local_ids.size(), lp.data(), ln.data(), lmin.data(), lmax.data(), lflags.data(),
                                      link.data(), ldesc.data()));
      VO_TRY(vo_tracker_track_local_map(trk, nullptr));
      VO_TRY(vo_tracker_results(trk, pose6, Tcw, &n_tracked, &n_inl, &n_m0, &n_m1, &status));
      ok = n_tracked >= 30;  // trackLocalMap's verdict (:304)
    }
    const auto t2 = std::chrono::steady_clock::now();
    if (!ok) {
      lost++;
      memcpy(Tcw, Tpred, sizeof(Tcw));
    } else {
      costs.push_back(std::chrono::duration<double>(t2 - t1).count() - paused);
    }
    memcpy(fr.Tcw, Tcw, sizeof(Tcw));
    // ---- the scripted map: observations of the matched inliers, new points for the unmatched features with depth
    double Twc[12], centre[3];
    inverse(Tcw, Twc);
    cam_centre(Tcw, centre);
    if (ok && i > 0) {
      VO_TRY(vo_tracker_get(trk, VO_TRACKER_ASSIGNED_LAST, asg_first.data(), (size_t)cap * 4));
      VO_TRY(vo_tracker_get(trk, VO_TRACKER_ASSIGNED_LOCAL, asg_local.data(), (size_t)cap * 4));
      VO_TRY(vo_tracker_get(trk, VO_TRACKER_FEATURE_HAS_POINT, fhas.data(), (size_t)cap));
      VO_TRY(vo_tracker_get(trk, VO_TRACKER_FEATURE_OUTLIER, foutl.data(), (size_t)cap));
      for (int k = 0; k < fr.n; k++) {
        int m = -1;
        if (asg_local[k] >= 0) m = local_ids[asg_local[k]];
        else if (asg_first[k] >= 0 && fhas[k]) m = last_ids[asg_first[k]];
        if (m < 0 || foutl[k]) continue;  // outliers lose their point (cullingOutliersOfFrame, :888-905)
        bool seen = false;  // one observation per (point, frame)
        for (const auto &o : map[m].obs) seen |= o.first == i;
        if (seen) continue;
        fr.mp[k] = m;
        map[m].obs.push_back({i, k});
        map[m].last_frame = i;
        add_normal(map[m], centre);
      }
    }
    for (int k = 0; k < fr.n; k++) {
      if (fr.mp[k] >= 0 || !(fr.dep[k] > 0)) continue;
      MapPoint M;
      const double z = fr.dep[k], xc = ((double)fr.ux[k] - cam5[2]) * z / cam5[0], yc = ((double)fr.uy[k] - cam5[3]) * z / cam5[1];
      for (int r = 0; r < 3; r++) M.p[r] = Twc[3 * r] * xc + Twc[3 * r + 1] * yc + Twc[3 * r + 2] * z + Twc[9 + r];
      memcpy(M.desc, fr.desc.data() + (size_t)k * 32, 32);
      M.nsum[0] = M.nsum[1] = M.nsum[2] = 0;
      memcpy(M.ref_c, centre, sizeof(centre));
      M.level = fr.oct[k], M.last_frame = i;
      M.obs.push_back({i, k});
      add_normal(M, centre);
      fr.mp[k] = (int)map.size();
      map.push_back(M);
    }
    frames.push_back(std::move(fr));
    // ---- scripted local BA (localMapping.cpp:38 -> Optimizer::solveLocalBAPoseAndPoint) over the window
    if (i > 0 && i % kBaEvery == 0) {
      const int f0 = std::max(0, i - kWindow + 1), nc = i - f0 + 1;
      std::vector<double> poses((size_t)nc * 6), pts, eobs, eis;
      std::vector<uint8_t> fixed(nc, 0);
      fixed[0] = 1;  // the oldest frame of the window holds the gauge
      std::vector<int32_t> ecam, ept;
      std::vector<int> pid;
      std::vector<std::pair<int, int>> eref;  // (map point, index in its observation list)
      for (int c = 0; c < nc; c++) VO_TRY(vo_se3_log(frames[f0 + c].Tcw, frames[f0 + c].Tcw + 9, poses.data() + 6 * c));
      for (size_t m = 0; m < map.size(); m++) {
        int in_win = 0;
        for (const auto &o : map[m].obs) in_win += o.first >= f0;
        if (in_win < 2) continue;
        const int j = (int)pid.size();
        pid.push_back((int)m);
        pts.insert(pts.end(), map[m].p, map[m].p + 3);
        for (size_t q = 0; q < map[m].obs.size(); q++) {
          const auto &o = map[m].obs[q];
          if (o.first < f0) continue;
          const FrameRec &F = frames[o.first];
          ecam.push_back(o.first - f0), ept.push_back(j);
          eobs.push_back(F.ux[o.second]), eobs.push_back(F.uy[o.second]), eobs.push_back(F.ur[o.second]);
          eis.push_back(1.0 / (double)sf[F.oct[o.second]]);
          eref.push_back({(int)m, (int)q});
        }
      }
      if (!pid.empty()) {
        vo_ba *ba = nullptr;
        VO_TRY(vo_ba_create(&ba, nc, poses.data(), fixed.data(), (int)pid.size(), pts.data(), (int)ecam.size(), ecam.data(), ept.data(),
                            eobs.data(), eis.data(), cam5d));
        std::vector<uint8_t> erase(ecam.size(), 0);
        vo_lm_summary sums[2];
        const int rc = vo_ba_local_ba(ba, nullptr, erase.data(), sums);
        if (rc != VO_OK) {
          fprintf(stderr, "vo_ba_local_ba: status %d: %s\n", rc, vo_last_error());
          return 1;
        }
        VO_TRY(vo_ba_get_state(ba, poses.data(), pts.data()));
        vo_ba_destroy(ba);
        for (int c = 0; c < nc; c++) VO_TRY(vo_se3_exp(poses.data() + 6 * c, frames[f0 + c].Tcw, frames[f0 + c].Tcw + 9));
        for (size_t j = 0; j < pid.size(); j++) memcpy(map[pid[j]].p, pts.data() + 3 * j, 24);
        // erased edges (:757-800: eraseObservation / removeMapPoint): the observation goes, back to front per point
        int n_erased = 0;
        for (size_t e = ecam.size(); e-- > 0;) {
          if (!erase[e]) continue;
          MapPoint &M = map[eref[e].first];
          const auto o = M.obs[eref[e].second];
          frames[o.first].mp[o.second] = -1;
          M.obs.erase(M.obs.begin() + eref[e].second);
          n_erased++;
        }
        n_ba++;
        printf("local BA %d after frame %d: %d frames, %zu points, %zu edges, %d + %d LM iterations, cost %.6g -> %.6g, %d edges erased\n",
               n_ba, i, nc, pid.size(), ecam.size(), sums[0].iterations, sums[1].iterations, sums[0].initial_cost, sums[1].final_cost,
               n_erased);
      }
    }
    // ---- motion model (:168-172 of run(): Tcl_ = Tcw * Twl) with the poses as they stand after the BA
    if (i > 0) {
      if (ok) {
        double Twl[12];
        inverse(frames[i - 1].Tcw, Twl);
        compose(frames[i].Tcw, Twl, Tcl);
      } else {
        const double eye[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        memcpy(Tcl, eye, sizeof(eye));  // Tcl_ = SE3(), motionModel_ = false (:126-131)
      }
    }
    if (ok) {
      stamps.push_back(rt);
      traj_frame.push_back(i);
    }
    printf("frame %d: %d key-points, %d / %d matches, %d inliers (%d tracked from the map), status %d, %zu map points, local map %zu\n", i,
           frames[i].n, n_m0, n_m1, n_inl, n_tracked, status, map.size(), local_ids.size());
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
  // the trajectory as it stands at the end (poses refined by the local BA), like the reference's dump of the key-frame
  // poses after the run (vo_run.cpp:163-232)
  for (int f : traj_frame) {
    double Twc[12], q[4];
    inverse(frames[f].Tcw, Twc);
    quat_xyzw(Twc, q);
    traj.insert(traj.end(), {Twc[9], Twc[10], Twc[11], q[0], q[1], q[2], q[3]});
  }
  std::vector<const char *> ts;
  for (const std::string &s : stamps) ts.push_back(s.c_str());
  VO_TRY(vo_trajectory_write(out_path.c_str(), (int)ts.size(), ts.data(), traj.data()));
  printf("camera trajectory saved !!!\n");
  if (dump_path) {  // every pose at full precision (tests)
    FILE *fp = fopen(dump_path, "w");
    if (!fp) return 1;
    for (const FrameRec &F : frames) {
      for (int k = 0; k < 12; k++) fprintf(fp, "%.17g ", F.Tcw[k]);
      fprintf(fp, "\n");
    }
    fclose(fp);
  }
  vo_tracker_destroy(trk);
  vo_dataset_close(ds);
  return 0;
}
