// include/myslam_shim/optimizer_hip.inl -- replacement bodies for every member of myslam::Optimizer:
// solvePoseOnlySE3, solveLocalBAPoseAndPoint, solveLoopSim3 and solvePoseGraphLoop (reference
// src/optimizer_ceres.cpp:157-314, 446-808, 810-1030, 1036-1305).  #include at the bottom of a copy of
// optimizer_ceres.cpp from which those functions and the cost-function classes were removed: Ceres is no
// longer needed at all (only Eigen and Sophus, which the reference's own headers pull in).  The pointer-graph
// gather and the write-back stay on the host; the solves, the chi2 tests and the LM loops run on the GPU.
//
// Needs the reference's headers (Frame, KeyFrame, MapPoint, Map, LoopClosing, Sophus): compile inside the
// reference tree.  tests/test_shims_compile.py checks the syntax against stub declarations of those types.
#include <algorithm>
#include <cstring>
#include <iostream>
#include <map>
#include <set>
#include <vector>

#include "vo_hip.h"

namespace myslam {

static inline void se3_to_array(const SE3 &T, double out[6]) {
  Eigen::Matrix<double, 6, 1> xi = T.log();  // [upsilon; omega], like :163 / :474
  for (int i = 0; i < 6; i++) out[i] = xi[i];
}

int Optimizer::solvePoseOnlySE3(Frame *frame) {
  double pose[6];
  se3_to_array(frame->Tcw_, pose);
  Camera *c = frame->camera_;
  const double cam[5] = {c->fx_, c->fy_, c->cx_, c->cy_, c->bf_};  // float members widened, :166-172
  std::vector<double> pts, obs, isg;
  std::vector<int> index;
  {
    unique_lock<mutex> lock(MapPoint::mutexOptimizer_);
    for (int i = 0, N = (int)frame->mappoints_.size(); i < N; i++) {
      MapPoint *mp = frame->mappoints_[i];
      if (!mp) continue;
      const cv::KeyPoint &kp = frame->unKeypoints_[i];
      const Vector3d p = mp->getPose();
      pts.insert(pts.end(), {p[0], p[1], p[2]});
      obs.insert(obs.end(), {(double)kp.pt.x, (double)kp.pt.y, (double)frame->uRight_[i]});
      isg.push_back(1.0 / static_cast<double>(frame->scaleFactors_[kp.octave]));
      index.push_back(i);
      frame->outliers_[i] = false;
    }
  }
  if (index.empty()) return 0;  // :204-205
  const int32_t offsets[2] = {0, (int32_t)index.size()};
  std::vector<uint8_t> outlier(index.size());
  int32_t inliers = 0;
  if (vo_pose_only_solve(1, offsets, pts.data(), obs.data(), isg.data(), cam, pose, outlier.data(), &inliers,
                         nullptr) != VO_OK)
    return 0;  // no error channel in the reference: report "no inliers", leave the pose untouched
  for (size_t k = 0; k < index.size(); k++) frame->outliers_[index[k]] = outlier[k] != 0;
  Eigen::Map<const Eigen::Matrix<double, 6, 1>> se3(pose);
  frame->setPose(SE3::exp(se3));  // :311
  return inliers;
}

void Optimizer::solveLocalBAPoseAndPoint(KeyFrame *keyframe, bool &stopFlag, Map *map_curr) {
  // ---- gather (the same sets the reference builds at :449-528)
  std::vector<KeyFrame *> cams;
  std::map<KeyFrame *, int> cam_index;
  std::vector<uint8_t> fixed;
  auto add_cam = [&](KeyFrame *kf, bool is_fixed) {
    cam_index[kf] = (int)cams.size();
    cams.push_back(kf);
    fixed.push_back(is_fixed || kf->id_ == 0);  // :578-579
  };
  add_cam(keyframe, false);
  keyframe->localBAKFId_ = keyframe->id_;
  for (KeyFrame *kf : keyframe->getOrderedKFs()) {
    kf->localBAKFId_ = keyframe->id_;
    if (!kf->isBad()) add_cam(kf, false);
  }
  std::vector<MapPoint *> points;
  const size_t n_local = cams.size();
  for (size_t k = 0; k < n_local; k++)
    for (MapPoint *mp : cams[k]->getMapPoints())
      if (mp && !mp->isBad() && mp->localBAKFId_ != keyframe->id_) {
        mp->localBAKFId_ = keyframe->id_;
        points.push_back(mp);
      }
  for (MapPoint *mp : points)
    for (auto &ob : mp->getObservedKFs()) {
      KeyFrame *kf = ob.first;
      if (kf->localBAKFId_ != keyframe->id_ && kf->BAFixId_ != keyframe->id_) {
        kf->BAFixId_ = keyframe->id_;
        if (!kf->isBad()) add_cam(kf, true);
      }
    }
  std::vector<double> poses(6 * cams.size()), pts(3 * points.size()), e_obs, e_is;
  std::vector<int32_t> e_cam, e_pt;
  std::vector<std::pair<KeyFrame *, MapPoint *>> edges;
  for (size_t k = 0; k < cams.size(); k++) se3_to_array(cams[k]->getPose(), &poses[6 * k]);
  for (size_t j = 0; j < points.size(); j++) {
    const Vector3d p = points[j]->getPose();
    pts[3 * j] = p[0], pts[3 * j + 1] = p[1], pts[3 * j + 2] = p[2];
    for (auto &ob : points[j]->getObservedKFs()) {  // same visiting order as :548-590
      auto it = cam_index.find(ob.first);
      if (it == cam_index.end()) continue;  // bad key-frame that never got a pose block
      const cv::KeyPoint &kp = ob.first->unKeypoints_[ob.second];
      e_cam.push_back(it->second), e_pt.push_back((int32_t)j);
      e_obs.insert(e_obs.end(), {(double)kp.pt.x, (double)kp.pt.y, (double)ob.first->uRight_[ob.second]});
      e_is.push_back(1.0 / static_cast<double>(ob.first->scaleFactors_[kp.octave]));
      edges.emplace_back(ob.first, points[j]);
    }
  }
  Camera *c = keyframe->camera_;
  const double cam[5] = {c->fx_, c->fy_, c->cx_, c->cy_, c->bf_};
  // ONE handle per calling thread, re-used for every key-frame's problem (vo_ba_reset): LocalMapping::run builds a new local
  // window per key-frame (localMapping.cpp:38), and a create / destroy pair per call would put hipMalloc, a stream and a
  // device-synchronising hipFree on the mapping thread's hot path (2.5 ms against 1.1 ms per call at 10 key-frames x 3000
  // points, bench.py local_ba.end_to_end).  Never destroyed: thread-exit order against the HIP runtime's teardown.
  static thread_local vo_ba *h = nullptr;
  const int rc_new = h ? vo_ba_reset(h, (int)cams.size(), poses.data(), fixed.data(), (int)points.size(), pts.data(), (int)edges.size(),
                                     e_cam.data(), e_pt.data(), e_obs.data(), e_is.data(), cam)
                       : vo_ba_create(&h, (int)cams.size(), poses.data(), fixed.data(), (int)points.size(), pts.data(),
                                      (int)edges.size(), e_cam.data(), e_pt.data(), e_obs.data(), e_is.data(), cam);
  if (rc_new != VO_OK) return;
  // ---- solve (problem 1, chi2, problem 2, chi2) with the reference's stopFlag polling points: the library
  // reads the caller's LIVE flag (a bool written by the tracking thread, localMapping.cpp:72,540) as a byte at
  // :594 and again at :612, so a stop raised while problem 1 is being queued is still seen
  std::vector<uint8_t> erase(edges.size() + 1, 0);
  static_assert(sizeof(bool) == 1, "stopFlag is polled as a byte");
  const int rc = vo_ba_local_ba(h, reinterpret_cast<const volatile unsigned char *>(&stopFlag), erase.data(), nullptr);
  if (rc != VO_OK) return;  // VO_ERR_STOPPED mirrors the early return at :594-595 (no write-back)
  vo_ba_get_state(h, poses.data(), pts.data());  // (the state came back with the results: no round trip of its own)
  // ---- write-back (:757-804)
  unique_lock<mutex> lock(map_curr->mutexMapUpdate_);
  for (size_t e = 0; e < edges.size(); e++)
    if (erase[e]) {
      const int idx = edges[e].second->getIndexInKeyFrame(edges[e].first);
      if (idx > 0) edges[e].first->setMapPointNull(idx);  // Q-B3: feature 0 is skipped there too
      edges[e].second->eraseObservedKF(edges[e].first);
    }
  for (size_t k = 0; k < cams.size(); k++) {
    if (fixed[k]) continue;
    Eigen::Map<const Eigen::Matrix<double, 6, 1>> se3(&poses[6 * k]);
    SE3 Tcw = SE3::exp(se3);  // KeyFrame::setPose takes a non-const reference (keyframe.h)
    cams[k]->setPose(Tcw);
  }
  for (size_t j = 0; j < points.size(); j++) {
    points[j]->setPose(Vector3d(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]));
    points[j]->updateNormalAndDepth();
  }
}

// Optimizer::solveLoopSim3, reference optimizer_ceres.cpp:810-1030
int Optimizer::solveLoopSim3(KeyFrame *kf_curr, KeyFrame *kf_match, vector<MapPoint *> &inlierMappoints,
                             Sophus::Sim3 &Scm, const bool &fixScaleFlag) {
  const SE3 Tcw = kf_curr->getPose(), Tmw = kf_match->getPose();
  const vector<MapPoint *> mps1 = kf_curr->getMapPoints();
  double pose[6], scale = Scm.scale();
  const Matrix3d Rcm = Scm.rotation_matrix();
  const Vector3d tcm = Scm.translation();
  {  // angle-axis of Rcm (:826 uses ceres::RotationMatrixToAngleAxis; any log map of SO(3) gives the same vector)
    double R[9], t0[3] = {0, 0, 0}, xi[6];
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = Rcm(r, cc);
    vo_se3_log(R, t0, xi);
    pose[0] = xi[3], pose[1] = xi[4], pose[2] = xi[5];
  }
  memcpy(pose + 3, tcm.data(), 3 * sizeof(double));
  Camera *c = kf_curr->camera_;
  const double camera[4] = {c->fx_, c->fy_, c->cx_, c->cy_};
  std::vector<double> cam_m, pix_c, is_c, cam_c, pix_m, is_m;
  std::vector<int> index;
  for (int i = 0, N = (int)inlierMappoints.size(); i < N; i++) {  // :846-878
    MapPoint *mpm = inlierMappoints[i], *mpc = mps1[i];
    if (!mpm || mpm->isBad() || !mpc || mpc->isBad()) continue;
    const int im = mpm->getIndexInKeyFrame(kf_match);
    if (im < 0) continue;
    const Vector3d pm = Tmw * mpm->getPose(), pc = Tcw * mpc->getPose();
    const cv::KeyPoint &km = kf_match->unKeypoints_[im], &kc = kf_curr->unKeypoints_[i];
    cam_m.insert(cam_m.end(), {pm[0], pm[1], pm[2]});
    pix_m.insert(pix_m.end(), {(double)km.pt.x, (double)km.pt.y});
    is_m.push_back(1.0 / static_cast<double>(kf_match->scaleFactors_[km.octave]));
    cam_c.insert(cam_c.end(), {pc[0], pc[1], pc[2]});
    pix_c.insert(pix_c.end(), {(double)kc.pt.x, (double)kc.pt.y});
    is_c.push_back(1.0 / static_cast<double>(kf_curr->scaleFactors_[kc.octave]));
    index.push_back(i);
  }
  const int32_t offsets[2] = {0, (int32_t)index.size()};
  std::vector<uint8_t> outlier(index.size() + 1, 0);
  int32_t inliers = 0;
  vo_lm_summary sums[2];
  if (vo_sim3_solve(1, offsets, cam_m.data(), pix_c.data(), is_c.data(), cam_c.data(), pix_m.data(), is_m.data(), camera,
                    fixScaleFlag ? 1 : 0, pose, &scale, outlier.data(), &inliers, sums) != VO_OK)
    return 0;  // no error channel in the reference: "no inliers", Scm untouched
  for (size_t k = 0; k < index.size(); k++)
    if (outlier[k]) inlierMappoints[index[k]] = static_cast<MapPoint *>(nullptr);
  if (sums[0].reserved != 2) return 0;  // fewer than 10 survivors of problem 1: :950-951 returns before Scm is written
  // both problems ran: Scm = Scm2 even when the final test leaves no inlier (:1024-1027)
  double R[9], t0[3];
  const double xi[6] = {0, 0, 0, pose[0], pose[1], pose[2]};
  vo_se3_exp(xi, R, t0);
  Matrix3d Rm;
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) Rm(r, cc) = R[3 * r + cc];
  Scm = Sophus::Sim3(Sophus::ScSO3(scale, Rm), Vector3d(pose[3], pose[4], pose[5]));
  return inliers;
}

// Optimizer::solvePoseGraphLoop, reference optimizer_ceres.cpp:1036-1305: pose collection (:1062-1083), the four
// edge families (:1094-1236: loop connections, spanning tree, earlier loop edges, covisibility >= 100), the solve
// (:1238-1258), key-frame write-back (:1264-1278) and map-point re-anchoring (:1281-1301).
int Optimizer::solvePoseGraphLoop(Map *map_curr, KeyFrame *keyframe_match, KeyFrame *keyframe_curr,
                                  const LoopClosing::KeyFrameAndPose &uncorrectPose,
                                  const LoopClosing::KeyFrameAndPose &correctPose,
                                  const map<KeyFrame *, set<KeyFrame *>> &loopConnections, const bool &fixScaleFlag) {
  vector<KeyFrame *> allKeyFrames = map_curr->getAllKeyFrames();
  vector<MapPoint *> allMapPoints = map_curr->getAllMapPoints();
  const unsigned long maxKFId = map_curr->maxKFId_;
  const size_t n_ids = maxKFId + 1;
  vector<Sophus::Sim3, Eigen::aligned_allocator<Sophus::Sim3>> Scw(n_ids);
  std::vector<double> quats(4 * n_ids, 0.0), trans(3 * n_ids, 0.0), scales(n_ids, 1.0), q_meas, t_meas, s_meas;
  std::vector<int32_t> e_i, e_j;
  for (size_t i = 0; i < n_ids; i++) quats[4 * i + 3] = 1.0;  // ids without a key-frame: identity, no edge touches them
  for (KeyFrame *kf : allKeyFrames) {  // :1062-1083
    const unsigned long idx = kf->id_;
    LoopClosing::KeyFrameAndPose::const_iterator it = correctPose.find(kf);
    if (it != correctPose.end()) {
      Scw[idx] = it->second;
    } else {
      SE3 Tiw = kf->getPose();
      Scw[idx] = Sophus::Sim3(Sophus::ScSO3(Tiw.unit_quaternion()), Tiw.translation());
    }
    const Quaterniond q = Scw[idx].quaternion().normalized();
    memcpy(&quats[4 * idx], q.coeffs().data(), 32);  // Eigen coefficient order x, y, z, w
    const Vector3d t = Scw[idx].translation();
    memcpy(&trans[3 * idx], t.data(), 24);
    scales[idx] = Scw[idx].scale();
  }
  auto add_edge = [&](unsigned long id1, unsigned long id2, const Sophus::Sim3 &Sji) {  // one AddResidualBlock
    const Quaterniond q = Sji.quaternion().normalized();
    const Vector3d t = Sji.translation();
    e_i.push_back((int32_t)id1), e_j.push_back((int32_t)id2);
    q_meas.insert(q_meas.end(), q.coeffs().data(), q.coeffs().data() + 4);
    t_meas.insert(t_meas.end(), t.data(), t.data() + 3);
    s_meas.push_back(Sji.scale());
  };
  auto pose_or = [&](const LoopClosing::KeyFrameAndPose &m, KeyFrame *kf) -> Sophus::Sim3 {
    LoopClosing::KeyFrameAndPose::const_iterator it = m.find(kf);
    return it != m.end() ? it->second : Scw[kf->id_];
  };
  const int minFeat = 100;
  set<pair<unsigned long, unsigned long>> insertedLoopEdges;
  for (auto it = loopConnections.begin(); it != loopConnections.end(); it++) {  // :1094-1125, corrected poses
    KeyFrame *kf = it->first;
    const unsigned long id1 = kf->id_;
    const Sophus::Sim3 Swi = Scw[id1].inverse();
    for (KeyFrame *kc : it->second) {
      const unsigned long id2 = kc->id_;
      // Q-B5: `id2 != curr || id2 != match` is always true, so only the weight gate acts (:1105)
      if ((id2 != keyframe_curr->id_ || id2 != keyframe_match->id_) && kf->getWeight(kc) < minFeat) continue;
      add_edge(id1, id2, Scw[id2] * Swi);
      insertedLoopEdges.insert(make_pair(min(id1, id2), max(id1, id2)));
    }
  }
  for (KeyFrame *kf : allKeyFrames) {  // :1128-1236, uncorrected poses
    const unsigned long id1 = kf->id_;
    const Sophus::Sim3 Swi = pose_or(uncorrectPose, kf).inverse();
    KeyFrame *parentKF = kf->getParent();
    if (parentKF) add_edge(id1, parentKF->id_, pose_or(uncorrectPose, parentKF) * Swi);  // spanning tree :1141-1166
    const set<KeyFrame *> loopEdges = kf->loopEdges_;
    for (KeyFrame *kfl : loopEdges)  // loop edges found before this closure :1168-1198
      if (kfl->id_ < keyframe_curr->id_) add_edge(id1, kfl->id_, pose_or(uncorrectPose, kfl) * Swi);
    for (KeyFrame *kfw : kf->getCovisiblesByWeight(minFeat)) {  // covisibility >= 100 :1200-1236
      if (!kfw || kfw == parentKF || kf->children_.count(kfw) || loopEdges.count(kfw)) continue;
      if (kfw->isBad() || !(kfw->id_ < kf->id_)) continue;
      if (insertedLoopEdges.count(make_pair(kfw->id_, kf->id_))) continue;
      add_edge(id1, kfw->id_, pose_or(uncorrectPose, kfw) * Swi);
    }
  }
  // :1238-1258: key-frame `match` constant, quaternion parameterisation, scales constant, <= 20 iterations
  const int pg_rc = vo_pose_graph_solve((int)n_ids, quats.data(), trans.data(), scales.data(), (int)keyframe_match->id_,
                                        (int)e_i.size(), e_i.data(), e_j.data(), q_meas.data(), t_meas.data(), s_meas.data(),
                                        fixScaleFlag ? 1 : 0, 20, nullptr);
  if (pg_rc != VO_OK) {
    // The reference never inspects Ceres' summary (:1258), but a solver that did not run must not write anything back:
    // `quats` / `trans` still hold the uncorrected poses, and re-anchoring the map points through them would apply the
    // loop correction to the points and not to the key-frames.  Leave the map as it is and say why.
    std::cerr << "solvePoseGraphLoop: vo_pose_graph_solve failed (" << pg_rc << "): " << vo_last_error()
              << " -- loop closure not applied" << std::endl;
    return 0;
  }
  {
    unique_lock<mutex> lock(map_curr->mutexMapUpdate_);
    std::vector<double> S_rw(8 * n_ids, 0.0), S_wr(8 * n_ids, 0.0);  // Sim3 as quaternion (x y z w), translation, scale
    auto pack = [](const Sophus::Sim3 &S, double *o) {
      const Quaterniond q = S.quaternion().normalized();
      const Vector3d t = S.translation();
      memcpy(o, q.coeffs().data(), 32), memcpy(o + 4, t.data(), 24);
      o[7] = S.scale();
    };
    for (KeyFrame *kf : allKeyFrames) {  // :1264-1278
      const unsigned long id = kf->id_;
      const Quaterniond uq(quats[4 * id + 3], quats[4 * id], quats[4 * id + 1], quats[4 * id + 2]);  // (w, x, y, z)
      const Vector3d t(trans[3 * id], trans[3 * id + 1], trans[3 * id + 2]);
      const double s = scales[id];
      SE3 Tiw(uq, t / s);
      kf->setPose(Tiw);
      const Sophus::Sim3 Siw(Sophus::ScSO3(s, Tiw.rotation_matrix()), t);
      pack(Scw[id], &S_rw[8 * id]);
      pack(Siw.inverse(), &S_wr[8 * id]);
    }
    const size_t np = allMapPoints.size();  // :1281-1301, all points in one call
    std::vector<double> pin(3 * np + 3), pout(3 * np + 3);
    std::vector<int32_t> ref(np + 1, -1);
    for (size_t i = 0; i < np; i++) {
      MapPoint *mp = allMapPoints[i];
      if (mp->isBad()) continue;
      ref[i] = (int32_t)(mp->loopCorrectByKF_ == keyframe_curr->id_ ? mp->correctReference_ : mp->keyFrame_ref_->id_);
      const Vector3d p = mp->getPose();
      pin[3 * i] = p[0], pin[3 * i + 1] = p[1], pin[3 * i + 2] = p[2];
    }
    if (np > 0 && vo_sim3_reanchor_points((int)np, pin.data(), ref.data(), (int)n_ids, S_rw.data(), S_wr.data(),
                                          pout.data()) == VO_OK)
      for (size_t i = 0; i < np; i++)
        if (ref[i] >= 0) {
          allMapPoints[i]->setPose(Vector3d(pout[3 * i], pout[3 * i + 1], pout[3 * i + 2]));
          allMapPoints[i]->updateNormalAndDepth();
        }
  }
  return 0;  // Q-B6: the reference falls off the end of an int function; its callers ignore the value
}

}  // namespace myslam
