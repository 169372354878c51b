// include/myslam_shim/optimizer_hip.inl -- replacement bodies for Optimizer::solvePoseOnlySE3 and
// Optimizer::solveLocalBAPoseAndPoint (reference src/optimizer_ceres.cpp:157-314, 446-808).
// #include at the bottom of a copy of optimizer_ceres.cpp from which those two functions were
// removed (the Sim3 / pose-graph functions keep using Ceres).  Only the pointer-graph gather and the
// write-back stay on the host; the Ceres solves, the chi2 tests and the LM loop run on the GPU.
//
// Needs the reference's headers (Frame, KeyFrame, MapPoint, Map, Sophus): compile inside the
// reference tree.
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
  vo_ba *h = nullptr;
  if (vo_ba_create(&h, (int)cams.size(), poses.data(), fixed.data(), (int)points.size(), pts.data(),
                   (int)edges.size(), e_cam.data(), e_pt.data(), e_obs.data(), e_is.data(), cam) != VO_OK)
    return;
  // ---- solve (problem 1, chi2, problem 2, chi2) with the reference's stopFlag polling points
  // The reference polls `stopFlag` (a bool written by the tracking thread) at :594 and :612.  The
  // device schedule lasts about a millisecond, so both polls collapse into this one; the C-ABI takes
  // an int flag for callers that want the second poll as well.
  std::vector<uint8_t> erase(edges.size() + 1, 0);
  volatile int stop = stopFlag ? 1 : 0;
  const int rc = vo_ba_local_ba(h, &stop, erase.data(), nullptr);
  if (rc != VO_OK) {  // VO_ERR_STOPPED mirrors the early return at :594-595 (no write-back)
    vo_ba_destroy(h);
    return;
  }
  vo_ba_get_state(h, poses.data(), pts.data());
  vo_ba_destroy(h);
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
    cams[k]->setPose(SE3::exp(se3));
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
  ceres::RotationMatrixToAngleAxis(Rcm.data(), pose);  // any log map of SO(3) will do here
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
  vo_sim3_solve(1, offsets, cam_m.data(), pix_c.data(), is_c.data(), cam_c.data(), pix_m.data(), is_m.data(), camera,
                fixScaleFlag ? 1 : 0, pose, &scale, outlier.data(), &inliers, nullptr);
  for (size_t k = 0; k < index.size(); k++)
    if (outlier[k]) inlierMappoints[index[k]] = static_cast<MapPoint *>(nullptr);
  if (inliers == 0 && (int)index.size() - (int)std::count(outlier.begin(), outlier.end(), 1) < 10) return 0;  // :950-951
  double R[9];
  ceres::AngleAxisToRotationMatrix(pose, R);
  Scm = Sophus::Sim3(Sophus::ScSO3(scale, Eigen::Map<const Matrix3d>(R)), Eigen::Map<const Vector3d>(pose + 3));
  return inliers;
}

// The solve and the map-point re-anchoring inside Optimizer::solvePoseGraphLoop, reference
// optimizer_ceres.cpp:1036-1305.  The edge construction (:1094-1236) is the reference's own code with
// `problem.AddResidualBlock(...)` replaced by `pg.add(id1, id2, Sji)`; only what changes is shown.
struct PoseGraphArrays {
  std::vector<double> quats, trans, scales, q_meas, t_meas, s_meas;
  std::vector<int32_t> e_i, e_j;
  explicit PoseGraphArrays(size_t n_ids) : quats(4 * n_ids, 0.0), trans(3 * n_ids, 0.0), scales(n_ids, 1.0) {
    for (size_t i = 0; i < n_ids; i++) quats[4 * i + 3] = 1.0;
  }
  void set_node(unsigned long id, const Sophus::Sim3 &S) {  // :1074-1076
    const Eigen::Quaterniond q = S.quaternion().normalized();
    memcpy(&quats[4 * id], q.coeffs().data(), 32);          // x, y, z, w
    memcpy(&trans[3 * id], S.translation().data(), 24);
    scales[id] = S.scale();
  }
  void add(unsigned long id1, unsigned long id2, const Sophus::Sim3 &Sji) {
    const Eigen::Quaterniond q = Sji.quaternion().normalized();
    e_i.push_back((int32_t)id1), e_j.push_back((int32_t)id2);
    q_meas.insert(q_meas.end(), q.coeffs().data(), q.coeffs().data() + 4);
    t_meas.insert(t_meas.end(), Sji.translation().data(), Sji.translation().data() + 3);
    s_meas.push_back(Sji.scale());
  }
  int solve(unsigned long fixed_id, bool fixScaleFlag) {  // :1238-1258
    return vo_pose_graph_solve((int)scales.size(), quats.data(), trans.data(), scales.data(), (int)fixed_id,
                               (int)e_i.size(), e_i.data(), e_j.data(), q_meas.data(), t_meas.data(), s_meas.data(),
                               fixScaleFlag ? 1 : 0, 20, nullptr);
  }
};
// After pg.solve(keyframe_match->id_, fixScaleFlag): key-frame poses are written as at :1264-1278
// (SE3(uq, t / s)), the Sim3 pairs (Scw[idm], optimizedSwc[idm]) are packed 8 doubles each and
// vo_sim3_reanchor_points() produces every corrected map-point position of :1281-1301 in one call
// (ref = correctReference_ or keyFrame_ref_->id_, -1 for bad points).

}  // namespace myslam
