// include/myslam_shim/matcher_hip.inl -- replacement bodies for the hot members of myslam::Matcher
// (reference src/matcher.cpp).  #include this file at the bottom of a copy of matcher.cpp from which
// computeDistance (:1240-1256), the projection searches (:18-148, :150-272, :274-353), the two
// searchByBoW overloads (:449-559, :561-677), searchForTriangulation (:867-1010), fuseMapPoints
// (:1012-1133) and the loop-closure members (:356-447, :679-865, :1135-1238) were removed, i.e. every
// member of Matcher except the constructor and computeThreeMax.
//
// Needs the reference's headers (Frame, MapPoint, Camera): compile inside the reference tree.
#include <cstring>
#include <vector>

#include "vo_hip.h"

namespace myslam {

// gather a Frame into the flat view the C-ABI takes (frame.h:26-45)
struct FrameFlat {
  std::vector<float> x, y, angle;
  std::vector<int32_t> octave;
  vo_frame_view view;
  template <class F>  // Frame and KeyFrame expose the same members (frame.h:26-45, keyframe.h)
  explicit FrameFlat(F *f) {
    const int n = (int)f->unKeypoints_.size();
    x.resize(n), y.resize(n), angle.resize(n), octave.resize(n);
    for (int i = 0; i < n; i++) {
      const cv::KeyPoint &k = f->unKeypoints_[i];
      x[i] = k.pt.x, y[i] = k.pt.y, angle[i] = k.angle, octave[i] = k.octave;
    }
    view.n = n;
    view.x = x.data(), view.y = y.data(), view.octave = octave.data(), view.angle = angle.data();
    view.uright = f->uRight_.data();
    view.desc = f->descriptors_.data;  // N x 32 CV_8U, continuous (ORBextractor output)
    view.xmin = f->xMin_, view.ymin = f->yMin_, view.xmax = f->xMax_, view.ymax = f->yMax_;
  }
};

// Matcher::computeDistance (matcher.cpp:1240-1256): ONE pair is eight host popcounts -- a GPU round trip per
// 256-bit distance would turn microseconds into milliseconds.  Its only N x N caller, MapPoint::computeDescriptor,
// has its own shim (mappoint_hip.inl: one kernel launch per map point or per batch of map points).
int Matcher::computeDistance(const Mat &a, const Mat &b) {
  const uint8_t *pa = a.ptr<uint8_t>(), *pb = b.ptr<uint8_t>();
  int d = 0;
  for (int w = 0; w < 4; w++) {
    unsigned long long x, y;
    memcpy(&x, pa + 8 * w, 8), memcpy(&y, pb + 8 * w, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

// Matcher::searchByProjection(Frame*, Frame*, radius, checkRot), reference matcher.cpp:18-148
int Matcher::searchByProjection(Frame *cur, Frame *last, const float radius, bool checkRot) {
  Camera *cam = cur->camera_;
  SE3 Tcw = cur->Tcw_;
  SE3 Tlc = last->Tcw_ * Tcw.inverse();
  const bool forward = static_cast<float>(Tlc.translation()[2]) > cam->b_;
  const bool backward = -static_cast<float>(Tlc.translation()[2]) > cam->b_;
  const int nq = (int)last->mappoints_.size();
  std::vector<uint8_t> flags(nq, 0), qdesc((size_t)nq * 32, 0);
  std::vector<float> u(nq), v(nq), invz(nq), ang(nq);
  std::vector<int32_t> oct(nq);
  std::vector<MapPoint *> mps(nq, nullptr);
  for (int i = 0; i < nq; i++) {
    MapPoint *mp = last->mappoints_[i];
    if (!mp || last->outliers_[i]) continue;
    Vector3d pc = Tcw * mp->getPose();
    const float z = static_cast<float>(pc[2]);
    if (z < 0.0f) continue;
    Vector2d px = cam->camera2pixel(pc);
    const float uu = px[0], vv = px[1];
    if (uu < (int)cur->xMin_ || uu > (int)cur->xMax_ || vv < (int)cur->yMin_ || vv > (int)cur->yMax_) continue;
    flags[i] = 1 | (mp->observe_cnt_ > 0 ? 2 : 0);
    u[i] = uu, v[i] = vv, invz[i] = 1.0f / z;
    oct[i] = last->unKeypoints_[i].octave, ang[i] = last->unKeypoints_[i].angle;
    memcpy(&qdesc[(size_t)i * 32], mp->getDescriptor().data, 32);
    mps[i] = mp;
  }
  FrameFlat ff(cur);
  const int nf = ff.view.n;
  std::vector<uint8_t> blocked(nf, 0);
  std::vector<int32_t> assigned(nf, -1);
  for (int k = 0; k < nf; k++) blocked[k] = cur->mappoints_[k] && cur->mappoints_[k]->observe_cnt_ > 0;
  int n = 0;
  vo_match_frame_projection(&ff.view, nq, flags.data(), u.data(), v.data(), invz.data(), oct.data(), ang.data(),
                            qdesc.data(), radius, cam->bf_, forward ? 1 : (backward ? 2 : 0), checkRot ? 1 : 0,
                            (int)cur->scaleFactors_.size(), cur->scaleFactors_.data(), blocked.data(),
                            assigned.data(), &n);
  for (int k = 0; k < nf; k++)
    if (assigned[k] >= 0) cur->mappoints_[k] = mps[assigned[k]];
  return n;
}

// Matcher::searchByProjection(Frame*, const vector<MapPoint*>&, thRadius), reference matcher.cpp:274-353
int Matcher::searchByProjection(Frame *frame, const vector<MapPoint *> &mappoints, const float thRadius) {
  const int nq = (int)mappoints.size();
  std::vector<uint8_t> flags(nq, 0), qdesc((size_t)nq * 32, 0);
  std::vector<float> u(nq), v(nq), ur(nq), vc(nq);
  std::vector<int32_t> lvl(nq);
  for (int i = 0; i < nq; i++) {
    MapPoint *mp = mappoints[i];
    if (mp->isBad() || !mp->trackInLocalMap_) continue;
    flags[i] = 1 | (mp->getObsCnt() > 0 ? 2 : 0);
    u[i] = mp->trackProj_u_, v[i] = mp->trackProj_v_, ur[i] = mp->trackProj_uR_;
    lvl[i] = mp->trackScaleLevel_, vc[i] = mp->viewCos_;
    memcpy(&qdesc[(size_t)i * 32], mp->getDescriptor().data, 32);
  }
  FrameFlat ff(frame);
  const int nf = ff.view.n;
  std::vector<uint8_t> blocked(nf, 0);
  std::vector<int32_t> assigned(nf, -1);
  for (int k = 0; k < nf; k++) blocked[k] = frame->mappoints_[k] && frame->mappoints_[k]->getObsCnt() > 0;
  int n = 0;
  vo_match_local_map(&ff.view, nq, flags.data(), u.data(), v.data(), ur.data(), lvl.data(), vc.data(), qdesc.data(),
                     thRadius, ratio_, frame->scaleFactors_.data(), blocked.data(), assigned.data(), &n);
  for (int k = 0; k < nf; k++)
    if (assigned[k] >= 0) frame->mappoints_[k] = mappoints[assigned[k]];
  return n;
}

// DBoW3::FeatureVector (std::map<NodeId, std::vector<unsigned>>) -> CSR view
struct BowFlat {
  std::vector<uint32_t> node, feat;
  std::vector<int32_t> start;
  vo_bow_view view;
  explicit BowFlat(const DBoW3::FeatureVector &fv) {
    start.push_back(0);
    for (const auto &kv : fv) {
      node.push_back(kv.first);
      feat.insert(feat.end(), kv.second.begin(), kv.second.end());
      start.push_back((int32_t)feat.size());
    }
    view.n_nodes = (int32_t)node.size();
    view.node_id = node.data(), view.start = start.data(), view.feat = feat.data();
  }
};

// Matcher::searchByProjection(Frame*, KeyFrame*, radius, distThreshold, found, checkRot), :150-272
int Matcher::searchByProjection(Frame *cur, KeyFrame *kf, const float radius, const float distThreshold,
                                const set<MapPoint *> &found, bool checkRot) {
  Camera *cam = cur->camera_;
  const int xMax = cur->xMax_, xMin = cur->xMin_, yMax = cur->yMax_, yMin = cur->yMin_;
  const SE3 Tcw = cur->Tcw_;
  const Vector3d Ow = Tcw.inverse().translation();
  const vector<MapPoint *> mps = kf->getMapPoints();
  const int nq = (int)mps.size();
  std::vector<uint8_t> flags(nq, 0), qdesc((size_t)nq * 32, 0);
  std::vector<float> u(nq), v(nq), ang(nq);
  std::vector<int32_t> lvl(nq);
  for (int i = 0; i < nq; i++) {
    MapPoint *mp = mps[i];
    if (!mp || mp->isBad() || found.count(mp)) continue;
    const Vector3d pc = Tcw * mp->getPose();
    if (static_cast<float>(pc[2]) <= 0) continue;
    const Vector2d px = cam->camera2pixel(pc);
    const float uu = px[0], vv = px[1];
    if (uu > xMax || uu < xMin || vv > yMax || vv < yMin) continue;
    const float dist = (mp->getPose() - Ow).norm();
    if (dist < mp->getMinDistanceThreshold() || dist > mp->getMaxDistanceThreshold()) continue;
    flags[i] = 1, u[i] = uu, v[i] = vv, lvl[i] = mp->predictScale(dist, cur);
    ang[i] = kf->unKeypoints_[i].angle;
    memcpy(&qdesc[(size_t)i * 32], mp->getDescriptor().data, 32);
  }
  FrameFlat ff(cur);
  std::vector<uint8_t> has(ff.view.n);
  std::vector<int32_t> assigned(ff.view.n, -1);
  for (int k = 0; k < ff.view.n; k++) has[k] = cur->mappoints_[k] != nullptr;
  int n = 0;
  vo_match_frame_keyframe(&ff.view, nq, flags.data(), u.data(), v.data(), lvl.data(), ang.data(), qdesc.data(), radius,
                          distThreshold, checkRot ? 1 : 0, kf->scaleFactors_.data(), has.data(), assigned.data(), &n);
  for (int k = 0; k < ff.view.n; k++)
    if (assigned[k] >= 0) cur->mappoints_[k] = mps[assigned[k]];
  return n;
}

// Matcher::searchByBoW(KeyFrame*, Frame*, matches, checkRot), :449-559
int Matcher::searchByBoW(KeyFrame *kf, Frame *frame, vector<MapPoint *> &matches, bool checkRot) {
  matches.assign(frame->N_, static_cast<MapPoint *>(nullptr));
  const vector<MapPoint *> mps = kf->getMapPoints();
  std::vector<uint8_t> valid(mps.size());
  for (size_t i = 0; i < mps.size(); i++) valid[i] = mps[i] && !mps[i]->isBad();
  FrameFlat a(kf), b(frame);
  BowFlat an(kf->featVec_), bn(frame->featVec_);
  std::vector<int32_t> m(frame->N_, -1);
  int n = 0;
  vo_match_bow(&a.view, valid.data(), &an.view, &b.view, nullptr, &bn.view, 0, ratio_, checkRot ? 1 : 0, m.data(), &n);
  for (int k = 0; k < (int)frame->N_; k++)
    if (m[k] >= 0) matches[k] = mps[m[k]];
  return n;
}

// Matcher::searchByBoW(KeyFrame*, KeyFrame*, matches, checkRot), :561-677
int Matcher::searchByBoW(KeyFrame *kf1, KeyFrame *kf2, vector<MapPoint *> &matches12, bool checkRot) {
  const vector<MapPoint *> mps1 = kf1->getMapPoints(), mps2 = kf2->getMapPoints();
  matches12.assign(mps1.size(), static_cast<MapPoint *>(nullptr));
  std::vector<uint8_t> v1(mps1.size()), v2(mps2.size());
  for (size_t i = 0; i < mps1.size(); i++) v1[i] = mps1[i] && !mps1[i]->isBad();
  for (size_t i = 0; i < mps2.size(); i++) v2[i] = mps2[i] && !mps2[i]->isBad();
  FrameFlat a(kf1), b(kf2);
  BowFlat an(kf1->featVec_), bn(kf2->featVec_);
  std::vector<int32_t> m(mps1.size(), -1);
  int n = 0;
  vo_match_bow(&a.view, v1.data(), &an.view, &b.view, v2.data(), &bn.view, 1, ratio_, checkRot ? 1 : 0, m.data(), &n);
  for (size_t k = 0; k < m.size(); k++)
    if (m[k] >= 0) matches12[k] = mps2[m[k]];
  return n;
}

// Matcher::searchForTriangulation(kf1, kf2, matchIdxs, F12, checkRot), :867-1010
int Matcher::searchForTriangulation(KeyFrame *kf1, KeyFrame *kf2, vector<pair<int, int>> &matchIdxs,
                                    Eigen::Matrix3d &F12, bool checkRot) {
  const vector<MapPoint *> mps1 = kf1->getMapPoints(), mps2 = kf2->getMapPoints();
  std::vector<uint8_t> h1(mps1.size()), h2(mps2.size());
  for (size_t i = 0; i < mps1.size(); i++) h1[i] = mps1[i] != nullptr;
  for (size_t i = 0; i < mps2.size(); i++) h2[i] = mps2[i] != nullptr;
  const Vector3d C2 = kf2->getPose() * kf1->getCamCenter();
  const Vector2d e = kf2->camera_->camera2pixel(C2);
  double F[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) F[3 * r + c] = F12(r, c);
  FrameFlat a(kf1), b(kf2);
  BowFlat an(kf1->featVec_), bn(kf2->featVec_);
  std::vector<int32_t> m(kf1->N_, -1);
  int n = 0;
  vo_match_triangulation(&a.view, h1.data(), &an.view, &b.view, h2.data(), &bn.view, F, (float)e[0], (float)e[1],
                         kf2->scaleFactors_.data(), checkRot ? 1 : 0, m.data(), &n);
  matchIdxs.clear();
  matchIdxs.reserve(n);
  for (int i = 0; i < (int)kf1->N_; i++)
    if (m[i] >= 0) matchIdxs.push_back(make_pair(i, m[i]));
  return n;
}

// Matcher::fuseMapPoints(KeyFrame*, mappoints, threshold), :1012-1133.  Matching is a pure function of the
// projections and descriptors, so it runs as one batch; the map mutation (:1108-1127) is replayed in
// list order with the two gates an earlier merge can flip (isBad, beObserved) re-evaluated.
int Matcher::fuseMapPoints(KeyFrame *kf, vector<MapPoint *> &mappoints, const float &threshold) {
  Camera *cam = kf->camera_;
  const SE3 Tcw = kf->getPose();
  const Vector3d Ow = kf->getCamCenter();
  const int nq = (int)mappoints.size();
  std::vector<uint8_t> flags(nq, 0), qdesc((size_t)nq * 32, 0);
  std::vector<float> u(nq), v(nq), ur(nq);
  std::vector<int32_t> lvl(nq);
  for (int i = 0; i < nq; i++) {
    MapPoint *mp = mappoints[i];
    if (!mp || mp->isBad() || mp->beObserved(kf)) continue;
    const Vector3d pw = mp->getPose(), pc = Tcw * pw;
    const float z = static_cast<float>(pc[2]);
    if (z < 0.0f) continue;
    const float invz = 1.0f / z;
    const float uu = cam->fx_ * (static_cast<float>(pc[0]) * invz) + cam->cx_;
    const float vv = cam->fy_ * (static_cast<float>(pc[1]) * invz) + cam->cy_;
    if (!kf->isInImg(uu, vv)) continue;
    const Vector3d line = pw - Ow;
    const float dist = line.norm();
    if (dist < mp->getMinDistanceThreshold() || dist > mp->getMaxDistanceThreshold()) continue;
    if (line.dot(mp->getNormalVector()) < 0.5 * dist) continue;
    flags[i] = 1, u[i] = uu, v[i] = vv, ur[i] = uu - cam->bf_ * invz, lvl[i] = mp->predictScale(dist, kf);
    memcpy(&qdesc[(size_t)i * 32], mp->getDescriptor().data, 32);
  }
  FrameFlat ff(kf);
  std::vector<int32_t> best(nq, -1);
  int n = 0, cnt = 0;
  vo_match_fuse(&ff.view, nq, flags.data(), u.data(), v.data(), ur.data(), lvl.data(), qdesc.data(), threshold,
                kf->scaleFactors_.data(), best.data(), &n);
  for (int i = 0; i < nq; i++) {
    if (best[i] < 0) continue;
    MapPoint *mp = mappoints[i];
    if (mp->isBad() || mp->beObserved(kf)) continue;
    MapPoint *org = kf->mappoints_[best[i]];
    if (org) {
      if (!org->isBad()) {
        if (org->getObsCnt() > mp->getObsCnt())
          mp->replaceMapPoint(org);
        else
          org->replaceMapPoint(mp);
      }
    } else {
      mp->addObservation(kf, best[i]);
      kf->addMapPoint(mp, best[i]);
    }
    cnt++;
  }
  return cnt;
}

// projection of a map point under a similarity / rigid pose with the gates shared by the loop-closure
// members (:380-406, :1163-1187): z >= 0, inside the image, distance range, viewing angle
struct LoopQuery {
  std::vector<uint8_t> flags, desc;
  std::vector<float> u, v;
  std::vector<int32_t> level;
  explicit LoopQuery(int n) : flags(n, 0), desc((size_t)n * 32, 0), u(n), v(n), level(n) {}
  void set(int i, MapPoint *mp, float uu, float vv, int lvl) {
    flags[i] = 1, u[i] = uu, v[i] = vv, level[i] = lvl;
    memcpy(&desc[(size_t)i * 32], mp->getDescriptor().data, 32);
  }
};

// Matcher::searchByProjection(KeyFrame*, Sim3&, loopMapPoints, matchMapPoints, th), :356-447
int Matcher::searchByProjection(KeyFrame *kf, Sophus::Sim3 &Scw, vector<MapPoint *> &loopMapPoints,
                                vector<MapPoint *> &matchMapPoints, int th) {
  Camera *cam = kf->camera_;
  const double scale = Scw.scale();
  const Eigen::Matrix3d Rcw = Scw.rotation_matrix() / scale;
  const Vector3d tcw = Scw.translation() / scale, Ow = -Rcw.transpose() * tcw;
  set<MapPoint *> found(matchMapPoints.begin(), matchMapPoints.end());
  found.erase(static_cast<MapPoint *>(nullptr));
  const int nq = (int)loopMapPoints.size();
  LoopQuery q(nq);
  for (int i = 0; i < nq; i++) {
    MapPoint *mp = loopMapPoints[i];
    if (!mp || mp->isBad() || found.count(mp)) continue;
    const Vector3d pc = Rcw * mp->getPose() + tcw;
    const float z = static_cast<float>(pc[2]);
    if (z < 0) continue;
    const float invz = 1.0f / z;
    const float uu = cam->fx_ * (static_cast<float>(pc[0]) * invz) + cam->cx_;
    const float vv = cam->fy_ * (static_cast<float>(pc[1]) * invz) + cam->cy_;
    if (!kf->isInImg(uu, vv)) continue;
    const Vector3d line = mp->getPose() - Ow;
    const float dist = line.norm();
    if (dist < mp->getMinDistanceThreshold() || dist > mp->getMaxDistanceThreshold()) continue;
    if (line.dot(mp->getNormalVector()) < 0.5 * dist) continue;
    q.set(i, mp, uu, vv, mp->predictScale(dist, kf));
  }
  FrameFlat ff(kf);
  std::vector<uint8_t> occupied(ff.view.n);
  for (int k = 0; k < ff.view.n; k++) occupied[k] = matchMapPoints[k] != nullptr;
  std::vector<int32_t> assigned(ff.view.n, -1);
  int n = 0;
  vo_match_sim3_projection(&ff.view, nq, q.flags.data(), q.u.data(), q.v.data(), q.level.data(), q.desc.data(), th,
                           kf->scaleFactors_.data(), occupied.data(), assigned.data(), &n);
  for (int k = 0; k < ff.view.n; k++)
    if (assigned[k] >= 0) matchMapPoints[k] = loopMapPoints[assigned[k]];
  return n;
}

// Matcher::searchBySim3, :679-865
int Matcher::searchBySim3(KeyFrame *kf1, KeyFrame *kf2, vector<MapPoint *> &matches12, Sophus::Sim3 &S12,
                          const float th) {
  Camera *cam = kf1->camera_;
  const vector<MapPoint *> mps1 = kf1->getMapPoints(), mps2 = kf2->getMapPoints();
  const int N1 = (int)mps1.size(), N2 = (int)mps2.size();
  std::vector<bool> matched1(N1, false), matched2(N2, false);
  const SE3 Tcw1 = kf1->getPose(), Tcw2 = kf2->getPose();
  const Sophus::Sim3 S21 = S12.inverse();
  for (int i = 0; i < N1; i++)
    if (MapPoint *mp = matches12[i]) {
      matched1[i] = true;
      const int idx2 = mp->getIndexInKeyFrame(kf2);
      if (idx2 >= 0 && idx2 < N2) matched2[idx2] = true;
    }
  LoopQuery q1(N1), q2(N2);
  auto project = [&](const Vector3d &p, KeyFrame *into, bool strict, float &uu, float &vv, float &dist) {
    const float z = static_cast<float>(p[2]);
    if (strict ? z <= 0 : z < 0) return false;  // :729 uses `<`, :797 uses `<=`
    const float invz = 1.0f / z;
    uu = cam->fx_ * (static_cast<float>(p[0]) * invz) + cam->cx_;
    vv = cam->fy_ * (static_cast<float>(p[1]) * invz) + cam->cy_;
    dist = p.norm();
    return into->isInImg(uu, vv);
  };
  for (int i = 0; i < N1; i++) {
    MapPoint *mp = mps1[i];
    if (!mp || matched1[i] || mp->isBad()) continue;
    float uu, vv, d;
    if (!project(S21 * (Tcw1 * mp->pos_), kf2, false, uu, vv, d)) continue;
    if (d < mp->getMinDistanceThreshold() || d > mp->getMaxDistanceThreshold()) continue;
    q1.set(i, mp, uu, vv, mp->predictScale(d, kf2));
  }
  for (int j = 0; j < N2; j++) {
    MapPoint *mp = mps2[j];
    if (!mp || matched2[j]) continue;
    float uu, vv, d;
    if (!project(S12 * (Tcw2 * mp->pos_), kf1, true, uu, vv, d)) continue;
    if (d < mp->getMinDistanceThreshold() || d > mp->getMaxDistanceThreshold()) continue;
    q2.set(j, mp, uu, vv, mp->predictScale(d, kf1));
  }
  FrameFlat f1(kf1), f2(kf2);
  std::vector<int32_t> m12(N1, -1);
  int n = 0;
  vo_match_sim3_mutual(&f1.view, &f2.view, q1.flags.data(), q1.u.data(), q1.v.data(), q1.level.data(), q1.desc.data(),
                       q2.flags.data(), q2.u.data(), q2.v.data(), q2.level.data(), q2.desc.data(), th,
                       kf1->scaleFactors_.data(), kf2->scaleFactors_.data(), m12.data(), &n);
  for (int i = 0; i < N1; i++)
    if (m12[i] >= 0) matches12[i] = mps2[m12[i]];
  return n;
}

// Matcher::fuseByPose, :1135-1238 (matching as one batch, mutation :1215-1230 replayed in list order)
int Matcher::fuseByPose(KeyFrame *kf, Sophus::Sim3 &Scw, vector<MapPoint *> &loopMapPoints,
                        vector<MapPoint *> &replaceMapPoints, const float th) {
  Camera *cam = kf->camera_;
  const SE3 Tcw(Scw.rotation_matrix(), Scw.translation());
  const Vector3d Ow = -Tcw.rotation_matrix().transpose() * Tcw.translation();
  set<MapPoint *> found;
  for (MapPoint *mp : kf->mappoints_)
    if (mp && !mp->isBad()) found.insert(mp);
  const int nq = (int)loopMapPoints.size();
  LoopQuery q(nq);
  for (int i = 0; i < nq; i++) {
    MapPoint *mp = loopMapPoints[i];
    if (!mp || mp->isBad() || found.count(mp)) continue;
    const Vector3d pc = Tcw * mp->getPose();
    const float z = static_cast<float>(pc[2]);
    if (z < 0) continue;
    const float invz = 1.0f / z;
    const float uu = cam->fx_ * (static_cast<float>(pc[0]) * invz) + cam->cx_;
    const float vv = cam->fy_ * (static_cast<float>(pc[1]) * invz) + cam->cy_;
    if (!kf->isInImg(uu, vv)) continue;
    const Vector3d line = mp->getPose() - Ow;
    const float dist = line.norm();
    if (dist < mp->getMinDistanceThreshold() || dist > mp->getMaxDistanceThreshold()) continue;
    if (line.dot(mp->getNormalVector()) < 0.5 * dist) continue;
    q.set(i, mp, uu, vv, mp->predictScale(dist, kf));
  }
  FrameFlat ff(kf);
  std::vector<int32_t> best(nq, -1);
  int n = 0, fused = 0;
  vo_match_area_best(&ff.view, nq, q.flags.data(), q.u.data(), q.v.data(), q.level.data(), q.desc.data(), th,
                     kf->scaleFactors_.data(), 50, best.data(), &n);
  for (int i = 0; i < nq; i++) {
    if (best[i] < 0) continue;
    MapPoint *mp = loopMapPoints[i], *mpKF = kf->mappoints_[best[i]];
    if (mpKF) {
      if (!mpKF->isBad()) replaceMapPoints[i] = mpKF;
    } else {
      mp->addObservation(kf, best[i]);
      kf->addMapPoint(mp, best[i]);
    }
    fused++;
  }
  return fused;
}

// Frame::computeBow / KeyFrame::computeBow (frame.cpp:248-253, keyframe.cpp:394-398):
// voc_->transform(descriptors, bowVec_, featVec_, 3).  The tree descent runs on the device; filling the
// two std::maps is what DBoW3::Vocabulary::transform does after it (word weights summed per word, L1
// normalisation for the TF-IDF / L1 vocabulary shipped with ORB-SLAM2, features listed per node).
// `voc_dev` is created once from the parsed vocabulary with vo_vocab_create.
template <class F>
void computeBowHip(F *f, const vo_vocab *voc_dev) {
  if (!f->featVec_.empty() && !f->bowVec_.empty()) return;
  const int n = f->descriptors_.rows;
  std::vector<int32_t> word(n), node(n);
  std::vector<double> weight(n);
  vo_bow_transform(voc_dev, n, f->descriptors_.data, 3, word.data(), weight.data(), node.data());
  f->bowVec_.clear(), f->featVec_.clear();
  for (int i = 0; i < n; i++)
    if (weight[i] > 0) {
      f->bowVec_.addWeight(word[i], weight[i]);
      f->featVec_.addFeature(node[i], i);
    }
  f->bowVec_.normalize(DBoW3::L1);
}

}  // namespace myslam
