// include/myslam_shim/matcher_hip.inl -- replacement bodies for the hot members of myslam::Matcher
// (reference src/matcher.cpp).  #include this file at the bottom of a copy of matcher.cpp from which
// computeDistance (:1240-1256) and the two projection searches (:18-148, :274-353) were removed; the
// remaining members (BoW / Sim3 / fuse) keep calling Matcher::computeDistance, which now goes to the
// device for batches and stays scalar for single pairs.
//
// Needs the reference's headers (Frame, MapPoint, Camera): compile inside the reference tree.
#include <vector>

#include "vo_hip.h"

namespace myslam {

// gather a Frame into the flat view the C-ABI takes (frame.h:26-45)
struct FrameFlat {
  std::vector<float> x, y, angle;
  std::vector<int32_t> octave;
  vo_frame_view view;
  explicit FrameFlat(Frame *f) {
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

int Matcher::computeDistance(const Mat &a, const Mat &b) {
  uint16_t d = 0;
  vo_hamming_matrix(a.ptr<uint8_t>(), 1, b.ptr<uint8_t>(), 1, &d);
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

}  // namespace myslam
