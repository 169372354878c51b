// include/myslam_shim/sim3solver_hip.inl -- replacement body for Sim3Solver::iterate (reference
// src/sim3Solver.cpp:98-177).  #include at the bottom of a copy of sim3Solver.cpp from which iterate, computeSim3
// and checkInliers were removed (the constructor, setRansacParameters and randomInt stay: they fill pcams*_, pixels*_,
// maxError*_ and draw the samples).
//
// The reference evaluates one hypothesis per loop trip: three rand() draws without replacement (:119-137), Horn's
// closed form (:179-252), inlier test of all correspondences (:254-280), early return at the first hypothesis with more
// than ransacInlierThreshold_ inliers (:141-160).  The shim keeps that trip structure -- draw three, evaluate, book-keep,
// possibly return -- so that rand() is advanced by EXACTLY the reference's number of calls: loopClosing.cpp:233-275 keeps
// iterating the other candidates' solvers (and this one again) after an early return, and their samples depend on the
// generator's state (round 3 drew all triplets of a call first, i.e. up to twelve draws too many behind an
// early success).  Each trip is one vo_sim3_ransac_eval call with one hypothesis (Horn + the inlier test of all
// correspondences on the device); a caller that does not share rand() with anyone may define VO_SIM3_DRAW_AHEAD to get
// the one-launch-per-call form back (same results for the trips the reference runs, rand() advanced for the rest).
#include <vector>

#include "vo_hip.h"

namespace myslam {

Sophus::Sim3 Sim3Solver::iterate(int iterations_req, bool &stopFlag, bool &emptyFlag, vector<bool> &inlierFlags,
                                 int &inliers_cnt) {
  stopFlag = false;
  emptyFlag = false;
  inlierFlags = vector<bool>(matches_cnt_, false);
  inliers_cnt = 0;
  if ((int)mappoints1_.size() < ransacInlierThreshold_) {  // :106-112
    stopFlag = true;
    emptyFlag = true;
    return Sophus::Sim3();
  }
  const int n = (int)mappoints1_.size();
  const int trips = std::max(0, std::min(ransacMaxIters_ - iterations_global_, iterations_req));
  std::vector<double> pc1((size_t)3 * n), pc2((size_t)3 * n), px1((size_t)2 * n), px2((size_t)2 * n);
  std::vector<int32_t> me1(maxError1_.begin(), maxError1_.end()), me2(maxError2_.begin(), maxError2_.end());
  for (int i = 0; i < n; i++) {
    for (int r = 0; r < 3; r++) pc1[3 * i + r] = pcams1_[i][r], pc2[3 * i + r] = pcams2_[i][r];
    for (int r = 0; r < 2; r++) px1[2 * i + r] = pixels1_[i][r], px2[2 * i + r] = pixels2_[i][r];
  }
  Camera *camera = keyframe1_->camera_;
  const float cam4[4] = {camera->fx_, camera->fy_, camera->cx_, camera->cy_};
  auto draw = [&](int32_t *tri) {  // :117-137, the sampling only
    vector<int> availableIdxs = idxForRandom_;
    for (int i = 0; i < 3; ++i) {
      const int randi = randomInt(0, (int)availableIdxs.size() - 1);
      tri[i] = availableIdxs[randi];
      availableIdxs[randi] = availableIdxs.back();
      availableIdxs.pop_back();
    }
  };
#ifdef VO_SIM3_DRAW_AHEAD
  const int chunk = std::max(trips, 1);
#else
  const int chunk = 1;
#endif
  std::vector<int32_t> triplets((size_t)3 * chunk), counts(chunk);
  std::vector<uint8_t> flags((size_t)chunk * n);
  std::vector<double> sims((size_t)13 * chunk);
  for (int k0 = 0; k0 < trips; k0 += chunk) {
    const int nk = std::min(chunk, trips - k0);
    for (int k = 0; k < nk; k++) draw(&triplets[3 * k]);
    // the correspondences go to the device with the first trip of this call and stay there for the others (NULL arrays)
    const bool first = k0 == 0;
    if (vo_sim3_ransac_eval(n, first ? pc1.data() : nullptr, first ? pc2.data() : nullptr, first ? px1.data() : nullptr,
                            first ? px2.data() : nullptr, first ? me1.data() : nullptr, first ? me2.data() : nullptr, cam4, nk,
                            triplets.data(), fixScale_ ? 1 : 0, counts.data(), flags.data(), sims.data()) != VO_OK) {
      stopFlag = true;  // no error channel in the reference: report "nothing found, stop"
      emptyFlag = true;
      return Sophus::Sim3();
    }
    for (int k = 0; k < nk; k++) {  // :138-160 on the evaluated hypotheses
      iterations_global_++;
      inliers_cnt_ = counts[k];
      if (inliers_cnt_ >= inliers_best_) {
        inliers_best_ = inliers_cnt_;
        inlierFlags_.assign(n, false);
        for (int i = 0; i < n; i++) inlierFlags_[i] = flags[(size_t)k * n + i] != 0;
        inlierFlags_best_ = inlierFlags_;
        const double *S = &sims[13 * k];
        Matrix3d R;
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) R(r, c) = S[3 * r + c];
        R12_ = R, t12_ = Vector3d(S[9], S[10], S[11]), s12_ = S[12];
        T12_ = Sophus::Sim3(Sophus::ScSO3(s12_, R12_), t12_);  // :240-241
        T12_best_ = T12_, R12_best_ = R12_, t12_best_ = t12_, s12_best_ = s12_;
        if (inliers_cnt_ > ransacInlierThreshold_) {
          inliers_cnt = inliers_cnt_;
          for (int i = 0; i < n; i++)
            if (inlierFlags_[i]) inlierFlags[matchedIndexs_[i]] = true;
          return T12_best_;
        }
      }
    }
  }
  if (iterations_global_ >= ransacMaxIters_) stopFlag = true;
  emptyFlag = true;
  return Sophus::Sim3();
}

}  // namespace myslam
