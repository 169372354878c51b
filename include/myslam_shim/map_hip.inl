// include/myslam_shim/map_hip.inl -- Map::score (reference src/map.cpp:335-376, the L1 similarity of two DBoW3 BoW
// vectors) for ALL candidates of Map::detectRelocalizationCandidates (:138-151) or Map::detectLoopCandidates (:262-275) in
// one launch (vo_bow_score) instead of one sorted-map merge per candidate on the host.  #include near the top of map.cpp;
// the two candidate loops change like this:
//
//   before `for (auto it = sharingWordKFs.begin(); ...)` (:138 / :262):
//       std::vector<KeyFrame *> scored_kfs;
//       for (KeyFrame *kf : sharingWordKFs) if (kf->relocateWordCnt_ > minCommonWords) scored_kfs.push_back(kf);   // loopWordCnt_ at :266
//       const std::vector<double> scores = vo_shim::scoreCandidates(frame->bowVec_, scored_kfs);
//       size_t next = 0;
//   the line `float sc = score(frame->bowVec_, kf->bowVec_);` (:145 / :268) becomes:
//       float sc = (float)scores[next++];
//
// Map::score itself stays as it is for any other caller.  The device sums the common words in ascending word order, as
// the host merge does: the doubles agree to the last bit except for the association of the final -score / 2 (tests).
#include <vector>

#include "vo_hip.h"

namespace myslam {
namespace vo_shim {

inline std::vector<double> scoreCandidates(const DBoW3::BowVector &query, const std::vector<KeyFrame *> &candidates) {
  std::vector<double> scores(candidates.size(), 0.0);
  if (candidates.empty()) return scores;
  std::vector<int32_t> qw, cw, cstart(1, 0);
  std::vector<double> qv, cv;
  for (const auto &e : query) qw.push_back((int32_t)e.first), qv.push_back(e.second);  // std::map: ascending word ids
  for (KeyFrame *kf : candidates) {
    for (const auto &e : kf->bowVec_) cw.push_back((int32_t)e.first), cv.push_back(e.second);
    cstart.push_back((int32_t)cw.size());
  }
  if (vo_bow_score((int)qw.size(), qw.data(), qv.data(), (int)candidates.size(), cstart.data(), cw.data(), cv.data(),
                   scores.data()) != VO_OK)
    scores.assign(candidates.size(), 0.0);  // no error channel in the reference: "no similarity"
  return scores;
}

}  // namespace vo_shim
}  // namespace myslam
