// include/myslam_shim/mappoint_hip.inl -- replacement body for MapPoint::computeDescriptor (reference
// src/mappoint.cpp:118-179): the N x N Hamming distances of the observing descriptors, the per-row medians and
// the arg-min run in ONE kernel launch (vo_median_descriptor); nothing but the chosen index comes back.
// #include at the bottom of a copy of mappoint.cpp from which that function was removed.
#include <vector>

#include "vo_hip.h"

namespace myslam {

void MapPoint::computeDescriptor() {
  map<KeyFrame *, size_t> observedKFs;
  {
    unique_lock<mutex> lock(mutexFeature_);
    if (badFlag_) return;
    observedKFs = observedKFs_;
  }
  if (observedKFs.empty()) return;
  std::vector<uint8_t> desc;
  std::vector<Mat> rows;
  for (auto it = observedKFs.begin(); it != observedKFs.end(); it++) {  // :131-136
    KeyFrame *kf = it->first;
    if (kf->isBad()) continue;
    Mat row = kf->descriptors_.row((int)it->second);
    desc.insert(desc.end(), row.ptr<uint8_t>(), row.ptr<uint8_t>() + 32);
    rows.push_back(row);
  }
  if (rows.empty()) return;
  const int32_t offsets[2] = {0, (int32_t)rows.size()};
  int32_t best = 0;
  if (vo_median_descriptor(desc.data(), 1, offsets, &best) != VO_OK || best < 0) return;
  unique_lock<mutex> lock(mutexFeature_);
  descriptor_ = rows[best].clone();  // :175-178
}

// Local mapping recomputes the descriptors of all points a new key-frame touches (localMapping.cpp:134-167,
// :388-418): the same selection for a whole list of map points in one launch.
inline void computeDescriptorsBatch(const std::vector<MapPoint *> &mps) {
  std::vector<uint8_t> desc;
  std::vector<int32_t> offsets(1, 0);
  std::vector<std::vector<Mat>> rows(mps.size());
  for (size_t k = 0; k < mps.size(); k++) {
    MapPoint *mp = mps[k];
    if (mp && !mp->isBad())
      for (auto &ob : mp->getObservedKFs()) {
        if (ob.first->isBad()) continue;
        Mat row = ob.first->descriptors_.row((int)ob.second);
        desc.insert(desc.end(), row.ptr<uint8_t>(), row.ptr<uint8_t>() + 32);
        rows[k].push_back(row);
      }
    offsets.push_back((int32_t)(desc.size() / 32));
  }
  std::vector<int32_t> best(mps.size() + 1, -1);
  if (vo_median_descriptor(desc.data(), (int)mps.size(), offsets.data(), best.data()) != VO_OK) return;
  for (size_t k = 0; k < mps.size(); k++)
    if (best[k] >= 0) {
      unique_lock<mutex> lock(mps[k]->mutexFeature_);
      mps[k]->descriptor_ = rows[k][best[k]].clone();
    }
}

}  // namespace myslam
