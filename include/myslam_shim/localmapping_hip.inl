// include/myslam_shim/localmapping_hip.inl -- the linear triangulation of LocalMapping::createNewMapPoints (reference
// src/localMapping.cpp:234-251: a 4 x 4 cv::SVD per match) for all matches of a neighbour key-frame in one launch
// (vo_triangulate), and MapPoint::computeDescriptor for every point the new key-frame touches in one launch
// (computeDescriptorsBatch, mappoint_hip.inl).  #include near the top of localMapping.cpp; the function itself changes
// in two places:
//
//   before the `for (int j = 0; j < matchIdxs.size(); j++)` loop (:192):
//       std::vector<Vector3d> tri_points; std::vector<char> tri_ok;
//       vo_shim::triangulateMatches(keyframe_curr_, kf, matchIdxs, Tcw1, Tcw2, tri_points, tri_ok);
//   the SVD block (:234-251) becomes:
//       if (!tri_ok[j]) continue;          // |x_3| < 1e-8 (:245-246)
//       p3d = tri_points[j];
//
// Every other statement of the function -- the parallax gates that decide whether a match is triangulated at all
// (:218-232), the depth, reprojection and scale checks behind it (:255-335), the map mutation -- is unchanged host code.
// The triangulation is evaluated for every match (the gate only decides whether its result is used): a 4 x 4
// eigen-decomposition per match is cheaper than selecting.
#include <vector>

#include "vo_hip.h"

namespace myslam {
namespace vo_shim {

inline void triangulateMatches(KeyFrame *kf1, KeyFrame *kf2, const vector<pair<int, int>> &matchIdxs, const Mat &Tcw1,
                               const Mat &Tcw2, std::vector<Vector3d> &points, std::vector<char> &ok) {
  const int n = (int)matchIdxs.size();
  points.assign(n, Vector3d(0, 0, 0));
  ok.assign(n, 0);
  if (n == 0) return;
  Camera *camera = kf1->camera_;
  std::vector<float> xn1((size_t)2 * n), xn2((size_t)2 * n), out((size_t)3 * n);
  std::vector<uint8_t> good(n);
  for (int j = 0; j < n; j++) {  // Camera::pixel2camera(kp, 1) (:209-210): the normalised coordinates, float like A (:236-239)
    const cv::KeyPoint &kp1 = kf1->unKeypoints_[matchIdxs[j].first], &kp2 = kf2->unKeypoints_[matchIdxs[j].second];
    xn1[2 * j] = (float)((kp1.pt.x - camera->cx_) / camera->fx_), xn1[2 * j + 1] = (float)((kp1.pt.y - camera->cy_) / camera->fy_);
    xn2[2 * j] = (float)((kp2.pt.x - camera->cx_) / camera->fx_), xn2[2 * j + 1] = (float)((kp2.pt.y - camera->cy_) / camera->fy_);
  }
  float T1[12], T2[12];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) T1[4 * r + c] = Tcw1.ptr<float>(r)[c], T2[4 * r + c] = Tcw2.ptr<float>(r)[c];
  if (vo_triangulate(n, xn1.data(), xn2.data(), T1, T2, 0, out.data(), good.data()) != VO_OK) return;
  for (int j = 0; j < n; j++) {
    ok[j] = (char)good[j];
    points[j] = Vector3d(out[3 * j], out[3 * j + 1], out[3 * j + 2]);
  }
}

}  // namespace vo_shim
}  // namespace myslam
