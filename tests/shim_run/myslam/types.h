// Test scaffolding (tests/shim_run/README.md): the builder's own, FUNCTIONAL re-declaration of the reference classes the
// shims of include/myslam_shim/ operate on -- only the members those shims touch, each doing the obvious thing the
// reference's accessor of that name does (include/myslam/{camera,frame,keyframe,mappoint,map}.h).  Members the executed
// paths never reach are declared and defined trivially so that every shim member links.
#pragma once
#include <DBoW3/DBoW3.h>
#include "myslam/common_include.h"
#include "myslam_shim/ORBextractor.h"
namespace myslam {
class KeyFrame; class MapPoint; class Map; class Frame;
#define FRAME_GRID_COLS 64
#define FRAME_GRID_ROWS 48
class Camera {
 public:
  float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0, bf_ = 0, b_ = 0, xMin_ = 0, xMax_ = 0, yMin_ = 0, yMax_ = 0, gridPerPixelWidth_ = 0, gridPerPixelHeight_ = 0;
  Mat K_, distCoef_;
  Vector2d camera2pixel(const Vector3d &p) { return Vector2d(fx_ * p[0] / p[2] + cx_, fy_ * p[1] / p[2] + cy_); }
};
class Frame {
 public:
  unsigned long id_ = 0; string timeStamp_; Camera *camera_ = nullptr; SE3 Tcw_; bool poseExist_ = false; KeyFrame *keyframe_trackRef_ = nullptr;
  vector<cv::KeyPoint> keypoints_, unKeypoints_; vector<float> depth_, uRight_; Mat descriptors_; vector<MapPoint *> mappoints_;
  vector<float> scaleFactors_; size_t N_ = 0; float xMin_ = 0, xMax_ = 0, yMin_ = 0, yMax_ = 0, gridPerPixelWidth_ = 0, gridPerPixelHeight_ = 0;
  vector<int> gridKeypoints_[FRAME_GRID_COLS][FRAME_GRID_ROWS]; vector<bool> outliers_;
  DBoW3::Vocabulary *voc_ = nullptr; DBoW3::BowVector bowVec_; DBoW3::FeatureVector featVec_; ORB_SLAM2::ORBextractor *orb_ = nullptr;
  Frame() {}
  Frame(Mat &grayImg, Mat &depthImg, string timeStamp, Camera *camera, ORB_SLAM2::ORBextractor *orb);  // frame_hip.inl
  void setPose(SE3 Tcw) { Tcw_ = Tcw, poseExist_ = true; }
  void computeBow();  // frame_hip.inl
};
class KeyFrame {
 public:
  unsigned long id_ = 0; Camera *camera_ = nullptr; SE3 Tcw_;
  vector<cv::KeyPoint> unKeypoints_; vector<float> uRight_, depth_; Mat descriptors_; vector<MapPoint *> mappoints_;
  vector<float> scaleFactors_; size_t N_ = 0; float xMin_ = 0, xMax_ = 0, yMin_ = 0, yMax_ = 0;
  set<KeyFrame *> children_, loopEdges_; DBoW3::Vocabulary *voc_ = nullptr; DBoW3::BowVector bowVec_; DBoW3::FeatureVector featVec_;
  int relocateWordCnt_ = 0, loopWordCnt_ = 0; void computeBow();
  unsigned long localBAKFId_ = ~0ul, BAFixId_ = ~0ul;
  bool bad_ = false; vector<KeyFrame *> ordered_; KeyFrame *parent_ = nullptr; int set_pose_calls_ = 0;
  SE3 getPose() { return Tcw_; }
  void setPose(SE3 &Tcw) { Tcw_ = Tcw, set_pose_calls_++; }
  bool isInImg(const float &u, const float &v) { return u >= xMin_ && u < xMax_ && v >= yMin_ && v < yMax_; }
  Vector3d getCamCenter() { return Tcw_.inverse().translation(); }
  vector<KeyFrame *> getCovisiblesByWeight(const int &) { return ordered_; }
  vector<MapPoint *> getMapPoints() { return mappoints_; }
  KeyFrame *getParent() { return parent_; }
  vector<KeyFrame *> getOrderedKFs() { return ordered_; }
  int getWeight(KeyFrame *) { return 0; }
  void addMapPoint(MapPoint *mp, const size_t &i) { mappoints_[i] = mp; }
  void setMapPointNull(const size_t &i) { mappoints_[i] = nullptr; }
  bool isBad() { return bad_; }
};
class MapPoint {
 public:
  Vector3d pos_; KeyFrame *keyFrame_ref_ = nullptr; bool trackInLocalMap_ = false; Mat descriptor_; int observe_cnt_ = 0;
  unsigned long loopCorrectByKF_ = ~0ul, correctReference_ = 0, localBAKFId_ = ~0ul; int trackScaleLevel_ = 0;
  float trackProj_u_ = 0, trackProj_uR_ = 0, trackProj_v_ = 0, viewCos_ = 0;
  map<KeyFrame *, size_t> observedKFs_; mutex mutexFeature_; static mutex mutexOptimizer_; bool badFlag_ = false;
  int normal_updates_ = 0; Vector3d normal_; float minDist_ = 0, maxDist_ = 1e9f;
  map<KeyFrame *, size_t> getObservedKFs() { return observedKFs_; }
  void addObservation(KeyFrame *kf, size_t i) { observedKFs_[kf] = i, observe_cnt_ += kf->uRight_[i] >= 0 ? 2 : 1; }
  bool beObserved(KeyFrame *kf) { return observedKFs_.count(kf) != 0; }
  void updateNormalAndDepth() { normal_updates_++; }
  void computeDescriptor();
  int predictScale(const float &, Frame *) { return trackScaleLevel_; }
  int predictScale(const float &, KeyFrame *) { return trackScaleLevel_; }
  void replaceMapPoint(MapPoint *) {}
  int getIndexInKeyFrame(KeyFrame *kf) { return observedKFs_.count(kf) ? (int)observedKFs_[kf] : -1; }
  Vector3d getPose() { return pos_; }
  void setPose(const Vector3d &p) { pos_ = p; }
  Mat getDescriptor() { return descriptor_; }
  Vector3d getNormalVector() { return normal_; }
  int getObsCnt() { return observe_cnt_; }
  void eraseObservedKF(KeyFrame *kf) {  // (mappoint.cpp:333-360 without the map bookkeeping of eraseMapPoint)
    if (!observedKFs_.count(kf)) return;
    observe_cnt_ -= kf->uRight_[observedKFs_[kf]] >= 0 ? 2 : 1;
    observedKFs_.erase(kf);
    if (observe_cnt_ <= 2) badFlag_ = true;
  }
  bool isBad() { return badFlag_; }
  float getMinDistanceThreshold() { return minDist_; }
  float getMaxDistanceThreshold() { return maxDist_; }
};
class Map {
 public:
  mutex mutexMapUpdate_; unsigned long maxKFId_ = 0; vector<KeyFrame *> kfs_; vector<MapPoint *> mps_;
  vector<KeyFrame *> getAllKeyFrames() { return kfs_; }
  vector<MapPoint *> getAllMapPoints() { return mps_; }
};
class LoopClosing {
 public:
  typedef map<KeyFrame *, Sophus::Sim3, less<KeyFrame *>, Eigen::aligned_allocator<pair<const KeyFrame *, Sophus::Sim3>>> KeyFrameAndPose;
};
class Matcher {
 public:
  Matcher() : ratio_(0.6f) {}
  Matcher(float ratio) : ratio_(ratio) {}
  int searchByProjection(Frame *, Frame *, const float radius, bool checkRot = true);
  int searchByProjection(Frame *, KeyFrame *, const float radius, const float distThreshold, const set<MapPoint *> &found, bool checkRot = true);
  int searchByProjection(Frame *, const vector<MapPoint *> &, const float thRadius);
  int searchByProjection(KeyFrame *, Sophus::Sim3 &, vector<MapPoint *> &, vector<MapPoint *> &, int th);
  int searchByBoW(KeyFrame *, Frame *, vector<MapPoint *> &, bool checkRot = true);
  int searchByBoW(KeyFrame *, KeyFrame *, vector<MapPoint *> &, bool checkRot);
  int searchBySim3(KeyFrame *, KeyFrame *, vector<MapPoint *> &, Sophus::Sim3 &, const float th);
  static int computeDistance(const Mat &, const Mat &);
  int searchForTriangulation(KeyFrame *, KeyFrame *, vector<pair<int, int>> &, Eigen::Matrix3d &F12, bool checkRot = true);
  int fuseMapPoints(KeyFrame *, vector<MapPoint *> &, const float &threshold);
  int fuseByPose(KeyFrame *, Sophus::Sim3 &, vector<MapPoint *> &, vector<MapPoint *> &, const float th);
 private:
  float ratio_;
};
class Optimizer {
 public:
  static int solvePoseOnlySE3(Frame *);
  static void solveLocalBAPoseAndPoint(KeyFrame *, bool &stopFlag, Map *);
  static int solveLoopSim3(KeyFrame *, KeyFrame *, vector<MapPoint *> &, Sophus::Sim3 &, const bool &fixScaleFlag);
  static int solvePoseGraphLoop(Map *, KeyFrame *, KeyFrame *, const LoopClosing::KeyFrameAndPose &,
                                const LoopClosing::KeyFrameAndPose &, const map<KeyFrame *, set<KeyFrame *>> &, const bool &);
};
}  // namespace myslam
