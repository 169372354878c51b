// stub declarations of the reference classes: only the members the shims touch (see ../README.md)
#pragma once
#include <DBoW3/DBoW3.h>
#include "myslam/common_include.h"
namespace myslam {
class KeyFrame; class MapPoint; class Map; class Frame;
class Camera {
 public:
  float fx_, fy_, cx_, cy_, bf_, b_;
  Vector2d camera2pixel(const Vector3d &);
};
class Frame {
 public:
  Camera *camera_; SE3 Tcw_;
  vector<cv::KeyPoint> unKeypoints_; vector<float> uRight_; Mat descriptors_; vector<MapPoint *> mappoints_;
  vector<float> scaleFactors_; size_t N_; float xMin_, xMax_, yMin_, yMax_; vector<bool> outliers_;
  DBoW3::BowVector bowVec_; DBoW3::FeatureVector featVec_;
  void setPose(SE3 Tcw);
};
class KeyFrame {
 public:
  unsigned long id_; Camera *camera_;
  vector<cv::KeyPoint> unKeypoints_; vector<float> uRight_; Mat descriptors_; vector<MapPoint *> mappoints_;
  vector<float> scaleFactors_; size_t N_; float xMin_, xMax_, yMin_, yMax_;
  set<KeyFrame *> children_, loopEdges_; DBoW3::BowVector bowVec_; DBoW3::FeatureVector featVec_;
  unsigned long localBAKFId_, BAFixId_;
  SE3 getPose(); void setPose(SE3 &Tcw); bool isInImg(const float &u, const float &v); Vector3d getCamCenter();
  vector<KeyFrame *> getCovisiblesByWeight(const int &w); vector<MapPoint *> getMapPoints(); KeyFrame *getParent();
  vector<KeyFrame *> getOrderedKFs(); int getWeight(KeyFrame *); void addMapPoint(MapPoint *, const size_t &);
  void setMapPointNull(const size_t &); bool isBad();
};
class MapPoint {
 public:
  Vector3d pos_; KeyFrame *keyFrame_ref_; bool trackInLocalMap_; Mat descriptor_; int observe_cnt_;
  unsigned long loopCorrectByKF_, correctReference_, localBAKFId_; int trackScaleLevel_;
  float trackProj_u_, trackProj_uR_, trackProj_v_, viewCos_;
  map<KeyFrame *, size_t> observedKFs_; mutex mutexFeature_; static mutex mutexOptimizer_; bool badFlag_;
  map<KeyFrame *, size_t> getObservedKFs(); void addObservation(KeyFrame *, size_t); bool beObserved(KeyFrame *);
  void updateNormalAndDepth(); void computeDescriptor(); int predictScale(const float &, Frame *);
  int predictScale(const float &, KeyFrame *); void replaceMapPoint(MapPoint *); int getIndexInKeyFrame(KeyFrame *);
  Vector3d getPose(); void setPose(const Vector3d &); Mat getDescriptor(); Vector3d getNormalVector(); int getObsCnt();
  void eraseObservedKF(KeyFrame *); bool isBad(); float getMinDistanceThreshold(); float getMaxDistanceThreshold();
};
class Map {
 public:
  mutex mutexMapUpdate_; unsigned long maxKFId_;
  vector<KeyFrame *> getAllKeyFrames(); vector<MapPoint *> getAllMapPoints();
};
class LoopClosing {
 public:
  typedef map<KeyFrame *, Sophus::Sim3, less<KeyFrame *>, Eigen::aligned_allocator<pair<const KeyFrame *, Sophus::Sim3>>> KeyFrameAndPose;
};
class Matcher {
 public:
  Matcher() {}
  Matcher(float ratio);
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
