// stub declarations of the reference classes: only the members the shims touch (see ../README.md)
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
  float fx_, fy_, cx_, cy_, bf_, b_, xMin_, xMax_, yMin_, yMax_, gridPerPixelWidth_, gridPerPixelHeight_;
  Mat K_, distCoef_;
  Vector2d camera2pixel(const Vector3d &);
};
class Frame {
 public:
  unsigned long id_; string timeStamp_; Camera *camera_; SE3 Tcw_; bool poseExist_; KeyFrame *keyframe_trackRef_;
  vector<cv::KeyPoint> keypoints_, unKeypoints_; vector<float> depth_, uRight_; Mat descriptors_; vector<MapPoint *> mappoints_;
  vector<float> scaleFactors_; size_t N_; float xMin_, xMax_, yMin_, yMax_, gridPerPixelWidth_, gridPerPixelHeight_;
  vector<int> gridKeypoints_[FRAME_GRID_COLS][FRAME_GRID_ROWS]; vector<bool> outliers_;
  DBoW3::Vocabulary *voc_; DBoW3::BowVector bowVec_; DBoW3::FeatureVector featVec_; ORB_SLAM2::ORBextractor *orb_;
  Frame(Mat &grayImg, Mat &depthImg, string timeStamp, Camera *camera, ORB_SLAM2::ORBextractor *orb);
  void setPose(SE3 Tcw); void computeBow();
};
class KeyFrame {
 public:
  unsigned long id_; Camera *camera_; SE3 Tcw_;
  vector<cv::KeyPoint> unKeypoints_; vector<float> uRight_, depth_; Mat descriptors_; vector<MapPoint *> mappoints_;
  vector<float> scaleFactors_; size_t N_; float xMin_, xMax_, yMin_, yMax_;
  set<KeyFrame *> children_, loopEdges_; DBoW3::Vocabulary *voc_; DBoW3::BowVector bowVec_; DBoW3::FeatureVector featVec_;
  int relocateWordCnt_, loopWordCnt_; void computeBow();
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
class Sim3Solver {
 public:
  Sophus::Sim3 iterate(int iterations_req, bool &stopFlag, bool &emptyFlag, vector<bool> &inlierFlags, int &inliers_cnt);
 protected:
  int randomInt(int min, int max);
  KeyFrame *keyframe1_, *keyframe2_; bool fixScale_; int matches_cnt_;
  vector<MapPoint *> mappoints1_, mappoints2_; vector<Vector3d> pcams1_, pcams2_; vector<Vector2d> pixels1_, pixels2_;
  vector<int> maxError1_, maxError2_, matchedIndexs_; vector<bool> inlierFlags_, inlierFlags_best_;
  int inliers_cnt_, inliers_best_; double s12_best_; Sophus::Sim3 T12_best_; Matrix3d R12_best_; Vector3d t12_best_;
  int iterations_global_; double ransacProb_; int ransacInlierThreshold_, ransacMaxIters_; vector<int> idxForRandom_;
  double s12_; Sophus::Sim3 T12_; Matrix3d R12_; Vector3d t12_;
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
