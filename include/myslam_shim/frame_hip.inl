// include/myslam_shim/frame_hip.inl -- replacement bodies for the Frame constructor (reference src/frame.cpp:14-34:
// ORB extraction :22, undistortKeyPoints :36-70, findDepth :108-133, assignFeaturesToGrid :72-89) and for
// Frame::computeBow / KeyFrame::computeBow (frame.cpp:248-253, keyframe.cpp:394-398).  #include at the bottom of a
// copy of frame.cpp from which the constructor, undistortKeyPoints, findDepth, assignFeaturesToGrid and computeBow were
// removed (KeyFrame::computeBow: likewise in keyframe.cpp; define VO_SHIM_KEYFRAME_BOW there).
//
// The image goes to the device once (vo_frames_construct: extraction, undistortion, depth look-up and the 64 x 48
// grid run there); what comes back is what Frame stores.  The BoW transform descends the vocabulary tree on the
// device (vo_bow_transform); the two std::map insert loops and the L1 normalisation of DBoW3::Vocabulary::transform
// stay here.  The vocabulary file is loaded once per DBoW3::Vocabulary object (vo_shim_register_vocabulary, called
// where the reference constructs it: test/vo_run.cpp:86).
//
// Needs the reference's headers (Frame, KeyFrame, Camera) and OpenCV / DBoW3: compile inside the reference tree.
// tests/test_shims_compile.py checks the syntax against stub declarations.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vo_hip.h"

namespace myslam {

namespace vo_shim {
// one frame store slot per host thread (Frame objects are constructed on the tracking thread, frame.cpp:22)
inline vo_frames *frame_store(Camera *camera) {
  thread_local vo_frames *h = nullptr;
  thread_local Camera *owner = nullptr;
  if (!h || owner != camera) {
    if (h) vo_frames_destroy(h);
    h = nullptr;
    if (vo_frames_create(&h, 1, 4096) != VO_OK) return nullptr;
    const float intr[5] = {camera->fx_, camera->fy_, camera->cx_, camera->cy_, camera->bf_};
    float dist[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5 && i < camera->distCoef_.rows; i++) dist[i] = camera->distCoef_.at<float>(i);  // camera.cpp:30-38
    vo_frames_set_camera(h, intr, dist, camera->xMax_ - camera->xMin_, camera->yMax_ - camera->yMin_);
    owner = camera;
  }
  return h;
}
// DBoW3::Vocabulary object -> device tree
inline std::map<const void *, vo_vocab *> &vocabularies() {
  static std::map<const void *, vo_vocab *> m;
  return m;
}
inline std::mutex &vocabulary_mutex() {
  static std::mutex m;
  return m;
}
}  // namespace vo_shim

// call once next to `new DBoW3::Vocabulary(path)` (test/vo_run.cpp:86) with the same path
inline bool vo_shim_register_vocabulary(const DBoW3::Vocabulary *voc, const std::string &path) {
  vo_vocab *v = nullptr;
  if (vo_vocab_load(path.c_str(), &v, nullptr, nullptr, nullptr, nullptr) != VO_OK) return false;
  std::lock_guard<std::mutex> lock(vo_shim::vocabulary_mutex());
  vo_shim::vocabularies()[voc] = v;
  return true;
}

static long unsigned int factory_id = 0;

Frame::Frame(Mat &grayImg, Mat &depthImg, string timeStamp, Camera *camera, ORB_SLAM2::ORBextractor *orb)
    : timeStamp_(timeStamp), camera_(camera), Tcw_(SE3()), poseExist_(false), keyframe_trackRef_(nullptr),
      xMin_(camera->xMin_), xMax_(camera->xMax_), yMin_(camera->yMin_), yMax_(camera->yMax_),
      gridPerPixelWidth_(camera->gridPerPixelWidth_), gridPerPixelHeight_(camera->gridPerPixelHeight_), voc_(nullptr),
      orb_(orb) {
  id_ = factory_id++;
  scaleFactors_ = orb_->GetScaleFactors();
  N_ = 0;
  vo_frames *store = vo_shim::frame_store(camera);
  if (!store || grayImg.empty()) return;
  const int cap = 4096;
  keypoints_.resize(cap);
  static_assert(sizeof(cv::KeyPoint) == sizeof(vo_keypoint), "cv::KeyPoint layout");
  int n = 0;
  // depthImg is the CV_32F image in metres (visualOdometry.cpp:162-163 converted it): depth_kind 1
  const int rc = vo_frames_construct(store, 0, orb_->handle(), grayImg.data, grayImg.cols, grayImg.rows, (int)grayImg.step,
                                     depthImg.empty() ? nullptr : depthImg.data, depthImg.empty() ? 0 : 1, (int)depthImg.step,
                                     1.0f, reinterpret_cast<vo_keypoint *>(keypoints_.data()), cap, &n);
  if (rc != VO_OK) n = 0;  // the reference has no error channel: behave like "no key-points" (:26-27)
  keypoints_.resize(n);
  N_ = n;
  if (n == 0) return;
  std::vector<float> x(cap), y(cap), angle(cap);
  std::vector<int32_t> octave(cap), cell_start(64 * 48 + 1);
  std::vector<uint16_t> cell_items(cap);
  uRight_.assign(cap, -1.f), depth_.assign(cap, -1.f);
  descriptors_.create(n, 32, CV_8U);
  std::vector<uint8_t> desc((size_t)cap * 32);
  int m = 0;
  if (vo_frames_download(store, 0, &m, x.data(), y.data(), octave.data(), angle.data(), uRight_.data(), depth_.data(),
                         desc.data(), cell_start.data(), cell_items.data(), nullptr) != VO_OK || m != n) {
    keypoints_.clear();
    N_ = 0;
    return;
  }
  uRight_.resize(n), depth_.resize(n);
  std::memcpy(descriptors_.data, desc.data(), (size_t)n * 32);
  unKeypoints_ = keypoints_;  // :47-67: same attributes, undistorted position
  for (int i = 0; i < n; i++) unKeypoints_[i].pt.x = x[i], unKeypoints_[i].pt.y = y[i];
  for (int ix = 0; ix < FRAME_GRID_COLS; ix++)  // the CSR grid: cell = ix * 48 + iy, items in push_back order (:72-89)
    for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
      const int c = ix * FRAME_GRID_ROWS + iy;
      gridKeypoints_[ix][iy].assign(cell_items.begin() + cell_start[c], cell_items.begin() + cell_start[c + 1]);
    }
  mappoints_ = vector<MapPoint *>(N_, static_cast<MapPoint *>(nullptr));
  outliers_ = vector<bool>(N_, false);
}

namespace vo_shim {
// DBoW3::Vocabulary::transform(features, bowVec, featVec, levelsup): word / weight / node per feature on the device,
// then v.addWeight(word, weight) (TF_IDF / L1_NORM defaults), featVec.addFeature(node, i) in feature order, v.normalize(L1)
inline void transform(const void *voc, const Mat &descriptors, DBoW3::BowVector &bow, DBoW3::FeatureVector &feat, int levelsup) {
  vo_vocab *v = nullptr;
  {
    std::lock_guard<std::mutex> lock(vocabulary_mutex());
    auto it = vocabularies().find(voc);
    if (it != vocabularies().end()) v = it->second;
  }
  bow.clear(), feat.clear();
  const int n = descriptors.rows;
  if (!v || n == 0) return;
  std::vector<int32_t> word(n), node(n);
  std::vector<double> weight(n);
  if (vo_bow_transform(v, n, descriptors.data, levelsup, word.data(), weight.data(), node.data()) != VO_OK) return;
  for (int i = 0; i < n; i++) {
    if (weight[i] > 0) {  // Vocabulary::transform skips stop words (weight 0)
      bow.addWeight((unsigned)word[i], weight[i]);
      feat.addFeature((unsigned)node[i], (unsigned)i);
    }
  }
  bow.normalize(DBoW3::L1);
}
}  // namespace vo_shim

void Frame::computeBow() {
  if (featVec_.empty() || bowVec_.empty()) vo_shim::transform(voc_, descriptors_, bowVec_, featVec_, 3);
}

#ifdef VO_SHIM_KEYFRAME_BOW
void KeyFrame::computeBow() {
  if (bowVec_.empty()) vo_shim::transform(voc_, descriptors_, bowVec_, featVec_, 3);
}
#endif

}  // namespace myslam
