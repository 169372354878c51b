// include/myslam_shim/ORBextractor.h -- drop-in for the reference's include/myslam/ORBextractor.h.
//
// Same namespace, class name, constructor and operator() signature, accessors and the public
// mvImagePyramid member (reference ORBextractor.h:45-111), implemented over the C-ABI of
// include/vo_hip.h.  Put this directory in front of the reference's include path as
// "myslam/ORBextractor.h", drop src/ORBextractor.cpp from src/CMakeLists.txt and link libvo_hip.so;
// frame.cpp:22-23 and visualOdometry.cpp:31 then compile unchanged.
//
// Needs OpenCV headers (cv::Mat, cv::KeyPoint), which this repository's build image does not have:
// the file is compile-checked only where the reference's own dependencies exist.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <opencv2/core/core.hpp>

#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#include "vo_hip.h"

namespace ORB_SLAM2 {

// The reference declares the oct-tree node publicly (ORBextractor.h:31-43) although nothing outside ORBextractor.cpp
// uses it; kept so that code naming the type still compiles.  DistributeOctTree itself runs on the device (k_octree):
// DivideNode is declared, as there, and deliberately not defined here.
class ExtractorNode {
 public:
  ExtractorNode() : bNoMore(false) {}
  void DivideNode(ExtractorNode &n1, ExtractorNode &n2, ExtractorNode &n3, ExtractorNode &n4);
  std::vector<cv::KeyPoint> vKeys;
  cv::Point2i UL, UR, BL, BR;
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore;
};

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : nlevels_(nlevels), scaleFactor_(scaleFactor) {
    if (vo_orb_create(&h_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) != VO_OK)
      throw std::runtime_error(std::string("vo_orb_create: ") + vo_last_error());
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    vo_orb_scale_factors(h_, mvScaleFactor.data(), mvInvScaleFactor.data());
    mvImagePyramid.resize(nlevels);
    capacity_ = vo_orb_max_keypoints(h_);
  }
  ~ORBextractor() { vo_orb_destroy(h_); }
  ORBextractor(const ORBextractor &) = delete;
  ORBextractor &operator=(const ORBextractor &) = delete;

  // ORBextractor::operator(), reference src/ORBextractor.cpp:1051-1112.  mask is ignored there too.
  void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints,
                  cv::OutputArray descriptors) {
    if (_image.empty()) return;  // :1054-1055
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    static_assert(sizeof(cv::KeyPoint) == sizeof(vo_keypoint), "cv::KeyPoint layout");
    keypoints.resize(capacity_);
    cv::Mat desc(capacity_, 32, CV_8U);
    int n = 0;
    const int rc = vo_orb_extract(h_, image.data, image.cols, image.rows, (int)image.step,
                                  reinterpret_cast<vo_keypoint *>(keypoints.data()), desc.data, capacity_, &n);
    if (rc != VO_OK) n = 0;  // the reference has no error channel: behave like "no key-points"
    keypoints.resize(n);
    if (n == 0)
      descriptors.release();  // :1073-1074
    else
      desc.rowRange(0, n).copyTo(descriptors);
    pyramid_valid_ = false;
  }

  // the C-ABI handle, for the shims that keep a frame on the device behind the extraction (frame_hip.inl)
  vo_orb *handle() { return h_; }

  int inline GetLevels() { return nlevels_; }
  float inline GetScaleFactor() { return scaleFactor_; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }

  // The reference exposes the pyramid as a public member that nothing outside the class reads
  // (SURVEY.md Q-E5).  It is fetched from the device on demand.
  std::vector<cv::Mat> mvImagePyramid;
  void FetchImagePyramid() {
    for (int l = 0; l < nlevels_; l++) {
      int w = 0, hgt = 0;
      vo_orb_get_level(h_, 0, l, 0, nullptr, 0, &w, &hgt);
      mvImagePyramid[l].create(hgt, w, CV_8U);
      vo_orb_get_level(h_, 0, l, 0, mvImagePyramid[l].data, (int)mvImagePyramid[l].step, &w, &hgt);
    }
    pyramid_valid_ = true;
  }

 protected:
  vo_orb *h_ = nullptr;
  int nlevels_, capacity_ = 0;
  float scaleFactor_;
  bool pyramid_valid_ = false;
  std::vector<float> mvScaleFactor, mvInvScaleFactor;
};

}  // namespace ORB_SLAM2

#endif
