#include "myslam_shim/ORBextractor.h"
void use(cv::Mat &img, std::vector<cv::KeyPoint> &k, cv::Mat &d) {
  ORB_SLAM2::ORBextractor e(1000, 1.2f, 8, 20, 7);
  e(img, cv::Mat(), k, d);
  (void)e.GetScaleFactors();
}
