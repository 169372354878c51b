// stub (see ../../../README.md)
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
Mat imread(const std::string &, int flags = 1); bool imwrite(const std::string &, InputArray);
void imshow(const std::string &, InputArray); int waitKey(int = 0); void namedWindow(const std::string &, int = 1); void destroyAllWindows();
}
#define CV_LOAD_IMAGE_UNCHANGED -1
#define CV_LOAD_IMAGE_COLOR 1
