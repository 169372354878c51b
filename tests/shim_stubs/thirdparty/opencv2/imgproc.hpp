// stub (see ../../README.md)
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
void cvtColor(InputArray, OutputArray, int code, int dcn = 0);
void resize(InputArray, OutputArray, Size, double fx = 0, double fy = 0, int interp = 1);
void GaussianBlur(InputArray, OutputArray, Size, double, double = 0, int = 4);
void undistortPoints(InputArray, OutputArray, InputArray K, InputArray D, InputArray R = noArray(), InputArray P = noArray());
void circle(InputOutputArray, Point, int, const Scalar &, int = 1, int = 8, int = 0);
void line(InputOutputArray, Point, Point, const Scalar &, int = 1, int = 8, int = 0);
void rectangle(InputOutputArray, Point, Point, const Scalar &, int = 1, int = 8, int = 0);
void putText(InputOutputArray, const std::string &, Point, int, double, Scalar, int = 1, int = 8, bool = false);
Size getTextSize(const std::string &, int, double, int, int *);
enum { COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, COLOR_BGRA2GRAY = 10, COLOR_RGBA2GRAY = 11, COLOR_GRAY2BGR = 8, FONT_HERSHEY_PLAIN = 1, INTER_LINEAR = 1, BORDER_REFLECT_101 = 4 };
}
#define CV_BGR2GRAY 6
#define CV_RGB2GRAY 7
#define CV_BGRA2GRAY 10
#define CV_RGBA2GRAY 11
#define CV_GRAY2BGR 8
