// stub (see ../../README.md): just enough of cv:: for -fsyntax-only
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(x) ((void)(x))
namespace cv {
struct Point2f { float x, y; };
struct Point2i { int x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
class Mat;
class _InputArray { public: _InputArray(); _InputArray(const Mat &); bool empty() const; Mat getMat() const; };
class _OutputArray : public _InputArray { public: _OutputArray(); _OutputArray(Mat &); void release() const; };
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
class Mat {
 public:
  Mat(); Mat(int rows, int cols, int type);
  unsigned char *data; int rows, cols; size_t step;
  int type() const; bool empty() const;
  template <class T> T *ptr(int r = 0); template <class T> const T *ptr(int r = 0) const;
  template <class T> T &at(int r, int c = 0);
  Mat row(int r) const; Mat rowRange(int a, int b) const; Mat clone() const;
  void copyTo(OutputArray) const; void create(int rows, int cols, int type); void release();
};
}  // namespace cv
