// FUNCTIONAL minimal stand-in for the part of cv:: the shims of include/myslam_shim/ touch (tests/shim_run/README.md):
// a reference-counted 2-D single-channel Mat, KeyPoint with OpenCV's 28-byte layout, Input/OutputArray over Mat.
#pragma once
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#define CV_8U 0
#define CV_16U 2
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0
#define CV_32FC1 5
#define CV_Assert(x) assert(x)
namespace cv {
typedef unsigned char uchar;
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f; typedef Point_<int> Point2i; typedef Point2i Point;
struct KeyPoint {
  Point2f pt; float size, angle, response; int octave, class_id;
  KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};
inline size_t elem_bytes(int type) { return type == CV_8U ? 1 : type == CV_16U ? 2 : type == CV_64F ? 8 : 4; }
class Mat;
class _InputArray {
 public:
  const Mat *m_;
  _InputArray() : m_(nullptr) {}
  _InputArray(const Mat &m) : m_(&m) {}
  bool empty() const;
  Mat getMat() const;
};
class _OutputArray : public _InputArray {
 public:
  Mat *o_;
  _OutputArray() : o_(nullptr) {}
  _OutputArray(Mat &m) : _InputArray(m), o_(&m) {}
  void release() const;
};
typedef const _InputArray &InputArray; typedef const _OutputArray &OutputArray;
class Mat {
 public:
  std::shared_ptr<std::vector<unsigned char>> buf_;
  unsigned char *data = nullptr;
  int rows = 0, cols = 0, type_ = 0;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void *d, size_t st = 0) : data((unsigned char *)d), rows(r), cols(c), type_(type), step(st ? st : c * elem_bytes(type)) {}
  void create(int r, int c, int type) {
    if (buf_ && r == rows && c == cols && type == type_) return;
    buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * elem_bytes(type) + 64, 0);
    data = buf_->data(), rows = r, cols = c, type_ = type, step = c * elem_bytes(type);
  }
  void release() { buf_.reset(), data = nullptr, rows = cols = 0, step = 0; }
  int type() const { return type_; }
  bool empty() const { return !data || rows == 0 || cols == 0; }
  template <class T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
  template <class T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }
  template <class T> T &at(int r, int c = 0) { return ptr<T>(r)[c]; }
  template <class T> const T &at(int r, int c = 0) const { return ptr<T>(r)[c]; }
  Mat rowRange(int a, int b) const { Mat m = *this; m.data = data + (size_t)a * step, m.rows = b - a; return m; }
  Mat row(int r) const { return rowRange(r, r + 1); }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int r = 0; r < rows; r++) std::memcpy(m.ptr<unsigned char>(r), ptr<unsigned char>(r), cols * elem_bytes(type_));
    return m;
  }
  void copyTo(OutputArray o) const { *o.o_ = clone(); }
};
inline bool _InputArray::empty() const { return !m_ || m_->empty(); }
inline Mat _InputArray::getMat() const { return m_ ? *m_ : Mat(); }
inline void _OutputArray::release() const { if (o_) o_->release(); }
}  // namespace cv
