// stub (see ../../../README.md): just enough of cv:: for -fsyntax-only
#pragma once
#include <cstddef>
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>
#define CV_8U 0
#define CV_16U 2
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5
#define CV_64FC1 6
#define CV_Assert(x) ((void)(x))
#define CV_PI 3.1415926535897932384626433832795
namespace cv {
typedef unsigned char uchar;
typedef std::string String;
template <class T> struct Point_ { T x, y; Point_(); Point_(T, T); template <class U> Point_(const Point_<U> &); Point_ operator+(const Point_ &) const; Point_ operator-(const Point_ &) const; Point_ operator*(double) const; };
typedef Point_<float> Point2f; typedef Point_<int> Point2i; typedef Point_<double> Point2d; typedef Point2i Point;
template <class T> struct Point3_ { T x, y, z; Point3_(); Point3_(T, T, T); };
typedef Point3_<float> Point3f; typedef Point3_<double> Point3d;
template <class T> struct Size_ { T width, height; Size_(); Size_(T, T); };
typedef Size_<int> Size;
template <class T> struct Rect_ { T x, y, width, height; Rect_(); Rect_(T, T, T, T); };
typedef Rect_<int> Rect;
struct Range { int start, end; Range(); Range(int, int); static Range all(); };
template <class T, int N> struct Vec { T val[N]; T &operator[](int); const T &operator[](int) const; };
typedef Vec<double, 4> Scalar_d;
struct Scalar { double val[4]; Scalar(); Scalar(double); Scalar(double, double, double, double = 0); static Scalar all(double); };
struct KeyPoint {
  Point2f pt; float size, angle, response; int octave, class_id;
  KeyPoint(); KeyPoint(float x, float y, float size, float angle = -1, float response = 0, int octave = 0, int class_id = -1);
};
class Mat; class MatExpr;
template <class T> class Mat_;
class _InputArray {
 public:
  _InputArray(); _InputArray(const Mat &); _InputArray(const MatExpr &);
  template <class T> _InputArray(const std::vector<T> &); template <class T> _InputArray(const Mat_<T> &);
  _InputArray(const double &);
  bool empty() const; Mat getMat() const;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(); _OutputArray(Mat &); template <class T> _OutputArray(std::vector<T> &); template <class T> _OutputArray(Mat_<T> &);
  void release() const;
};
typedef const _InputArray &InputArray; typedef const _OutputArray &OutputArray; typedef const _OutputArray &InputOutputArray;
InputArray noArray();
class MatSize { public: int operator[](int) const; Size operator()() const; };
class Mat {
 public:
  Mat(); Mat(int rows, int cols, int type); Mat(int rows, int cols, int type, const Scalar &); Mat(int rows, int cols, int type, void *data, size_t step = 0);
  Mat(Size, int type); Mat(const Mat &); Mat(const MatExpr &); template <class T> explicit Mat(const std::vector<T> &, bool copy = false);
  Mat(const Mat &, const Rect &); Mat(const Mat &, const Range &, const Range &);
  Mat &operator=(const Mat &); Mat &operator=(const MatExpr &); Mat &operator=(const Scalar &);
  unsigned char *data; int rows, cols, flags, dims; size_t step; MatSize size;
  int type() const; int depth() const; int channels() const; bool empty() const; bool isContinuous() const; size_t total() const;
  size_t elemSize() const; size_t elemSize1() const; size_t step1(int = 0) const;
  template <class T> T *ptr(int r = 0); template <class T> const T *ptr(int r = 0) const;
  unsigned char *ptr(int r = 0); const unsigned char *ptr(int r = 0) const;
  template <class T> T &at(int r, int c = 0); template <class T> const T &at(int r, int c = 0) const;
  template <class T> T &at(Point); template <class T> const T &at(Point) const;
  Mat row(int r) const; Mat col(int c) const; Mat rowRange(int a, int b) const; Mat colRange(int a, int b) const; Mat clone() const;
  Mat operator()(const Rect &) const; Mat operator()(Range, Range) const;
  Mat reshape(int cn, int rows = 0) const; MatExpr t() const; MatExpr inv(int = 0) const; MatExpr mul(InputArray, double = 1) const;
  Mat cross(InputArray) const; double dot(InputArray) const;
  void copyTo(OutputArray) const; void copyTo(OutputArray, InputArray) const; void convertTo(OutputArray, int rtype, double alpha = 1, double beta = 0) const;
  void create(int rows, int cols, int type); void create(Size, int type); void release(); Mat &setTo(InputArray, InputArray = noArray());
  void push_back(const Mat &); template <class T> void push_back(const T &);
  Mat &operator/=(double); Mat &operator*=(double); Mat &operator+=(const Mat &); Mat &operator-=(const Mat &);
  static MatExpr zeros(int, int, int); static MatExpr zeros(Size, int); static MatExpr ones(int, int, int); static MatExpr eye(int, int, int);
};
class MatExpr {
 public:
  MatExpr(); MatExpr(const Mat &); operator Mat() const;
  MatExpr t() const; MatExpr inv(int = 0) const; MatExpr mul(const MatExpr &, double = 1) const; MatExpr mul(const Mat &, double = 1) const;
  Mat row(int) const; Mat col(int) const; Mat rowRange(int, int) const; Mat colRange(int, int) const; Mat cross(const Mat &) const; double dot(const Mat &) const;
  template <class T> T &at(int r, int c = 0);
};
MatExpr operator+(const Mat &, const Mat &); MatExpr operator-(const Mat &, const Mat &); MatExpr operator*(const Mat &, const Mat &);
MatExpr operator+(const MatExpr &, const MatExpr &); MatExpr operator-(const MatExpr &, const MatExpr &); MatExpr operator*(const MatExpr &, const MatExpr &);
MatExpr operator+(const MatExpr &, const Mat &); MatExpr operator-(const MatExpr &, const Mat &); MatExpr operator*(const MatExpr &, const Mat &);
MatExpr operator+(const Mat &, const MatExpr &); MatExpr operator-(const Mat &, const MatExpr &); MatExpr operator*(const Mat &, const MatExpr &);
MatExpr operator*(const Mat &, double); MatExpr operator*(double, const Mat &); MatExpr operator/(const Mat &, double);
MatExpr operator*(const MatExpr &, double); MatExpr operator*(double, const MatExpr &); MatExpr operator/(const MatExpr &, double);
MatExpr operator-(const Mat &); MatExpr operator-(const MatExpr &);
MatExpr operator+(const Mat &, const Scalar &); MatExpr operator-(const Mat &, const Scalar &);
std::ostream &operator<<(std::ostream &, const Mat &);
template <class T> class MatCommaInit { public: template <class V> MatCommaInit &operator,(V); operator Mat_<T>() const; operator Mat() const; };
template <class T> class Mat_ : public Mat {
 public:
  Mat_(); Mat_(int rows, int cols); Mat_(const Mat &); Mat_(const MatExpr &); template <class U> Mat_(const MatCommaInit<U> &);
  T &operator()(int r, int c = 0); const T &operator()(int r, int c = 0) const;
  template <class V> MatCommaInit<T> operator<<(V);
};
double norm(InputArray, int = 4); double norm(InputArray, InputArray, int = 4); double determinant(InputArray);
enum { NORM_L1 = 2, NORM_L2 = 4, NORM_HAMMING = 6 };
class SVD {
 public:
  enum { MODIFY_A = 1, NO_UV = 2, FULL_UV = 4 };
  static void compute(InputArray src, OutputArray w, OutputArray u, OutputArray vt, int flags = 0);
  static void compute(InputArray src, OutputArray w, int flags = 0);
};
class FileNode {
 public:
  operator int() const; operator float() const; operator double() const; operator std::string() const;
  bool empty() const; bool isNone() const; FileNode operator[](const std::string &) const; FileNode operator[](const char *) const;
};
void operator>>(const FileNode &, int &); void operator>>(const FileNode &, float &); void operator>>(const FileNode &, double &);
void operator>>(const FileNode &, std::string &); void operator>>(const FileNode &, Mat &);
class FileStorage {
 public:
  enum { READ = 0, WRITE = 1 };
  FileStorage(); FileStorage(const std::string &, int); ~FileStorage();
  bool open(const std::string &, int); bool isOpened() const; void release();
  FileNode operator[](const std::string &) const; FileNode operator[](const char *) const;
};
template <class T> T saturate_cast(double); int cvRound(double); int cvFloor(double); int cvCeil(double); float fastAtan2(float y, float x);
void hconcat(InputArray, InputArray, OutputArray); void vconcat(InputArray, InputArray, OutputArray);
}  // namespace cv
using cv::cvRound; using cv::cvFloor; using cv::cvCeil;
