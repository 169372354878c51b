// FUNCTIONAL minimal stand-in for the old, non-templated Sophus API the reference uses (tests/shim_run/README.md):
// SE3 as unit quaternion + translation, exp / log on [upsilon; omega] (translation part first, as Sophus orders it).
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <cmath>
#include <ostream>
namespace Sophus {
inline Eigen::Matrix3d hat3(const Eigen::Vector3d &w) {
  Eigen::Matrix3d W;
  W(0, 1) = -w[2], W(0, 2) = w[1], W(1, 0) = w[2], W(1, 2) = -w[0], W(2, 0) = -w[1], W(2, 1) = w[0];
  return W;
}
class SO3 {
 public:
  Eigen::Quaterniond q;
  SO3() {}
  SO3(const Eigen::Matrix3d &R) : q(R) {}
  SO3(const Eigen::Quaterniond &q_) : q(q_.normalized()) {}
  Eigen::Matrix3d matrix() const { return q.toRotationMatrix(); }
  const Eigen::Quaterniond &unit_quaternion() const { return q; }
  SO3 inverse() const { return SO3(q.conjugate()); }
  SO3 operator*(const SO3 &o) const { return SO3(q * o.q); }
  Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return q * p; }
  static SO3 exp(const Eigen::Vector3d &w) {
    const double th = w.norm(), h = 0.5 * th;
    const double s = th < 1e-10 ? 0.5 - th * th / 48.0 : std::sin(h) / th;
    return SO3(Eigen::Quaterniond(std::cos(h), s * w[0], s * w[1], s * w[2]));
  }
  Eigen::Vector3d log() const {
    Eigen::Quaterniond u = q;
    if (u.w() < 0) u = Eigen::Quaterniond(-u.w(), -u.x(), -u.y(), -u.z());
    const Eigen::Vector3d v = u.vec();
    const double n = v.norm();
    const double k = n < 1e-10 ? 2.0 / u.w() - 2.0 * n * n / (3.0 * u.w() * u.w() * u.w()) : 2.0 * std::atan2(n, u.w()) / n;
    return v * k;
  }
};
class SE3 {
 public:
  SO3 r;
  Eigen::Vector3d t;
  SE3() {}
  SE3(const SO3 &r_, const Eigen::Vector3d &t_) : r(r_), t(t_) {}
  SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t_) : r(R), t(t_) {}
  SE3(const Eigen::Quaterniond &q, const Eigen::Vector3d &t_) : r(q), t(t_) {}
  static SE3 exp(const Eigen::Matrix<double, 6, 1> &xi) {
    const Eigen::Vector3d u(xi[0], xi[1], xi[2]), w(xi[3], xi[4], xi[5]);
    const double th = w.norm();
    const Eigen::Matrix3d W = hat3(w), W2 = W * W;
    double a, b;  // V = I + a W + b W^2
    if (th < 1e-10) a = 0.5, b = 1.0 / 6.0;
    else a = (1.0 - std::cos(th)) / (th * th), b = (th - std::sin(th)) / (th * th * th);
    const Eigen::Matrix3d V = Eigen::Matrix3d::Identity() + W * a + W2 * b;
    return SE3(SO3::exp(w), V * u);
  }
  Eigen::Matrix<double, 6, 1> log() const {
    const Eigen::Vector3d w = r.log();
    const double th = w.norm();
    const Eigen::Matrix3d W = hat3(w), W2 = W * W;
    double c;  // V^-1 = I - W / 2 + c W^2
    if (th < 1e-10) c = 1.0 / 12.0;
    else c = (1.0 - th * std::cos(0.5 * th) / (2.0 * std::sin(0.5 * th))) / (th * th);
    const Eigen::Vector3d u = (Eigen::Matrix3d::Identity() - W * 0.5 + W2 * c) * t;
    Eigen::Matrix<double, 6, 1> xi;
    for (int i = 0; i < 3; i++) xi[i] = u[i], xi[3 + i] = w[i];
    return xi;
  }
  SE3 inverse() const { const SO3 ri = r.inverse(); return SE3(ri, -(ri * t)); }
  Eigen::Vector3d &translation() { return t; }
  const Eigen::Vector3d &translation() const { return t; }
  Eigen::Matrix3d rotation_matrix() const { return r.matrix(); }
  const Eigen::Quaterniond &unit_quaternion() const { return r.unit_quaternion(); }
  SO3 &so3() { return r; }
  const SO3 &so3() const { return r; }
  SE3 operator*(const SE3 &o) const { return SE3(r * o.r, t + r * o.t); }
  Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return r * p + t; }
};
}  // namespace Sophus
