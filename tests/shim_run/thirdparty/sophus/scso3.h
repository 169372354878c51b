// FUNCTIONAL minimal stand-in (tests/shim_run/README.md): scaled rotation = scale * unit quaternion
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace Sophus {
class ScSO3 {
 public:
  Eigen::Quaterniond q;
  double s = 1.0;
  ScSO3() {}
  ScSO3(const Eigen::Quaterniond &q_) : q(q_.normalized()), s(q_.coeffs().squaredNorm()) {}
  ScSO3(double scale, const Eigen::Matrix3d &R) : q(R), s(scale) {}
  double scale() const { return s; }
  Eigen::Matrix3d rotationMatrix() const { return q.toRotationMatrix(); }
  const Eigen::Quaterniond &quaternion() const { return q; }
  ScSO3 inverse() const { ScSO3 r; r.q = q.conjugate(), r.s = 1.0 / s; return r; }
  ScSO3 operator*(const ScSO3 &o) const { ScSO3 r; r.q = (q * o.q).normalized(), r.s = s * o.s; return r; }
  Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return (q * p) * s; }
};
}  // namespace Sophus
