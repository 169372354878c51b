// FUNCTIONAL minimal stand-in (tests/shim_run/README.md)
#pragma once
#include "sophus/scso3.h"
namespace Sophus {
class Sim3 {
 public:
  ScSO3 r;
  Eigen::Vector3d t;
  Sim3() {}
  Sim3(const ScSO3 &r_, const Eigen::Vector3d &t_) : r(r_), t(t_) {}
  double scale() const { return r.scale(); }
  Eigen::Matrix3d rotation_matrix() const { return r.rotationMatrix(); }
  Eigen::Vector3d &translation() { return t; }
  const Eigen::Vector3d &translation() const { return t; }
  const Eigen::Quaterniond &quaternion() const { return r.quaternion(); }
  Sim3 inverse() const { const ScSO3 ri = r.inverse(); return Sim3(ri, -(ri * t)); }
  Sim3 operator*(const Sim3 &o) const { return Sim3(r * o.r, t + r * o.t); }
  Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return r * p + t; }
};
}  // namespace Sophus
