#pragma once
#include "sophus/scso3.h"
namespace Sophus {
class Sim3 {
 public:
  Sim3(); Sim3(const ScSO3 &, const Eigen::Vector3d &);
  double scale() const; Eigen::Matrix3d rotation_matrix() const; Eigen::Vector3d translation() const;
  Eigen::Quaterniond quaternion() const; Sim3 inverse() const;
  Sim3 operator*(const Sim3 &) const; Eigen::Vector3d operator*(const Eigen::Vector3d &) const;
};
}  // namespace Sophus
