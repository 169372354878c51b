#pragma once
#include "sophus/scso3.h"
#include <iosfwd>
namespace Sophus {
class Sim3 {
 public:
  Sim3(); Sim3(const ScSO3 &, const Eigen::Vector3d &); Sim3(const Eigen::Quaterniond &, const Eigen::Vector3d &);
  double scale() const; Eigen::Matrix3d rotation_matrix() const; Eigen::Vector3d &translation(); const Eigen::Vector3d &translation() const;
  const Eigen::Quaterniond &quaternion() const; Sim3 inverse() const; ScSO3 &scso3(); const ScSO3 &scso3() const;
  Eigen::Matrix4d matrix() const; Eigen::Matrix<double, 7, 1> log() const; static Sim3 exp(const Eigen::Matrix<double, 7, 1> &);
  Sim3 operator*(const Sim3 &) const; Eigen::Vector3d operator*(const Eigen::Vector3d &) const;
};
std::ostream &operator<<(std::ostream &, const Sim3 &);
}  // namespace Sophus
