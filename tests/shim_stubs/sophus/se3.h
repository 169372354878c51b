// stub of the old non-templated Sophus API (see ../README.md)
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace Sophus {
class SE3 {
 public:
  SE3(); SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t); SE3(const Eigen::Quaterniond &q, const Eigen::Vector3d &t);
  Eigen::Matrix<double, 6, 1> log() const;
  static SE3 exp(const Eigen::Matrix<double, 6, 1> &);
  SE3 inverse() const; Eigen::Vector3d translation() const; Eigen::Matrix3d rotation_matrix() const;
  Eigen::Quaterniond unit_quaternion() const;
  SE3 operator*(const SE3 &) const; Eigen::Vector3d operator*(const Eigen::Vector3d &) const;
};
}  // namespace Sophus
