#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace Sophus {
class ScSO3 {
 public:
  ScSO3(); ScSO3(const Eigen::Quaterniond &); ScSO3(double scale, const Eigen::Matrix3d &R); ScSO3(const Eigen::Matrix3d &sR);
  double scale() const; Eigen::Matrix3d rotationMatrix() const; Eigen::Matrix3d matrix() const; const Eigen::Quaterniond &quaternion() const;
  ScSO3 inverse() const; ScSO3 operator*(const ScSO3 &) const; Eigen::Vector3d operator*(const Eigen::Vector3d &) const;
};
}  // namespace Sophus
