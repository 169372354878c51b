// stub of the old non-templated Sophus API (see ../../README.md)
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <iosfwd>
namespace Sophus {
class SO3 {
 public:
  SO3(); SO3(const Eigen::Matrix3d &); SO3(const Eigen::Quaterniond &); SO3(double, double, double);
  Eigen::Matrix3d matrix() const; Eigen::Vector3d log() const; static SO3 exp(const Eigen::Vector3d &); SO3 inverse() const;
  const Eigen::Quaterniond &unit_quaternion() const; SO3 operator*(const SO3 &) const; Eigen::Vector3d operator*(const Eigen::Vector3d &) const;
  static Eigen::Matrix3d hat(const Eigen::Vector3d &);
};
class SE3 {
 public:
  SE3(); SE3(const SO3 &, const Eigen::Vector3d &); SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t); SE3(const Eigen::Quaterniond &q, const Eigen::Vector3d &t);
  Eigen::Matrix<double, 6, 1> log() const; static SE3 exp(const Eigen::Matrix<double, 6, 1> &);
  SE3 inverse() const; Eigen::Vector3d &translation(); const Eigen::Vector3d &translation() const; Eigen::Matrix3d rotation_matrix() const;
  const Eigen::Quaterniond &unit_quaternion() const; Eigen::Matrix4d matrix() const; SO3 &so3(); const SO3 &so3() const;
  void setRotationMatrix(const Eigen::Matrix3d &); void setQuaternion(const Eigen::Quaterniond &);
  SE3 operator*(const SE3 &) const; Eigen::Vector3d operator*(const Eigen::Vector3d &) const; SE3 &operator*=(const SE3 &);
};
std::ostream &operator<<(std::ostream &, const SE3 &);
}  // namespace Sophus
