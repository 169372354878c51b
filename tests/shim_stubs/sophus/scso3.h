#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace Sophus {
class ScSO3 { public: ScSO3(); ScSO3(const Eigen::Quaterniond &); ScSO3(double scale, const Eigen::Matrix3d &R); };
}  // namespace Sophus
