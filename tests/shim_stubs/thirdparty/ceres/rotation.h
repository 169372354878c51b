#pragma once
namespace ceres {
template <class T> void AngleAxisToRotationMatrix(const T *angle_axis, T *R);
template <class T> void RotationMatrixToAngleAxis(const T *R, T *angle_axis);
template <class T> void AngleAxisRotatePoint(const T angle_axis[3], const T pt[3], T result[3]);
template <class T> void AngleAxisToQuaternion(const T *angle_axis, T *q);
template <class T> void QuaternionToAngleAxis(const T *q, T *angle_axis);
}
