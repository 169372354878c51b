// the names the reference's common header provides (tests/shim_run/README.md)
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <opencv2/core/core.hpp>
#include <sophus/se3.h>
#include <sophus/sim3.h>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>
using namespace std;
using cv::Mat;
using Eigen::Matrix3d;
using Eigen::Quaterniond;
using Eigen::Vector2d;
using Eigen::Vector3d;
using Sophus::SE3;
