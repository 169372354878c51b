// stub (see ../../README.md): the OpenCV 1.x umbrella header the reference still includes
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/imgproc.hpp>
#include <opencv2/calib3d.hpp>
