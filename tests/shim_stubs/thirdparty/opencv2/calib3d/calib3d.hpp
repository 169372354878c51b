#pragma once
#include <opencv2/calib3d.hpp>
