// stub (see ../../README.md)
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
enum { SOLVEPNP_ITERATIVE = 0, SOLVEPNP_EPNP = 1, SOLVEPNP_P3P = 2 };
bool solvePnPRansac(InputArray objectPoints, InputArray imagePoints, InputArray K, InputArray dist, OutputArray rvec, OutputArray tvec,
                    bool useExtrinsicGuess = false, int iterationsCount = 100, float reprojectionError = 8.0, double confidence = 0.99,
                    OutputArray inliers = _OutputArray(), int flags = 0);
bool solvePnP(InputArray, InputArray, InputArray, InputArray, OutputArray, OutputArray, bool = false, int = 0);
void Rodrigues(InputArray src, OutputArray dst, OutputArray jacobian = _OutputArray());
}
