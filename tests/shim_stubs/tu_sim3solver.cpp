#include "myslam/types.h"
#include "myslam_shim/sim3solver_hip.inl"
