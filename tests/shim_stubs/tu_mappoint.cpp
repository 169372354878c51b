#include "myslam/types.h"
#include "myslam_shim/mappoint_hip.inl"
