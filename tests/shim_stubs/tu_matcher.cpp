#include "myslam/types.h"
#include "myslam_shim/matcher_hip.inl"
