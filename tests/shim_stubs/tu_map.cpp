#include "myslam/types.h"
#include "myslam_shim/map_hip.inl"
