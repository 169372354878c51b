#include "myslam/types.h"
#include "myslam_shim/localmapping_hip.inl"
