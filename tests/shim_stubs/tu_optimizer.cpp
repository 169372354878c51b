#include "myslam/types.h"
#include "myslam_shim/optimizer_hip.inl"
