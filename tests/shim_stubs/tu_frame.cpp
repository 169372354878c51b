#include "myslam/types.h"
#define VO_SHIM_KEYFRAME_BOW 1
#include "myslam_shim/frame_hip.inl"
