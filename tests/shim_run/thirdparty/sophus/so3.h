#pragma once
#include "sophus/se3.h"
