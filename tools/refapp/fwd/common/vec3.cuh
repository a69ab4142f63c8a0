// forwards `#include "common/vec3.cuh"` (PTRTtransfer.cuh:27) to the mirror's vec3 -- the same type Scene's API takes
#pragma once
#include "ptrt/math.hpp"
