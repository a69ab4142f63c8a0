// forwards `#include "common/glfw_view_interop.hpp"` to the rtgl:: calls over the HIP presentation ring
#pragma once
#include "ptrt/view.hpp"
