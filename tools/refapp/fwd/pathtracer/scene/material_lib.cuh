// forwards `#include "pathtracer/scene/material_lib.cuh"`: struct Material and iorToF0 live in the mirror's scene.hpp
#pragma once
#include "ptrt/scene.hpp"
