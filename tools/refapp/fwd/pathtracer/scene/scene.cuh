// forwards the reference's `#include "pathtracer/scene/scene.cuh"` to the MI355X mirror of class Scene
#pragma once
#include "ptrt/scene.hpp"
