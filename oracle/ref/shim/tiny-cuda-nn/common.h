// Host stand-in for tiny-cuda-nn/common.h (only TCNN_HOST_DEVICE is used by the reference's sensors.h); see vec.h.
#pragma once
#include <limits>
#include "vec.h"
