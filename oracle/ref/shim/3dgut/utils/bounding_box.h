// Shadows threedgut_tracer/include/3dgut/utils/bounding_box.h for the host build of oracle/ref/ref_projector.cpp: the
// projection code only needs the TYPE to exist inside RenderParameters (renderParameters.h:41); the reference's box
// arithmetic is not exercised there (the oracle's ray / box test is pinned through the renderer tests instead).
// TEST INFRASTRUCTURE ONLY; contains no reference code.
#pragma once
#include <tiny-cuda-nn/common.h>
namespace threedgut {
struct BoundingBox {
    tcnn::vec3 min, max;
};
}  // namespace threedgut
