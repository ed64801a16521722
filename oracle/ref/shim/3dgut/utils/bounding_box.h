// Shadows threedgut_tracer/include/3dgut/utils/bounding_box.h for the host builds under oracle/ref.
//  * ref_projector.cpp: the projection code only needs the TYPE to exist inside RenderParameters (renderParameters.h:41);
//    the stand-in below is used and the reference's box arithmetic is not exercised.
//  * ref_gut_render.cpp defines REF_REAL_BOUNDING_BOX: the shadow then steps aside and the reference's own header (with
//    BoundingBox::ray_intersect, used by initializeRay) is the one compiled, from where it lies.
// TEST INFRASTRUCTURE ONLY; contains no reference code.
#pragma once
#ifdef REF_REAL_BOUNDING_BOX
#include_next <3dgut/utils/bounding_box.h>
#else
#include <tiny-cuda-nn/common.h>
namespace threedgut {
struct BoundingBox {
    tcnn::vec3 min, max;
};
}  // namespace threedgut
#endif
