// Stand-in for <torch/extension.h>: two kernel templates of the reference's particlePrimitives.cu take
// torch::PackedTensorAccessor32 arguments; they are never instantiated by oracle/ref/ref_grt_proxies.cpp, the declarations
// only have to parse.  TEST INFRASTRUCTURE ONLY; contains no reference code.
#pragma once
namespace torch {
struct RestrictPtrTraits {};
template <typename T, int N, typename P = void>
struct PackedTensorAccessor32 {
    struct Row { T& operator[](int) const; };
    Row operator[](int) const;
};
}  // namespace torch
