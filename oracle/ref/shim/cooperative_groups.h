// Host stand-in for cooperative_groups: this_grid().thread_rank() returns the index the driver loop in
// oracle/ref/ref_adam.cpp is currently at.  Test infrastructure only; contains no reference code.
#pragma once
#include <cstdint>
namespace cooperative_groups {
extern thread_local uint64_t shim_thread_rank;
struct grid_group { uint64_t thread_rank() const { return shim_thread_rank; } };
inline grid_group this_grid() { return {}; }
}  // namespace cooperative_groups
