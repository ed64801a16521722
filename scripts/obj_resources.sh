#!/bin/bash
# obj_resources.sh FILE.o [filter] — LDS / scratch / VGPRs / spills of the gfx950 kernels inside a host object (from its code-object metadata)
set -e
o=$(readlink -f "$1"); flt=${2:-}
d=$(mktemp -d); cd "$d"
cp "$o" x.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1 || true
co=$(ls x.o.*gfx950* 2>/dev/null | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | grep -E '^\s+\.name:|\.vgpr_count|\.private_segment_fixed_size|\.vgpr_spill|\.group_segment_fixed_size' | paste - - - - - | sed 's/ \+/ /g' | c++filt \
  | sed -E 's/.*group_segment_fixed_size: ([0-9]+).*\.name: (void )?(grut::)?(\(anonymous namespace\)::)?([A-Za-z0-9_]+(<[^>(]*>)?).*private_segment_fixed_size: ([0-9]+).*vgpr_count: ([0-9]+).*vgpr_spill_count: ([0-9]+).*/lds \1 scratch \7 vgpr \8 spill \9  \5/' | grep -- "$flt"
rm -rf "$d"
