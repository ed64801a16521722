#!/bin/bash
# build_variant.sh NAME FILE.hip [extra hipcc flags...] — development aid: links a variant of the library in which one source
# is compiled with extra flags (e.g. -DGRUT_DIAG_...) into variants/libgrut_amd_NAME.so; select it with GRUT_AMD_LIB=...
set -e
cd "$(dirname "$0")/../3dgrut_amd/csrc"
name=$1; src=$2; shift 2; out=${OUTNAME:-${src%.hip}}
mkdir -p ../../variants   # (scratch: variants/ is not tracked and is emptied after every experiment)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize"
[ "$src" = "grt_kernels.hip" ] && FLAGS="$FLAGS -ffp-contract=on"
/opt/rocm/bin/hipcc -x hip $FLAGS "$@" -c "$src" -o "../../variants/${name}_${src%.hip}.o"
objs=""
for f in scan_sort gut_kernels gut_poses gut_render gut_api grt_kernels grt_api optim; do
  if [ "$f.hip" = "$src" ]; then objs="$objs ../../variants/${name}_${f}.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o "../../variants/libgrut_amd_${name}.so"
echo "variants/libgrut_amd_${name}.so"
