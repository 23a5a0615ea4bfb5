#!/bin/bash
# Builds libgigl_rt.so with extra -D flags into build/lib<TAG>.so (A/B timing with GIGL_RT_LIB).
# usage: tools/build_variant.sh TAG -DGIGL_X=1 ...
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p build/obj_$TAG
FLAGS="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 $*"
for f in gigl_rt gigl_fog gigl_shade gigl_multi gigl_photon; do
  /opt/rocm/bin/hipcc $FLAGS -c -o build/obj_$TAG/$f.o gigl_amd/csrc/$f.hip &
done
/opt/rocm/bin/hipcc $FLAGS -c -o build/obj_$TAG/kd_build.o gigl_amd/csrc/kd_build.cpp &
/opt/rocm/bin/hipcc $FLAGS -c -o build/obj_$TAG/kd_layout.o gigl_amd/csrc/kd_layout.cpp &
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o build/lib$TAG.so build/obj_$TAG/*.o -lpthread
echo build/lib$TAG.so
