#!/bin/bash
# Build a PROBE variant of the library (runs in the build container, before gpurun): one source compiled with extra -D flags, linked with the product's other objects.
#   scripts/probe_build.sh <name> <source.hip> [-DFLAG ...]   ->  mac-vo_amd/csrc/build_probe/libprobe_<name>.so   (load it with MACVO_HIP_LIB=...; *.so travels to the GPU box)
set -eu
cd "$(dirname "$0")/../mac-vo_amd/csrc"
name=$1; src=$2; shift 2
make -j8 >/dev/null
mkdir -p build_probe
base=${src%.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 -mllvm -unroll-threshold=1000000 \
      "$@" -c "$src" -o "build_probe/$base.$name.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "build_probe/libprobe_$name.so" $(ls build/*.o | grep -v "build/$base\.o") "build_probe/$base.$name.o"
rm -f "build_probe/$base.$name.o"
echo "built mac-vo_amd/csrc/build_probe/libprobe_$name.so"
