#!/bin/bash
# Build a development variant of the engine with the slab kernel's debug counters compiled in:
#   tools/build_debug.sh  ->  build/ab/libvrt_stats.so   (use with VRT_LIB=... VRT_DEBUG_STATS=1)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -DVRT_SLAB_STATS \
    -o build/ab/libvrt_stats.so volumeraytracer_amd/csrc/vrt_engine.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -DVRT_SLAB_TIMELINE \
    -o build/ab/libvrt_timeline.so volumeraytracer_amd/csrc/vrt_engine.hip
echo build/ab/libvrt_stats.so build/ab/libvrt_timeline.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -DVRT_SLAB_PHASES $VRT_EXTRA_DEFS \
    -o build/ab/libvrt_phases.so volumeraytracer_amd/csrc/vrt_engine.hip
echo build/ab/libvrt_phases.so
