#!/bin/bash
# Experiment builds of the full path tracer that leave, per 8x8 tile, one quantity of the wave's run in tile_cost
# (vrt_full.hip.h, VRT_EXP_STATS): 1 rounds (rays marched one after the other), 2 ticks inside march(), 3 ticks inside shadow(),
# 4 ticks of the whole pixel function, 5 march-loop trips of the wave (sum over rounds of the longest march), 6 the heaviest
# lane's own trips. Builds build/exp/libvrt_hip_stats<k>.so here (CPU box); tools/room_stats.py reads them on the GPU box.
set -eu
cd "$(dirname "$0")/../voxel-raytracer_amd/csrc"
make -j8 ../libvrt_hip.so > /dev/null
mkdir -p build/exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall"
OBJS=$(ls build/*.o | grep -v vrt_launch_full)
for k in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DVRT_EXP_STATS=$k -c -o build/exp/vrt_launch_full_stats$k.o vrt_launch_full.hip &&
    /opt/rocm/bin/hipcc $FLAGS -shared -Wl,--version-script=vrt_exports.map -o build/exp/libvrt_hip_stats$k.so $OBJS build/exp/vrt_launch_full_stats$k.o ) &
done
wait
ls -la build/exp/*.so
