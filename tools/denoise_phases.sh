#!/bin/bash
# Experiment builds of the display pass that leave, in the first pixel of every 32x16 tile, the ticks its first wave spent in ONE phase
# (vrt_denoise.hip.h VRT_PH): builds csrc/build/exp/libvrt_hip_phase<k>.so for k = 1..9 here; tools/denoise_phases.py sums them on the GPU box.
set -eu
cd "$(dirname "$0")/../voxel-raytracer_amd/csrc"
make -j8 ../libvrt_hip.so > /dev/null
mkdir -p build/exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall"
OBJS=$(ls build/*.o | grep -v vrt_launch_misc)
for k in 1 2 3 4 5 6 7 8 9; do
  ( /opt/rocm/bin/hipcc $FLAGS -DVRT_DENOISE_PHASE=$k -c -o build/exp/misc_phase$k.o vrt_launch_misc.hip &&
    /opt/rocm/bin/hipcc $FLAGS -shared -Wl,--version-script=vrt_exports.map -o build/exp/libvrt_hip_phase$k.so $OBJS build/exp/misc_phase$k.o ) &
  if [ $((k % 5)) -eq 0 ]; then wait; fi
done
wait
ls build/exp/libvrt_hip_phase*.so
