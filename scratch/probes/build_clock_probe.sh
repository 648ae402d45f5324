#!/usr/bin/env bash
# diagnostic libraries: libgcc_hip.so with the clock stamps of igemm_kernel compiled in (-DGCC_CLOCK_PROBE: entry, k loop start / end,
# kernel end per workgroup), one per value of GCC_IGEMM_ROT (the k loop with the barrier at the top of the step / between the k-slices)
set -euo pipefail
cd "$(dirname "$0")/../../gcc_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
for rot in ${ROTS:-0}; do
  hipcc $FLAGS -DGCC_CLOCK_PROBE -DGCC_IGEMM_ROT=$rot -c conv_igemm.hip -o build/conv_igemm_probe$rot.o &
done
wait
for rot in ${ROTS:-0}; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/probes/libgcc_hip_probe_rot$rot.so build/conv_igemm_probe$rot.o build/conv_halo.o build/conv_wgrad.o build/norm_act.o build/misc.o build/dwconv.o build/spectral.o build/attention.o build/srgan.o build/metric.o build/comm.o build/replay.o -ldl -lpthread
done
echo built
