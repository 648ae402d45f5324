#!/usr/bin/env bash
# Build libgcc_hip.so for gfx950 in-tree (no GPU needed: hipcc cross-compiles).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libgcc_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
mkdir -p build
pids=()
for f in conv_igemm conv_wgrad norm_act misc dwconv spectral attention srgan metric; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.hpp -nt build/$f.o ] || [ ../../include/gcc_hip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/conv_igemm.o build/conv_wgrad.o build/norm_act.o build/misc.o build/dwconv.o build/spectral.o build/attention.o build/srgan.o build/metric.o
echo "built $(realpath $OUT)"
