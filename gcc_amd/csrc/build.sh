#!/usr/bin/env bash
# Build libgcc_hip.so for gfx950 in-tree (no GPU needed: hipcc cross-compiles).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libgcc_hip.so
# -Wno-inline-asm: lds_dma16 (common.hpp) names m0 in its clobber list on purpose (the statement writes it); hipcc warns about any reserved register there
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
mkdir -p build
# GCC_BUILD_FORCE=1 (set by __graft_entry__.build()): recompile every source, whatever the timestamps of shipped objects say
if [ "${GCC_BUILD_FORCE:-0}" = "1" ]; then rm -f build/*.o $OUT; fi
pids=()
for f in conv_igemm conv_halo conv_wgrad norm_act misc dwconv spectral attention srgan metric comm replay; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.hpp -nt build/$f.o ] || [ igemm_common.hpp -nt build/$f.o ] || [ ../../include/gcc_hip.h -nt build/$f.o ]; then
    rm -f build/$f.o
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
for f in conv_igemm conv_halo conv_wgrad norm_act misc dwconv spectral attention srgan metric comm replay; do
  [ -f build/$f.o ] || { echo "compile of $f.hip failed" >&2; exit 1; }
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/conv_igemm.o build/conv_halo.o build/conv_wgrad.o build/norm_act.o build/misc.o build/dwconv.o build/spectral.o build/attention.o build/srgan.o build/metric.o build/comm.o build/replay.o -ldl -lpthread
echo "built $(realpath $OUT)"
