#!/usr/bin/env bash
# Build libgcc_hip.so for gfx950 in-tree (no GPU needed: hipcc cross-compiles), and beside it libgcc_hip_diag.so: the same
# sources with -DGCC_DIAG_BUILD (common.hpp: timing ablations whose results are wrong, the error-path switch of the grid
# InstanceNorm, gcc_diag_set) -- only the four sources that hold such code are compiled twice.  The product never loads the
# diagnostic variant; tests and probes name it through GCC_HIP_LIB.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libgcc_hip.so
OUT_DIAG=../libgcc_hip_diag.so
# -Wno-inline-asm: lds_dma16 (common.hpp) names m0 in its clobber list on purpose (the statement writes it); hipcc warns about any reserved register there
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
SRCS="conv_igemm conv_halo conv_wgrad conv_thinout conv_ring3 norm_act misc dwconv spectral attention srgan metric comm replay"
DIAG_SRCS="conv_igemm conv_wgrad norm_act misc"
mkdir -p build
# GCC_BUILD_FORCE=1 (set by __graft_entry__.build()): recompile every source, whatever the timestamps of shipped objects say
if [ "${GCC_BUILD_FORCE:-0}" = "1" ]; then rm -f build/*.o $OUT $OUT_DIAG; fi
stale() {   # object, source
  [ ! -f $1 ] || [ $2 -nt $1 ] || [ common.hpp -nt $1 ] || [ igemm_common.hpp -nt $1 ] || [ ../../include/gcc_hip.h -nt $1 ]
}
pids=()
for f in $SRCS; do
  if stale build/$f.o $f.hip; then
    rm -f build/$f.o
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
if [ "${GCC_BUILD_DIAG:-1}" = "1" ]; then
  for f in $DIAG_SRCS; do
    if stale build/$f.diag.o $f.hip; then
      rm -f build/$f.diag.o
      hipcc $FLAGS -DGCC_DIAG_BUILD -c $f.hip -o build/$f.diag.o &
      pids+=($!)
    fi
  done
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
objs=""; dobjs=""
for f in $SRCS; do
  [ -f build/$f.o ] || { echo "compile of $f.hip failed" >&2; exit 1; }
  objs="$objs build/$f.o"
  if [[ " $DIAG_SRCS " == *" $f "* ]]; then dobjs="$dobjs build/$f.diag.o"; else dobjs="$dobjs build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs -ldl -lpthread
echo "built $(realpath $OUT)"
if [ "${GCC_BUILD_DIAG:-1}" = "1" ]; then
  for f in $DIAG_SRCS; do [ -f build/$f.diag.o ] || { echo "diagnostic compile of $f.hip failed" >&2; exit 1; }; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT_DIAG $dobjs -ldl -lpthread
  echo "built $(realpath $OUT_DIAG)"
fi
