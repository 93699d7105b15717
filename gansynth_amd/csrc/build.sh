#!/bin/bash
# Builds libgansynth_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [--clean] [outdir]
#   --clean   drop every object first (obj/ is git-ignored but travels with the tree: without the flag the build is incremental by mtime)
set -e
cd "$(dirname "$0")"
if [ "$1" = "--clean" ]; then
  rm -rf obj
  shift
fi
OUT=${1:-..}
mkdir -p obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $GS_EXTRA_FLAGS"
pids=()
for f in conv_igemm conv_api elementwise small_ops spectral spectral_wave; do
  [ -f $f.hip ] || continue
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ gs_common.h -nt obj/$f.o ] || [ gs_prof.h -nt obj/$f.o ] || [ conv_shared.h -nt obj/$f.o ] || [ spectral_plan.h -nt obj/$f.o ] || [ ../../include/gansynth_hip.h -nt obj/$f.o ]; then
    EXTRA=""
    # MFMA accumulators in VGPRs (hipcc otherwise parks them in AGPRs and every epilogue value costs a v_accvgpr_read)
    [ $f = conv_igemm ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form $GS_IGEMM_FLAGS"
    # (complex arithmetic as float2: SLP-packing it into v_pk_* costs more register shuffling than it saves -- measured -8 %)
    [ $f = spectral_wave ] && EXTRA="-fno-slp-vectorize $GS_SW_FLAGS"
    ( hipcc $FLAGS $EXTRA -c $f.hip -o obj/$f.o ) &
    pids+=($!)
  fi
done
if [ ! -f obj/core.o ] || [ core.cpp -nt obj/core.o ] || [ gs_common.h -nt obj/core.o ] || [ gs_prof.h -nt obj/core.o ]; then
  ( hipcc $FLAGS -x hip -c core.cpp -o obj/core.o ) &
  pids+=($!)
fi
if [ ! -f obj/comm.o ] || [ comm.cpp -nt obj/comm.o ] || [ gs_common.h -nt obj/comm.o ] || [ ../../include/gansynth_hip.h -nt obj/comm.o ]; then
  ( hipcc $FLAGS -x hip -c comm.cpp -o obj/comm.o ) &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/*.o -ldl -o $OUT/libgansynth_hip.so
echo "built $OUT/libgansynth_hip.so"
