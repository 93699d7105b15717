#!/bin/bash
# builds ab/lib_<tag>.so variants of the library that differ in conv_igemm.hip compile flags (A/B measurements inside one gpurun call).
# usage: scripts/build_ab.sh "tag|flags" ...      (ab/ is git-ignored; delete it when done: it ships with every gpurun call)
cd "$(dirname "$0")/../gansynth_amd/csrc"
mkdir -p ../../ab
./build.sh > /dev/null
pids=()
for v in "$@"; do
  IFS='|' read -r tag flags <<< "$v"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $flags -c conv_igemm.hip -o ../../ab/igemm_$tag.o 2>/dev/null \
    && hipcc --offload-arch=gfx950 -shared -fPIC ../../ab/igemm_$tag.o $(ls obj/*.o | grep -v conv_igemm.o) -ldl -o ../../ab/lib_$tag.so && rm ../../ab/igemm_$tag.o && echo built $tag ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la ../../ab
