#!/bin/bash
# builds igemm_trace_<tag> probes with forced tile configurations / ablations (see conv_igemm.hip GS_FORCE_CFG, GS_ABL_*)
# usage: build_variants.sh  "tag|mode|cfg|extra flags" ...
cd "$(dirname "$0")"
pids=()
for v in "$@"; do
  IFS='|' read -r tag mode cfg extra <<< "$v"
  F=""
  [ -n "$cfg" ] && F="-DGS_FORCE_MODE=$mode -DGS_FORCE_CFG=$cfg"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I../../include -I../../gansynth_amd/csrc $F $extra -o igemm_trace_$tag igemm_trace.hip 2>&1 | grep -E "error" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls igemm_trace_*
