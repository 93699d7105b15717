#!/bin/bash
# burst-average time of every built igemm_trace_m<mode>_<x> variant on the conv layers of the fully grown networks (batch 8)
cd "$(dirname "$0")"
export GS_TRACE_QUIET=1
run() { # mode, layer
  echo "== mode $1 layer N H W IC OC = $2"
  for b in igemm_trace_def igemm_trace_m$1_?; do
    [ -x $b ] || continue
    printf "%-20s " $b; timeout 60 ./$b $1 $2 10 2>&1 | head -1 | sed 's/; kernel (first.*mean block/ mb/'
  done
}
for L in "8 128 1024 32 32" "8 64 512 64 64" "8 32 256 128 128" "8 16 128 256 256" "8 8 64 256 256" "8 4 32 256 256" "8 2 16 256 256"; do run 0 "$L"; done
for L in "8 128 1024 32 64" "8 64 512 64 128" "8 32 256 128 256" "8 16 128 256 256" "8 8 64 256 256" "8 4 32 256 256"; do run 1 "$L"; done
for L in "8 64 512 64 32" "8 32 256 128 64" "8 16 128 256 128" "8 8 64 256 256" "8 4 32 256 256" "8 2 16 256 256"; do run 2 "$L"; done
