#!/bin/bash
cd "$(dirname "$0")"
export GS_TRACE_QUIET=1
run() { echo "== mode $1 layer N H W IC OC = $2"; for b in igemm_trace_def igemm_trace_m$1_*; do [ -x $b ] || continue; printf "%-28s " $b; timeout 60 ./$b $1 $2 10 2>&1 | head -1 | cut -c1-120; done; }
for L in "8 16 128 256 128" "8 8 64 256 256" "8 4 32 256 256" "8 2 16 256 256"; do run 2 "$L"; done
