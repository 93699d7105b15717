#!/bin/bash
cd "$(dirname "$0")"
export GS_TRACE_QUIET=1
run() { echo "== mode $1 layer N H W IC OC = $2"; for b in igemm_trace_def igemm_trace_m$1_*; do [ -x $b ] || continue; printf "%-28s " $b; timeout 60 ./$b $1 $2 10 2>&1 | head -1 | cut -c1-110; done; }
run 0 "8 128 1024 32 32"
run 2 "8 64 512 64 32"
run 1 "8 128 1024 32 64"
