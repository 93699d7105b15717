#!/bin/bash
# burst-average time of the given probe variants on the mid / top conv layers.  usage: run_set.sh <variant> [<variant> ...]
cd "$(dirname "$0")"
export GS_TRACE_QUIET=1
for L in "0 8 16 128 256 256" "0 8 32 256 128 128" "0 8 64 512 64 64" "0 8 128 1024 32 32" "1 8 32 256 128 256" "1 8 64 512 64 128" "2 8 16 128 256 128" "2 8 32 256 128 64" "0 8 8 64 256 256" "0 8 2 16 256 256"; do
  echo "== $L"
  for b in "$@"; do printf "%-10s " $b; timeout 60 ./igemm_trace_$b $L 10 2>&1 | head -1; done
done
