#!/bin/bash
# burst-average time of the named probe variants per mode on the mid conv layers.  usage: run_mid.sh  (variants igemm_trace_m<mode>_*)
cd "$(dirname "$0")"
export GS_TRACE_QUIET=1
run() { echo "== mode $1 layer $2"; for b in igemm_trace_def igemm_trace_m$1_*; do [ -x $b ] || continue; printf "%-20s " $b; timeout 60 ./$b $1 $2 10 2>&1 | head -1 | sed 's/; kernel (first.*mean block/ mb/'; done; }
for L in "8 64 512 64 64" "8 32 256 128 128" "8 16 128 256 256"; do run 0 "$L"; done
for L in "8 64 512 64 128" "8 32 256 128 256"; do run 1 "$L"; done
for L in "8 32 256 128 64" "8 16 128 256 128"; do run 2 "$L"; done
