#!/bin/bash
# same-box comparison of several environment settings on the whole step: ab_multi.sh reps "VAR=1 OTHER=2" "VAR=3" ...   ("-" = no setting)
R=$1; shift
for i in $(seq $R); do
  for V in "$@"; do
    E="$V"; [ "$V" = "-" ] && E=""
    echo "[$V]: $(env $E python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4))')"
  done
done
