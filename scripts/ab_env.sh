#!/bin/bash
# same-box A/B of an environment switch on the whole step: ab_env.sh "VAR=1" [reps]   (prints ms per step, alternating off / on)
V="$1"; R=${2:-3}
for i in $(seq $R); do
  for m in off on; do
    E=""; [ $m = on ] && E="$V"
    echo "$m ($V): $(env $E python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), round(d["value"],1))')"
  done
done
