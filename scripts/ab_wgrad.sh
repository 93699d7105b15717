#!/bin/bash
# A/B of ab/lib_<tag>.so variants on the grouped weight gradients (scripts/bench_wgrad_group.py), same box.  usage: ab_wgrad.sh tag ...
cp gansynth_amd/libgansynth_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for t in "$@"; do
  cp ab/lib_$t.so gansynth_amd/libgansynth_hip.so
  for st in 1 2; do for n in 16 24; do echo "$t: $(python scripts/bench_wgrad_group.py $n $st 2>/dev/null | tail -1)"; done; done
done; done
cp /tmp/lib_keep.so gansynth_amd/libgansynth_hip.so
