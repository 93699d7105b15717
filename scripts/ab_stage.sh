#!/bin/bash
# A/B of library builds ab/lib_<tag>.so inside one gpurun call: bench value + per-stage times of the small layers.
# usage: scripts/ab_stage.sh <tag> [<tag> ...]   (the LAST tag stays installed)
for round in 1 2; do
for v in "$@"; do
  cp ab/lib_$v.so gansynth_amd/libgansynth_hip.so
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
small=sum(r['launches_per_iteration']*r['avg_us'] for r in d['stages'] if not r['stage'].startswith('wgrad') and any(('@ %s x' % s) in r['stage'] for s in ('2x16','4x32','8x64')))
allc=sum(r['launches_per_iteration']*r['avg_us'] for r in d['stages'] if not r['stage'].startswith('wgrad'))
wg=sum(r['launches_per_iteration']*r['avg_us'] for r in d['stages'] if r['stage'].startswith('wgrad'))
print('$v: %.1f img/s %.3f ms/step | igemm small layers %.0f us, all igemm %.0f us, wgrad %.0f us per iteration (eager)' % (d['value'], d['ms_per_step'], small, allc, wg))"
done; done
