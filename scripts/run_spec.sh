# A/B of igemm dispatch knobs inside one gpurun call: GS_SPEC (wave-specialised families), GS_NO_RB128
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k conv 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "full_size" 2>&1 | tail -2
for r in 1 2; do for v in "GS_SPEC=1" "GS_SPEC=0" "GS_NO_RB128=1"; do
 env $v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
small=sum(r['launches_per_iteration']*r['avg_us'] for r in d['stages'] if not r['stage'].startswith('wgrad') and any(('@ %s x' % s) in r['stage'] for s in ('2x16','4x32','8x64')))
allc=sum(r['launches_per_iteration']*r['avg_us'] for r in d['stages'] if not r['stage'].startswith('wgrad'))
print('$v %.1f img/s %.3f ms/step | small %.0f us, all igemm %.0f us' % (d['value'], d['ms_per_step'], small, allc))
" ; done; done 2>&1 | tee gpurun_out/spec_ab.txt
