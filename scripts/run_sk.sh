# A/B of the grouped (stream-K) weight gradients inside one gpurun call: GS_NO_WGRAD_GROUPS, GS_SK_TW16, GS_SK_UNITS_PER_BLOCK[_S2]
for r in 1 2; do for v in "GS_X=0" "GS_NO_WGRAD_GROUPS=1" "GS_SK_TW16=1" "GS_SK_UNITS_PER_BLOCK_S2=2" "GS_SK_UNITS_PER_BLOCK_S2=8" "GS_SK_UNITS_PER_BLOCK=1" "GS_SK_UNITS_PER_BLOCK=4"; do
 env $v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-spectral --no-launch-count --no-f32-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
wg=sum(r['launches_per_iteration']*r['avg_us'] for r in d['stages'] if r['stage'].startswith('wgrad'))
nl=sum(r['launches_per_iteration'] for r in d['stages'] if r['stage'].startswith('wgrad'))
print('$v %.1f img/s %.3f ms/step | wgrad kernels %.0f us in %.0f launches per iteration (eager)' % (d['value'], d['ms_per_step'], wg, nl))
if $r == 2 and '$v' == 'GS_X=0':
    for r in d['stages']:
        if r['stage'].startswith('wgrad'): print('   %-90s x%.0f %7.1f us  %6.1f TF/s' % (r['stage'], r['launches_per_iteration'], r['avg_us'], r['tflops']))
" ; done; done 2>&1 | tee gpurun_out/sk_ab.txt
