python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -2
for i in 1 2; do
for v in old new; do cp ab/lib_$v.so gansynth_amd/libgansynth_hip.so; echo "$v: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])')"; done; done
cp ab/lib_new.so gansynth_amd/libgansynth_hip.so
