#!/bin/bash
# A/B of two library builds inside one gpurun call (box-to-box variance is larger than most single changes):
#   ab/lib_old.so, ab/lib_new.so -> bench lines alternately; the new build stays installed.  usage: ab_bench.sh [pytest -k filter] [extra script]
K=${1:-conv}
python -m pytest tests/test_kernels_gpu.py -x -q -k "$K" 2>&1 | tail -2
[ -n "$2" ] && for v in old new; do cp ab/lib_$v.so gansynth_amd/libgansynth_hip.so; echo "== $v"; PYTHONPATH=. python $2 2>&1 | grep -v amdgpu.ids | head -${3:-16}; done
for i in 1 2; do
for v in old new; do cp ab/lib_$v.so gansynth_amd/libgansynth_hip.so; echo "$v: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])')"; done; done
cp ab/lib_new.so gansynth_amd/libgansynth_hip.so
