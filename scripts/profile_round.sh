#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the eager step, PMC traffic of the conv family, bench lines.
# usage (through gpurun): scripts/profile_round.sh <tag>      -> gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01_x}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graphs"
rm -rf /tmp/pk /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- $CMD > /tmp/pk.log 2>&1
f=$(find /tmp/pk -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD   (bf16, eager launches so that every kernel is visible)"; python $R/scripts/rocpd_summary.py $f 60; } > $OUT/${TAG}_kernel_stats.md
CMD1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graphs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o f -- $CMD1 > /tmp/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o w -- $CMD1 > /tmp/pw.log 2>&1
python $R/scripts/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) conv_igemm_kernel $OUT/${TAG}_pmc_igemm_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on \`$CMD1\` (bf16, batch 8)"
python $R/scripts/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) conv_wgrad $OUT/${TAG}_pmc_wgrad_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on \`$CMD1\` (bf16, batch 8)"
cd $R
# (the one compact line on stdout; the per-stage table and notes go to the detail file)
GS_BENCH_DETAIL=gpurun_out/${TAG}_bf16_bench_detail.json python bench.py > $OUT/${TAG}_bf16_bench.json 2> $OUT/${TAG}_bf16_bench.err
GS_BENCH_DETAIL=gpurun_out/${TAG}_f32_bench_detail.json python bench.py --dtype f32 --no-cpu-baseline > $OUT/${TAG}_f32_bench.json 2>> $OUT/${TAG}_bf16_bench.err
tail -c 600 $OUT/${TAG}_bf16_bench.json
