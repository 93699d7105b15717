#!/bin/bash
# SQ PMC passes over the grouped (stream-K) weight-gradient kernels: scripts/bench_wgrad_group.py <images> <stride>.
# usage (gpurun): scripts/pmc_wgrad_group.sh <out file under gpurun_out> [images]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-wgrad_group_sq_pmc.txt}
N=${2:-16}
mkdir -p $R/gpurun_out
: > $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for ST in 1 2; do
  CMD="python $R/scripts/bench_wgrad_group.py $N $ST"
  rm -rf /tmp/wp1 /tmp/wp2
  rocprofv3 --kernel-trace --pmc $P1 -d /tmp/wp1 -o p -- $CMD > /tmp/wp1.log 2>&1
  rocprofv3 --kernel-trace --pmc $P2 -d /tmp/wp2 -o p -- $CMD > /tmp/wp2.log 2>&1
  echo "== stride $ST, $N images: $(tail -1 /tmp/wp1.log)" >> $OUT
  python $R/scripts/pmc_table.py $(find /tmp/wp1 /tmp/wp2 -name "*.db") | grep -i "wgrad" >> $OUT 2>&1
done
cat $OUT
