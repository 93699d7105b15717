#!/bin/bash
# PMC passes over the conv micro-benchmark.  usage: scripts/pmc_conv.sh <dtype> <layer-filter> <kernel-grep>
# Runs on the GPU box (gpurun); each pass is its own rocprofv3 run (SQ has 8 slots per pass).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc$i -o p -- python $R/scripts/bench_conv.py $1 "$2" 5 > /tmp/pmc$i.log 2>&1
done
python $R/scripts/pmc_table.py $(find /tmp/pmc1 /tmp/pmc2 -name "*.db") | grep "$3"
