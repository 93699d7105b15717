#!/bin/bash
# PMC passes over the inverse spectral path (256 x (log-mel, IF) -> waveforms).  usage (gpurun): scripts/pmc_inverse.sh <tag>
# -> gpurun_out/<tag>_inverse_{pmc.txt,pmc_traffic.json}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_x}
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python $R/scripts/inverse_run.py 20"
rm -rf /tmp/ip1 /tmp/ip2 /tmp/if /tmp/iw
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $P1 -d /tmp/ip1 -o p -- $CMD > /tmp/ip1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d /tmp/ip2 -o p -- $CMD > /tmp/ip2.log 2>&1
python $R/scripts/pmc_table.py $(find /tmp/ip1 /tmp/ip2 -name "*.db") | grep -i "gemm\|inv_prep\|istft\|kernel" > $OUT/${TAG}_inverse_pmc.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/if -o f -- $CMD > /tmp/if.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/iw -o w -- $CMD > /tmp/iw.log 2>&1
python $R/scripts/pmc_traffic.py $(find /tmp/if -name "*.db" | head -1) $(find /tmp/iw -name "*.db" | head -1) gemm_bf16x6 $OUT/${TAG}_inverse_pmc_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on \`$CMD\` (256 examples)"
cat $OUT/${TAG}_inverse_pmc.txt
cat $OUT/${TAG}_inverse_pmc_traffic.json
