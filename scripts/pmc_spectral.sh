#!/bin/bash
# PMC passes + kernel stats over the spectral front end (configs[3]).  usage (gpurun): scripts/pmc_spectral.sh <tag> [dtype]
# -> gpurun_out/<tag>_spectral_{kernel_stats.md,pmc.txt,pmc_traffic.json}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_x}
DT=${2:-f32}
OUT=$R/gpurun_out
mkdir -p $OUT
CMD="python $R/scripts/spectral_run.py 10 $DT"
rm -rf /tmp/sk /tmp/sp1 /tmp/sp2 /tmp/sf /tmp/sw
rocprofv3 --kernel-trace --stats -d /tmp/sk -o k -- $CMD > /tmp/sk.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD"; tail -1 /tmp/sk.log; python $R/scripts/rocpd_summary.py $(find /tmp/sk -name "*.db" | head -1) 12; } > $OUT/${TAG}_spectral_kernel_stats.md
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $P1 -d /tmp/sp1 -o p -- $CMD > /tmp/sp1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d /tmp/sp2 -o p -- $CMD > /tmp/sp2.log 2>&1
python $R/scripts/pmc_table.py $(find /tmp/sp1 /tmp/sp2 -name "*.db") | grep -i "stft\|unwrap" > $OUT/${TAG}_spectral_pmc.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/sf -o f -- $CMD > /tmp/sf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/sw -o w -- $CMD > /tmp/sw.log 2>&1
python $R/scripts/pmc_traffic.py $(find /tmp/sf -name "*.db" | head -1) $(find /tmp/sw -name "*.db" | head -1) stft $OUT/${TAG}_spectral_pmc_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on \`$CMD\` (batch 256 x 64000 samples)"
cat $OUT/${TAG}_spectral_kernel_stats.md | head -12
cat $OUT/${TAG}_spectral_pmc.txt
