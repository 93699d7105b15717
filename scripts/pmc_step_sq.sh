#!/bin/bash
# SQ counters of EVERY kernel of one eager training iteration (two rocprofv3 --pmc passes over bench.py, kernel-trace only): per kernel
# instantiation and grid -- MFMA pipe busy, waves active / waiting / issue-stalled, instruction mix, LDS cycles and conflicts.
# usage (gpurun): scripts/pmc_step_sq.sh <out file under gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-step_sq_pmc.txt}
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graphs --no-spectral --no-launch-count"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rm -rf /tmp/sq1 /tmp/sq2
rocprofv3 --kernel-trace --pmc $P1 -d /tmp/sq1 -o p -- $CMD > /tmp/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d /tmp/sq2 -o p -- $CMD > /tmp/sq2.log 2>&1
{ echo "# SQ counters per kernel instantiation and grid over one eager iteration (+ warm-up / profiling iterations) of: $CMD"; python $R/scripts/pmc_table.py $(find /tmp/sq1 /tmp/sq2 -name "*.db"); } > $OUT
grep -c . $OUT
