#!/bin/bash
# SQ PMC passes over one conv layer run by a probe binary (scripts/probe/igemm_trace_*).
# usage (gpurun): scripts/pmc_probe.sh <out file> <probe binary> "<mode N H W IC OC>" ["<layer>" ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
BIN=$R/scripts/probe/$1; shift
mkdir -p $R/gpurun_out
: > $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for L in "$@"; do
  rm -rf /tmp/pp1 /tmp/pp2
  GS_TRACE_QUIET=1 rocprofv3 --kernel-trace --pmc $P1 -d /tmp/pp1 -o p -- $BIN $L 3 > /tmp/pp1.log 2>&1
  GS_TRACE_QUIET=1 rocprofv3 --kernel-trace --pmc $P2 -d /tmp/pp2 -o p -- $BIN $L 3 > /tmp/pp2.log 2>&1
  echo "== $L" >> $OUT
  python $R/scripts/pmc_table.py $(find /tmp/pp1 /tmp/pp2 -name "*.db") >> $OUT 2>&1
done
cat $OUT
