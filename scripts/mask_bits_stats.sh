R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for V in bits nobits; do
  E=""; [ $V = nobits ] && E="GS_NO_MASK_BITS=1"
  rm -rf /tmp/pk_$V
  env $E rocprofv3 --kernel-trace --stats -d /tmp/pk_$V -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graphs --no-spectral --no-f32-leg > /tmp/pk.log 2>&1
  f=$(find /tmp/pk_$V -name "*.db" | head -1)
  python $R/scripts/rocpd_summary.py $f 40 > $R/gpurun_out/mb_${V}_stats.md
done
