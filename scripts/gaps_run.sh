cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/pg -o g -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/pg.log 2>&1
tail -2 /tmp/pg.log | cut -c1-300
f=$(find /tmp/pg -name "*.db" | head -1)
python $R/scripts/rocpd_gaps.py $f 0.4
rocprofv3 --kernel-trace -d /tmp/pe -o e -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graphs > /tmp/pe.log 2>&1
tail -2 /tmp/pe.log | cut -c1-300
f=$(find /tmp/pe -name "*.db" | head -1)
python $R/scripts/rocpd_gaps.py $f 0.4
