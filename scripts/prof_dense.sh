# kernel durations of the dense kernels at the model's shapes (rocprofv3 --kernel-trace --stats over scripts/bench_dense.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pd
rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python $R/scripts/bench_dense.py > /tmp/pd.log 2>&1
python $R/scripts/rocpd_summary.py $(find /tmp/pd -name "*.db" | head -1) 16 | grep -E "dense|kernel \|"
