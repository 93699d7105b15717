# kernel durations of the inverse path (rocprofv3 --kernel-trace --stats over scripts/inverse_run.py); usage: prof_inverse.sh [ENV=1 ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pi
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/pi -o d -- python $R/scripts/inverse_run.py 200 > /tmp/pi.log 2>&1
tail -1 /tmp/pi.log
python $R/scripts/rocpd_summary.py $(find /tmp/pi -name "*.db" | head -1) 8
