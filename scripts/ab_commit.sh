#!/bin/bash
# Same-box A/B of the working tree against a COMMIT (default HEAD~1): a git worktree of that commit is built beside the tree (ab_prev/, git-ignored,
# travels with the gpurun snapshot) and both benches run in ONE gpurun call.  An A/B by environment switch inside one build says nothing about what a
# change did to kernels the switch does not touch (round 3: an epilogue variant pushed another instantiation into scratch, +0.6 ms, invisible to the
# switch).  usage (here, not on the GPU box): scripts/ab_commit.sh [commit]   -> prints prev / new ms per step, twice
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=${1:-HEAD~1}
cd $R
git worktree remove --force ab_prev 2>/dev/null || true
git worktree add -f ab_prev $C > /dev/null
(cd ab_prev/gansynth_amd/csrc && bash build.sh | tail -1)
B='python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-spectral 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"])"'
/usr/local/graft/bin/gpurun --timeout 1800 -- "for i in 1 2; do echo -n 'prev '; (cd ab_prev && $B); echo -n 'new  '; $B; done" 2>&1 | tail -5
git worktree remove --force ab_prev
