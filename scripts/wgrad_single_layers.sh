#!/bin/bash
# Does the L2 carry the re-reads of the grouped weight-gradient kernel when the blocks that share operand tiles are aligned and on one XCD?
# A single-layer "group" is naturally so (256 blocks, every channel-tile run a whole number of blocks, runs a multiple of 8 blocks apart).
for L in "256,256,8,64" "128,256,16,128" "64,128,32,256" "256,256,4,32"; do python scripts/bench_wgrad_group.py 16 2 "$L" 2>/dev/null | tail -1; done
for L in "256,256,16,128" "128,128,32,256" "64,64,64,512" "256,256,8,64"; do python scripts/bench_wgrad_group.py 16 1 "$L" 2>/dev/null | tail -1; done
python scripts/bench_wgrad_group.py 16 2 2>/dev/null | tail -1
python scripts/bench_wgrad_group.py 16 1 2>/dev/null | tail -1
