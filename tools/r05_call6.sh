#!/bin/bash
# round 5, call 6: energy-lean tile variants re-measured under the in-flight (power-capped) protocol
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout 400 python tools/ab_live.py --config fp16x4 --kernel mlp0 --rounds 6 --steps 40 "" SP_ABL=100 SP_MLP0_WIDE_MIN=100000 SP_NST2=2 > $O/ab_fp16x4_tiles_inflight.txt 2>&1; tail -6 $O/ab_fp16x4_tiles_inflight.txt
timeout 400 python tools/ab_live.py --config fp16x4-b8 --kernel mlp0 --rounds 4 --steps 10 "" SP_ABL=100 SP_MLP0_WIDE_MIN=0,SP_MLP0_WIDE_MAX=100000 > $O/ab_fp16x4_b8_tiles_inflight.txt 2>&1; tail -5 $O/ab_fp16x4_b8_tiles_inflight.txt
