#!/bin/bash
# round 5, call 5: the in-prologue InstanceNorm merge of small segments (40 launches per frame at 500 x 2000): parity, bit-identity, A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "schedules or trained or database_cache or benchmarked_shapes or small or mid or random_shapes or attention_layer_stage or instance_norm" 2>&1 | tail -8 > $O/pytest_subset.log
tail -4 $O/pytest_subset.log
timeout 300 python tools/ab_live.py --config real --kernel mlp3 --rounds 8 --steps 40 "" STAT_PROLOGUE=0 > $O/ab_real_stat_prologue.txt 2>&1; tail -3 $O/ab_real_stat_prologue.txt
timeout 300 python tools/ab_live.py --config fp16x4-real --kernel mlp3 --rounds 8 --steps 40 "" STAT_PROLOGUE=0 > $O/ab_fp16x4_real_stat_prologue.txt 2>&1; tail -3 $O/ab_fp16x4_real_stat_prologue.txt
timeout 300 python tools/ab_live.py --config real-b8 --kernel mlp3 --rounds 4 --steps 10 "" STAT_PROLOGUE=0 > $O/ab_real_b8_stat_prologue.txt 2>&1; tail -3 $O/ab_real_b8_stat_prologue.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_real -o r -- python $R/bench.py --config real --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics --streams 1 > $O/prof_real.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_real -name "*.db" | head -1) > $O/kernel_stats_real.txt 2>&1; find $O -name "*.db" -delete
head -16 $O/kernel_stats_real.txt | cut -c1-44,75-140
