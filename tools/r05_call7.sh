#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "gats_layer_stage or benchmarked_shapes or trained_weights_vs_reference or database_cache or small" 2>&1 | tail -4 > $O/pytest_subset.log; tail -2 $O/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
rocprofv3 --kernel-trace --stats -d $O/prof_s1_$i -o r -- python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics --streams 1 > $O/prof.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_s1_$i -name "*.db" | head -1) > $O/kernel_stats_s1_$i.txt 2>&1
grep -E "gats_leaf8x4|mlp0_kernel" $O/kernel_stats_s1_$i.txt | cut -c1-44,75-140
done
find $O -name "*.db" -delete
