#!/bin/bash
# rocprofv3 kernel stats of one bench config, one frame at a time:  tools/prof_config.sh <tag> <config> [more bench flags]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
C=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_$C -o r -- python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics --streams 1 --config $C "$@" > $O/prof_$C.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_$C -name "*.db" | head -1) > $O/kernel_stats_$C.txt 2>&1
find $O -name "*.db" -size +20M -delete
grep "^{" $O/prof_$C.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['config']['single_frame_latency_ms'])"
head -16 $O/kernel_stats_$C.txt | cut -c1-44,75-140
