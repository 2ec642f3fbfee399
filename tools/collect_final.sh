#!/bin/bash
# Trimmed end-of-round collection (about 8 minutes of box time): the GPU suite + smoke on the build that ships, the two headline lines,
# the fp16x4 lines, rocprofv3 kernel stats one frame at a time for fp32 and fp16x4.  tools/collect_profiles.sh is the full (25 min) set.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04f}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
git rev-parse HEAD > $O/head.txt 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider 2>&1 | tail -5 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --steps 20 --warmup 5 > $O/bench_headline_driver_protocol.json 2>/dev/null
for c in fp16x4 fp16x4-b8 fp32-b8; do
  steps=100; [[ $c == *b8* ]] && steps=20
  python bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline > $O/bench_$c.json 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics --streams 1"
rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- $B > $O/prof_s1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_fp16x4 -o r -- $B --config fp16x4 > $O/prof_fp16x4.log 2>&1
for d in prof_s1 prof_fp16x4; do python $R/tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/kernel_stats_$d.txt 2>&1; done
find $O -name "*.db" -delete
cat $O/pytest_gpu.log | head -8
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], d['value'], c.get('single_stream_frames_per_sec'), d['roofline']['frac'], d.get('parity_check',{}).get('argmax_flips'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
head -14 $O/kernel_stats_prof_s1.txt | cut -c1-44,75-140
head -14 $O/kernel_stats_prof_fp16x4.txt | cut -c1-44,75-140
