#!/bin/bash
# frames in flight (bench.py --streams) x the HIP runtime's hardware-queue pool (GPU_MAX_HW_QUEUES, default 4), one box, back to back
#   tools/sweep_queues.sh <tag> "<queue counts>" "<stream counts>" "<configs>"
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-sweepq}; mkdir -p $O
QS=${2:-"default 8"}; SS=${3:-"3 4 5 6"}; CS=${4:-"headline fp16x4 bf16x6 fp16x3 bf16x3"}   # all five arithmetics (round-4 judge, item 4)
out=$O/sweep_queues.txt; : > $out
for c in $CS; do
for q in $QS; do
for s in $SS; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --config $c --steps 150 --warmup 20 --reps 3 --no-cpu-baseline --no-side-arithmetics --streams $s 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config $c hwq $q streams $s:', d['value'], d['ms_per_step'], d['config'].get('single_stream_frames_per_sec'))" >> $out
done; done; done
cat $out
