#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/sweep_streams.log
: > $out
for s in 1 2 3 4 6 8; do
  echo "== streams=$s" >> $out
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --streams $s 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],'fps', d['ms_per_step'],'ms/step; latency', d['config']['single_frame_latency_ms'],'ms;', d['roofline']['kernel'], d['roofline']['kernel_ms'],'ms', d['roofline']['achieved'],'TF')" >> $out
done
cat $out
