#!/bin/bash
# A/B runs of tuning knobs on the GPU box (one process per variant; same inputs).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/tune_tiles.log
: > $out
run() { echo "== K=$K $*" >> $out; env "$@" python bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline --kernel "$K" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],'fps', d['ms_per_step'],'ms', d['roofline']['kernel'], d['roofline']['kernel_ms'],'ms', d['roofline']['achieved'],'TF')" >> $out; }
for v in "$@"; do K=${K:-mlp3} run $v; done
cat $out
