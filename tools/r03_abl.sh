#!/bin/bash
# kv_final store-policy / ablation A-B in the tuning build: event-timed kernel + one-frame-at-a-time rate
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r03d}; mkdir -p $O
cd $R
for a in ${2:-0 8 16 32 0 8}; do
  GATSSPG_KVF_ABL=$a python bench.py --tuning-lib --kernel kv_final --steps 60 --warmup 10 --reps 3 --no-side-arithmetics --no-cpu-baseline > $O/abl_$a.json 2>/dev/null
  python -c "
import json,sys
d=json.load(open('$O/abl_$a.json')); print('KVF_ABL=$a kernel_ms', d['roofline']['kernel_ms'], 'pair', d['roofline']['empty_event_pair_ms'], 'single', d['config']['single_stream_frames_per_sec'], 'inflight', d['value'], 'flips', d['parity_check']['argmax_flips'])"
done
