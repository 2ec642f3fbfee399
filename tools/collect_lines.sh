#!/bin/bash
# every separately reported bench line on the shipped build, current protocol (4 frames in flight, GPU_MAX_HW_QUEUES=8): about 3 minutes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-lines}; mkdir -p $O
cd $R
python bench.py --amortised --no-cpu-baseline --no-side-arithmetics > $O/bench_amortised.json 2>/dev/null
for c in bf16x3 bf16x6 fp16x3 bf16x6-b8 fp16x3-b8 real real-b8 fp16x4-real fp16x4-real-b8 stress stress-b4 fp16x4-stress fp16x4-stress-b4; do
  steps=100; [[ $c == *b8* || $c == *stress* ]] && steps=20
  python bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline --no-side-arithmetics > $O/bench_config_$c.json 2>/dev/null
done
python bench.py --extractor --no-cpu-baseline > $O/spp_bench_extractor.json 2>/dev/null
python bench.py --extractor --extractor-precision fp16x4 --no-cpu-baseline > $O/spp_bench_extractor_fp16x4.json 2>/dev/null
python bench.py --pipeline --matcher-precision fp16x4 --extractor-precision fp16x4 > $O/pipeline_bench_fp16x4_both_stages.json 2>/dev/null
python bench.py --pnp > $O/pnp_bench.json 2>/dev/null
for f in $O/*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], d['metric'], d['value'], d['ms_per_step'], c.get('single_stream_frames_per_sec'), (d.get('parity_check') or {}).get('argmax_flips'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
