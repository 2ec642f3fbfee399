#!/bin/bash
# round-3 GPU call 1: full GPU suite (new: torchrun world-1 RCCL path, stress-b4 / real goldens), smoke, bench lines of the new configs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r03a}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x 2>&1 | tail -15 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke > $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_headline_driver_protocol.json 2> $O/bench_headline.err
for c in real real-b8 stress-b4 bf16x6-b8; do
  python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_$c.json 2>/dev/null
done
tail -3 $O/pytest_gpu.log; cat $O/smoke.log; for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(c.get("name"), d["value"], c.get("single_stream_frames_per_sec"), d["roofline"]["frac"], d.get("parity_check",{}).get("argmax_flips"), d.get("parity_check",{}).get("max_abs_conf_err"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
