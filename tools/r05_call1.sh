#!/bin/bash
# round 5, call 1: the full GPU suite on the ABI-410 build (trained-weights goldens, data-bounded operator scale) + the headline line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x 2>&1 | tail -30 > $O/pytest_gpu.log
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -s -k "trained or message_operator" 2>&1 | grep -v "^$" | tail -80 > $O/pytest_trained_verbose.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
timeout 300 python bench.py --config trained --no-cpu-baseline --steps 100 > $O/bench_trained.json 2>/dev/null
timeout 300 python bench.py --config fp16x4-trained --no-cpu-baseline --steps 100 > $O/bench_fp16x4_trained.json 2>/dev/null
tail -5 $O/pytest_gpu.log; tail -3 $O/smoke.log
