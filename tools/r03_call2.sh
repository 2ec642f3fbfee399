#!/bin/bash
# round-3 GPU call: parity tests of the matcher, bench headline (with the two split arithmetics), rocprofv3 kernel stats one frame at a time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r03b}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_module_api.py -m gpu -q --timeout=600 -p no:cacheprovider -x 2>&1 | tail -25 > $O/pytest_parity.log
tail -5 $O/pytest_parity.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_headline.json 2> $O/bench_headline.err
python - $O/bench_headline.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print("headline", d["value"], c["single_stream_frames_per_sec"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d.get("parity_check"))
    print(json.dumps(c.get("other_gemm_arithmetics")))
except Exception as e: print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics"
rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- $B --streams 1 > $O/prof_s1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_b6 -o r -- $B --streams 1 --config bf16x6 > $O/prof_b6.log 2>&1
for d in prof_s1 prof_b6; do python $R/tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/kernel_stats_$d.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
cat $O/kernel_stats_prof_s1.txt | head -30
cat $O/kernel_stats_prof_b6.txt | head -12
