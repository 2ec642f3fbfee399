#!/bin/bash
# round-3 GPU call: parity tests, bench headline, kv_final ablations (tuning lib), rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r03c}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=600 -p no:cacheprovider -x 2>&1 | tail -8 > $O/pytest_parity.log
tail -3 $O/pytest_parity.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_headline.json 2> $O/bench_headline.err
python - $O/bench_headline.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print("headline", d["value"], c["single_stream_frames_per_sec"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d.get("parity_check"))
    for k,v in c.get("other_gemm_arithmetics",{}).items():
        if isinstance(v,dict): print(k, v["frames_per_sec"], v["single_stream_frames_per_sec"], v["argmax_flips_vs_reference_golden"], v["max_abs_conf_err_vs_reference_golden"])
except Exception as e: print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
for a in 0 1 2 4 3 7; do
  GATSSPG_KVF_ABL=$a python bench.py --tuning-lib --kernel kv_final --steps 30 --warmup 5 --reps 1 --no-side-arithmetics --no-cpu-baseline > $O/abl_$a.json 2>/dev/null
  python -c "
import json,sys
d=json.load(open('$O/abl_$a.json')); print('KVF_ABL=$a kernel_ms', d['roofline']['kernel_ms'], 'pair', d['roofline']['empty_event_pair_ms'], 'single', d['config']['single_stream_frames_per_sec'])"
done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics"
rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- $B --streams 1 > $O/prof_s1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_b6 -o r -- $B --streams 1 --config bf16x6 > $O/prof_b6.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_b3 -o r -- $B --streams 1 --config bf16x3 > $O/prof_b3.log 2>&1
for d in prof_s1 prof_b6 prof_b3; do python $R/tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/kernel_stats_$d.txt 2>&1; done
find $O -name "*.db" -size +20M -delete
head -13 $O/kernel_stats_prof_s1.txt | cut -c1-40,75-140
head -7 $O/kernel_stats_prof_b6.txt | cut -c1-40,75-140
head -7 $O/kernel_stats_prof_b3.txt | cut -c1-40,75-140
