#!/bin/bash
# round 5, call 4: wave / workgroup traces of the new and old mlp.0 epilogues (profiling build), A/B with the stat_final geometry fixed,
# the GATs softmax on 36 lanes, bf16x6 frames in flight
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "gats_layer_stage or attention_layer_stage or trained or database_cache or benchmarked_shapes or small or headline" 2>&1 | tail -6 > $O/pytest_subset.log
tail -3 $O/pytest_subset.log
GATSSPG_SP_SCHED=4 timeout 120 python tools/trace_sp.py fp16x4 > $O/trace_sp_ut1.txt 2>&1; tail -12 $O/trace_sp_ut1.txt
GATSSPG_SP_SCHED=4 GATSSPG_SP_UT=0 timeout 120 python tools/trace_sp.py fp16x4 > $O/trace_sp_ut0.txt 2>&1; tail -12 $O/trace_sp_ut0.txt
timeout 120 python tools/trace_mlp0.py > $O/trace_mlp0_fp32.txt 2>&1; tail -22 $O/trace_mlp0_fp32.txt
timeout 300 python tools/ab_live.py --config fp16x4 --kernel mlp0 --rounds 6 --steps 30 "" SP_UT=0 > $O/ab_fp16x4_ut.txt 2>&1; tail -3 $O/ab_fp16x4_ut.txt
timeout 300 python tools/ab_live.py --config fp16x4 --kernel stat_final --rounds 4 --steps 30 "" SP_UT=0 > $O/ab_fp16x4_ut_stat.txt 2>&1; tail -3 $O/ab_fp16x4_ut_stat.txt
timeout 300 python tools/ab_live.py --config fp16x4-real --kernel mlp0 --rounds 6 --steps 40 "" SP_UT=0 > $O/ab_fp16x4_real_ut.txt 2>&1; tail -3 $O/ab_fp16x4_real_ut.txt
timeout 300 python tools/ab_live.py --config headline --kernel gats --rounds 3 --steps 30 "" > $O/ab_gats.txt 2>&1; tail -2 $O/ab_gats.txt
for st in 3 4; do timeout 200 python bench.py --config bf16x6 --streams $st --steps 150 --reps 3 --no-cpu-baseline > $O/bench_bf16x6_s$st.json 2>/dev/null; python - $O/bench_bf16x6_s$st.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['config']['single_stream_frames_per_sec'])
PY
done
