#!/bin/bash
# Round-6 evidence collection on the GPU box:  tools/r06_collect.sh <part> [tag]
#   tests     pytest -m gpu + smoke()
#   lines     bench lines: headline (default + driver protocol), every BASELINE config, kernels with their own roofline, extractor / pipeline / pnp
#   prof      rocprofv3 kernel stats (one frame at a time, and 4 in flight for the headline) for headline, fp16x4 and the configs[2] / configs[4]
#             shapes (fp32-b8, fp16x4-b8, stress-b4, fp16x4-stress-b4)
#   pmc       FETCH_SIZE / WRITE_SIZE passes (separate runs) for the same six configs + the SQ pass of the headline -> profiles/pmc_traffic.json
#   eager     the stock PyTorch-ROCm baseline
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PART=${1:-tests}; TAG=${2:-r06}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
PROF="--reps 1 --min-timed-seconds 0 --no-cpu-baseline --no-side-arithmetics"
CONFIGS="headline fp16x4 fp32-b8 fp16x4-b8 stress-b4 fp16x4-stress-b4"
steps_of() { case $1 in *stress-b4) echo 6;; *b8) echo 10;; *) echo 50;; esac; }
case $PART in
tests)
  timeout 1700 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1
  tail -5 $O/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
  ;;
lines)
  python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
  python bench.py --steps 20 --warmup 5 > $O/bench_headline_driver_protocol.json 2>/dev/null
  python bench.py --amortised --no-cpu-baseline --no-side-arithmetics > $O/bench_amortised.json 2>/dev/null
  for c in fp16x4 fp16x4-b8 bf16x6 bf16x6-b8 fp16x3 bf16x3 fp32-b8 real real-b8 fp16x4-real fp16x4-real-b8 stress-b4 fp16x4-stress-b4 trained fp16x4-trained trained-hard; do
    steps=100; [[ $c == *b8* || $c == *stress* ]] && steps=20
    python bench.py --config $c --steps $steps --warmup 5 --no-cpu-baseline > $O/bench_config_$c.json 2>/dev/null
  done
  python bench.py --kernel gats --no-cpu-baseline --no-side-arithmetics > $O/bench_kernel_gats_hbm_roofline.json 2>/dev/null
  python bench.py --kernel conf_finalize --no-cpu-baseline --no-side-arithmetics > $O/bench_kernel_conf_finalize_hbm_roofline.json 2>/dev/null
  python bench.py --extractor --no-cpu-baseline > $O/spp_bench_extractor.json 2>/dev/null
  python bench.py --extractor --extractor-precision fp16x4 --no-cpu-baseline > $O/spp_bench_extractor_fp16x4.json 2>/dev/null
  python bench.py --pipeline --matcher-precision fp16x4 --extractor-precision fp16x4 > $O/pipeline_bench_fp16x4_both_stages.json 2>/dev/null
  python bench.py --pnp --no-cpu-baseline > $O/pnp_bench.json 2>/dev/null
  for f in $O/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
    print(f"{c.get('name', '?'):22s} value {d['value']:9.2f}  single {c.get('single_stream_frames_per_sec')}  module {c.get('module_forward_frames_per_sec')} / {c.get('module_forward_single_stream_frames_per_sec')}  roofline {d['roofline']['frac']}")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
  done
  ;;
prof)
  cd /tmp && export TMPDIR=/tmp
  for c in $CONFIGS; do
    st=$(steps_of $c)
    rocprofv3 --kernel-trace --stats -d $O/prof_$c -o r -- python $R/bench.py --config $c --steps $st --warmup 3 $PROF --streams 1 > $O/prof_$c.log 2>&1
    python $R/tools/rocpd_stats.py $(find $O/prof_$c -name "*.db" | head -1) > $O/kernel_stats_${c}_single_stream.txt 2>&1
  done
  rocprofv3 --kernel-trace --stats -d $O/prof_headline_s4 -o r -- python $R/bench.py --steps 50 --warmup 5 $PROF --streams 4 > $O/prof_headline_s4.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/prof_headline_s4 -name "*.db" | head -1) > $O/kernel_stats_headline_4_frames_in_flight.txt 2>&1
  find $O -name "*.db" -delete
  for c in $CONFIGS; do echo "== $c"; head -14 $O/kernel_stats_${c}_single_stream.txt | cut -c1-40,71-140; done
  ;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  HEAD=$(cat $R/gpurun_out/$TAG/head.txt 2>/dev/null)
  for c in $CONFIGS; do
    P="python $R/bench.py --config $c --steps 4 --warmup 2 $PROF --streams 1"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$c -o r -- $P > $O/pmc_fetch_$c.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$c -o r -- $P > $O/pmc_write_$c.log 2>&1
    for w in fetch write; do python $R/tools/rocpd_pmc.py $(find $O/pmc_${w}_$c -name "*.db" | head -1) > $O/pmc_${w}_$c.txt 2>&1; done
  done
  P="python $R/bench.py --steps 6 --warmup 2 $PROF --streams 1"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_sq -o r -- $P > $O/pmc_sq.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $O/pmc_sq -name "*.db" | head -1) > $O/pmc_sq_headline.txt 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_sq_s -o r -- $P --config fp16x4-stress-b4 --steps 3 > $O/pmc_sq_s.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $O/pmc_sq_s -name "*.db" | head -1) > $O/pmc_sq_fp16x4-stress-b4.txt 2>&1
  find $O -name "*.db" -delete
  # the traffic file: headline entries at the top level, the other configs under _configs
  cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
  python $R/tools/make_pmc_traffic.py $O/pmc_fetch_headline.txt $O/pmc_write_headline.txt --file $O/pmc_traffic.json > $O/pmc_traffic.new && mv $O/pmc_traffic.new $O/pmc_traffic.json
  for c in $CONFIGS; do
    [ $c = headline ] && continue
    python $R/tools/make_pmc_traffic.py $O/pmc_fetch_$c.txt $O/pmc_write_$c.txt --file $O/pmc_traffic.json --config $c > $O/pmc_traffic.new && mv $O/pmc_traffic.new $O/pmc_traffic.json
  done
  head -14 $O/pmc_fetch_stress-b4.txt | cut -c1-100
  ;;
eager)
  python bench.py --torch-eager --steps 30 > $O/torch_eager_baseline.json 2>/dev/null
  python bench.py --torch-eager --extractor --steps 30 > $O/torch_eager_baseline_extractor.json 2>/dev/null
  cat $O/torch_eager_baseline.json
  ;;
esac
