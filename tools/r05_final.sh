#!/bin/bash
# End-of-round collection, part 1 (tests + bench lines) and part 2 (rocprofv3 kernel stats + PMC passes): tools/r05_final.sh <1|2> [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PART=${1:-1}; TAG=${2:-r05f}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
git rev-parse HEAD > $O/head.txt 2>/dev/null
if [ "$PART" = 1 ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -6 > $O/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
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
  cat $O/pytest_gpu.log | tail -12
else
  cd /tmp && export TMPDIR=/tmp
  B="python $R/bench.py --steps 50 --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics"
  rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o r -- $B --streams 1 > $O/prof_s1.log 2>&1
  rocprofv3 --kernel-trace --stats -d $O/prof_s4 -o r -- $B --streams 4 > $O/prof_s4.log 2>&1
  for c in fp16x4 fp16x4-b8 real; do
    st=50; [[ $c == *b8* ]] && st=10
    rocprofv3 --kernel-trace --stats -d $O/prof_$c -o r -- python $R/bench.py --steps $st --warmup 5 --reps 1 --no-cpu-baseline --no-side-arithmetics --streams 1 --config $c > $O/prof_$c.log 2>&1
  done
  P="python $R/bench.py --steps 6 --warmup 2 --reps 1 --streams 1 --no-cpu-baseline --no-side-arithmetics"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $P > $O/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- $P > $O/pmc_write.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_fp16x4 -o r -- $P --config fp16x4 > $O/pmc_fetch_fp16x4.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_fp16x4 -o r -- $P --config fp16x4 > $O/pmc_write_fp16x4.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc_sq_fp16x4 -o r -- $P --config fp16x4 > $O/pmc_sq_fp16x4.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_sq -o r -- $P > $O/pmc_sq.log 2>&1
  for d in prof_s1 prof_s4 prof_fp16x4 prof_fp16x4-b8 prof_real; do python $R/tools/rocpd_stats.py $(find $O/$d -name "*.db" | head -1) > $O/kernel_stats_$d.txt 2>&1; done
  for d in pmc_fetch pmc_write pmc_fetch_fp16x4 pmc_write_fp16x4 pmc_sq_fp16x4 pmc_sq; do python $R/tools/rocpd_pmc.py $(find $O/$d -name "*.db" | head -1) > $O/$d.txt 2>&1; done
  find $O -name "*.db" -delete
  head -16 $O/kernel_stats_prof_s1.txt | cut -c1-44,75-140
  head -16 $O/kernel_stats_prof_fp16x4.txt | cut -c1-44,75-140
  head -14 $O/pmc_fetch.txt | cut -c1-100
fi
