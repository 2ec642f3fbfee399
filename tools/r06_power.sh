#!/bin/bash
# round 6 (same protocol as round 5): what the power manager does under the matcher's kernels -- rocm-smi socket power / clocks sampled while bench.py runs
# (usage: tools/r06_power.sh; writes gpurun_out/r06p/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06p; mkdir -p $O
cd $R
(rocm-smi --showmaxpower; rocm-smi --showpower --showclocks --showperflevel; amd-smi static --limit 2>/dev/null | head -30) > $O/smi_idle.txt 2>&1
sample() {   # $1 tag, rest: bench flags
  tag=$1; shift
  python bench.py "$@" --warmup 20 --reps 2 --no-cpu-baseline --no-side-arithmetics > $O/bench_$tag.json 2>/dev/null &
  BP=$!
  : > $O/smi_$tag.txt
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | tr '\n' ' ' >> $O/smi_$tag.txt; echo >> $O/smi_$tag.txt
    sleep 0.5
  done
  python - $O/bench_$tag.json $O/smi_$tag.txt <<'PY'
import json,re,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
w=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", open(sys.argv[2]).read())]
print(sys.argv[1].split('/')[-1], "frames/s", d['value'], "single", d['config']['single_stream_frames_per_sec'], "| socket power samples (W):", sorted(w)[-12:] if w else None)
PY
}
sample headline --config headline --steps 6000
sample headline_s1 --config headline --steps 6000 --streams 1
sample fp16x4 --config fp16x4 --steps 9000
sample fp16x4_s1 --config fp16x4 --steps 9000 --streams 1
sample bf16x6 --config bf16x6 --steps 7000
sample fp16x4_b8 --config fp16x4-b8 --steps 1200
sample stress_b4 --config stress-b4 --steps 300
sample fp16x4_stress_b4 --config fp16x4-stress-b4 --steps 400
head -30 $O/smi_idle.txt
