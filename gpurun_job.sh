cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 3 0 3; do GATSSPG_MLP3_TILE=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --kernel mlp3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp3 tile $v', d['value'], d['config']['single_frame_latency_ms'], d['roofline']['kernel_ms'])"; done
for p in 0 1; do echo pooltile=$p; SPP_POOL_TILE=$p SPP_BENCH_ARGS="--streams 1" tools/spp_tune.sh "0,0,0,1,1,1,1,1,1,1" "0,3,0,1,1,1,1,1,1,1"; SPP_POOL_TILE=$p tools/spp_tune.sh "0,0,0,1,1,1,1,1,1,1" "0,3,0,1,1,1,1,1,1,1"; done
SPP_POOL_TILE=1 timeout 300 python -m pytest tests/test_spp_hip_parity.py -x -q -m gpu -k "golden or dense" 2>&1 | tail -2
