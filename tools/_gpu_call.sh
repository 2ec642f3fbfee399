R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout=180 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/ab_tuning.py --kernel score_exp "" SCORE_TILE=0 2>&1 | tee $O/ab.log
