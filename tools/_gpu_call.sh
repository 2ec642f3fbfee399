R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout=180 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log; grep -h "flips" $O/pytest.log | head
