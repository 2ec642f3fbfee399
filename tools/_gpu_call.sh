R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=180 -p no:cacheprovider 2>&1 | tail -4
bash tools/collect_profiles.sh r02b 2>&1 | tail -3
