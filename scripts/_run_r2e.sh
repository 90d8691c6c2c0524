cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -5
timeout 300 python scripts/config_bench.py 2>/dev/null | tail -6
