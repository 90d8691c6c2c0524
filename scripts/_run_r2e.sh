cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "sincos" 2>&1 | grep -i "passed\|failed\|error\|assert" | tail -8
timeout 300 python scripts/config_bench.py 2>/dev/null | tail -8
