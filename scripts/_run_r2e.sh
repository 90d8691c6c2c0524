cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests -m gpu -q -x -k "split or whole or small or headline" 2>&1 | grep -i "passed\|failed\|error" | tail -5
timeout 300 python scripts/shard_bench.py 2>/dev/null | tail -4
