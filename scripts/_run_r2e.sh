cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -k "per_element" 2>&1 | grep -E "passed|failed|Error|assert|FAILED|^E " | tail -20
