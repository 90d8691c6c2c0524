cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "tile_iteration or small_element or split_whole or whole_iteration" 2>&1 | tail -30
