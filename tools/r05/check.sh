cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout 600 python3 __graft_entry__.py smoke 2>&1 | tail -1
