cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lsd_gpu.py tests/test_track_gpu.py -m gpu -x -q -s 2>&1 | grep -E "top-only|passed|failed|Error|^E" | cut -c1-400
