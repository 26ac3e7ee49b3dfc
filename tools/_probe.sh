cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "pose or opt or track or adapters" 2>&1 | tail -5
timeout 300 python bench.py --workload pose 2>/dev/null | tail -1 | cut -c1-900
