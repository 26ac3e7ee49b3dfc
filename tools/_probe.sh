cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
timeout 600 python tools/manhattan_probe.py 2>&1 | tail -8
