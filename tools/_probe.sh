cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
timeout 900 python -m pytest tests/test_track_gpu.py -q -x 2>&1 | tail -12
timeout 600 python tools/gen_golden_track_pose.py gpurun_out/track_pose_ref.npz 2>&1 | tail -9
