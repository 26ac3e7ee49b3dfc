cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
PLANAR_DUMP_KNIFE=gpurun_out/pose_knife.npz timeout 600 python tools/gen_golden_track_pose.py gpurun_out/track_pose_ref2.npz 2>&1 | grep -v "^step" | tail -8
