cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "pointwise_backward_kernel or golden or training_step" 2>&1 | tail -3
TRAIN=1 ONLY_TRAIN=1 ROWS=14 bash tests/micro/sfno_profile.sh 2>&1 | cut -c1-150 | tail -15
