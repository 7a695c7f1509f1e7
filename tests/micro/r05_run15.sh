cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py tests/test_training_kernels_gpu.py -m gpu -x -q 2>&1 | tail -4
ROWS=14 python tests/micro/train_step_ops.py 2>&1 | tail -16 | cut -c1-200
