cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "lifting_tail or golden or training_step or pointwise_backward_kernel or tiled_backward or lifting" 2>&1 | tail -4
CASES=150 python tests/micro/pw_tiles_fuzz.py 2>&1 | tail -1
TRAIN=1 ONLY_TRAIN=1 ROWS=16 bash tests/micro/sfno_profile.sh 2>&1 | cut -c1-150 | tail -17
