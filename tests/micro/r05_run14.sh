# fused Sobolev-loss backward: tests, then the C5 training step A/B (fused backward on / off) and the op list
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "sobolev or loss or training_step or gradients_golden or graphed" 2>&1 | tail -8
TRAIN=1 ONLY_TRAIN=1 python tests/bench_sfno.py 2>/dev/null | tail -1
TCFD_LOSS_FUSED_BWD=0 TRAIN=1 ONLY_TRAIN=1 python tests/bench_sfno.py 2>/dev/null | tail -1
TRAIN=1 ONLY_TRAIN=1 python tests/bench_sfno.py 2>/dev/null | tail -1
ROWS=30 python tests/micro/train_step_ops.py 2>&1 | tail -32 | cut -c1-200
