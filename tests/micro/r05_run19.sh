cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --no-sfno --no-c4 --no-cpu-baseline 2>gpurun_out/r05_torchrun1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','steps','ms_per_step','scaling')}, d.get('launch'))"
tail -2 gpurun_out/r05_torchrun1.err
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "contraction" 2>&1 | tail -1
