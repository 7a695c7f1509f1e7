cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -x -q -m gpu -k "sobolev or padding or 96 or dense_path or resampl" 2>&1 | tail -25 > gpurun_out/r04_run3_tests.log
cat gpurun_out/r04_run3_tests.log
python tests/micro/c4_pinned_overlap.py 2>&1 | tail -2 > gpurun_out/r04_c4_overlap.json
cat gpurun_out/r04_c4_overlap.json
python tests/bench_sfno.py 2>&1 | tail -5
python - <<'PY'
import sys, os, torch, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(32, 256, 256, 10, generator=g).to(dev); y = torch.randn(32, 256, 256, 10, generator=g).to(dev)
loss_fn = fno.SobolevLoss(n_grid=256, norm_order=0, relative=True).to(dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    r = {"fused_loss_ms": round(timeit(lambda: loss_fn(x, y)), 4), "value": float(loss_fn(x, y))}
    os.environ["TCFD_LOSS_FUSED"] = "0"
    r["composed_loss_ms"] = round(timeit(lambda: loss_fn(x, y)), 4); r["value_composed"] = float(loss_fn(x, y))
print(json.dumps(r))
PY
