# round 5, fourth GPU call: the pointwise block fused with the next layer's forward t / y transform -- tests, then the config-5 forward A/B
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "fused_with_the_next or fused_layer_hand_off or lifting_operator_through or config5 or sfno_tiny" 2>&1 | tail -12
for f in 1 0; do
  TCFD_FNO_FUSE_NEXT=$f python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
torch.manual_seed(0)
res = {}
for act in ("ReLU", "GELU"):
    model = fno.SFNO(24, 24, 5, width=10, num_spectral_layers=4, activation=act).to(dev).eval()
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
        res[act] = {"forward_ms": round(timeit(lambda: model(x)), 3), "forward_plus_loss_ms": round(timeit(lambda: loss_fn(model(x), y)), 3),
                    "kernels": bench.fno_kernel_times(lambda: model(x), dev, reps=3)}
print("FUSE_NEXT", os.environ["TCFD_FNO_FUSE_NEXT"], json.dumps(res))
PY
done
