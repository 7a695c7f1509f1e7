# round 5, fifth GPU call: the forward block of wide layers on the matrix pipe -- tests, then width lines with and without it
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "wide_forward or fused_pointwise or widths_16 or pointwise_backward_kernel" 2>&1 | tail -6
for f in 1 0; do
TCFD_PW_FWD_TILES=$f python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
out = {}
for key, width, act in (("w16_relu", 16, "ReLU"), ("w16_gelu", 16, "GELU"), ("w24", 24, "ReLU"), ("w32", 32, "ReLU"), ("w32_gelu", 32, "GELU")):
    r = bench.sfno_width_line(dev, width, act=act)
    out[key] = {k: r[k] for k in ("forward_ms", "forward_plus_loss_ms", "train_step_ms", "pointwise_fp32_TFLOPs_in_forward", "peak_memory_GB")}
    out[key]["pointwise_ms"] = r["train_kernels"].get("pointwise", {}).get("avg_ms")
    out[key]["pointwise_bwd_ms"] = r["train_kernels"].get("pointwise_bwd", {}).get("avg_ms")
print("FWD_TILES", os.environ["TCFD_PW_FWD_TILES"], json.dumps(out))
PY
done
