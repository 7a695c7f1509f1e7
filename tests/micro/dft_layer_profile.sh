cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/l96.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from torch_cfd_amd import fno
dev = torch.device("cuda:0")
X = int(os.environ.get("XS", 96))
layer = fno.SpectralConvS(10, 10, 24, 24, 5).to(dev)
x = torch.randn(32 if X < 256 else 8, 10, X, X, 10, device=dev)
with torch.no_grad():
    for _ in range(12): layer(x)
torch.cuda.synchronize()
PY
rm -rf $R/gpurun_out/prof_dft
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dft -o trace -- python /tmp/l96.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_dft/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), '%9.1f us avg' % (float(r['AverageNs']) / 1e3))
PY
