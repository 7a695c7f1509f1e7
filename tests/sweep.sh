#!/bin/bash
# tuning sweep on the GPU box (n=1024 fp64 B=64). usage: sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...
cd /root/repo
show() { python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line)
        print('  ms/step=%.2f steps/s=%.1f | '%(d['ms_per_step'],d['value'])+' '.join('%s=%.3f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items()))
"; }
for cfg in "$@"; do
  echo "[$cfg]"; env $cfg python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-sfno --no-probe ${BENCH_ARGS} 2>&1 | show
done
