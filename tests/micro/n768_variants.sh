# 768^2 x 64 fp64 (and other 3 * 2^k sizes) step rate with the default plan, and variants
for cfg in "768 f64 -1 0" "768 f64 -1 4" "768 f64 10 0" "768 f32 -1 0" "384 f64 -1 0" "192 f64 -1 0" "1024 f64 -1 0"; do
  set -- $cfg
  echo -n "n=$1 $2 chunk=$3 rows_v=$4: "
  TCFD_CHUNK=$3 TCFD_ROWS_V=$4 python bench.py --n $1 --dtype $2 --steps 20 --warmup 3 --no-sfno --no-c4 --no-cpu-baseline --no-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['chunking']['fields_per_chunk'], d['fused_steps_api']['steps_per_s'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done
