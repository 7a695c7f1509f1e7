#!/bin/bash
run() { python tests/micro/r06_solver_ab.py --measure . 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', min(d['regions_ms_per_step']), d['kernel_ms_per_step'])"; }
export AB_N=512 AB_B=64 AB_STEPS=40
for dt in f32 f64; do
export AB_DTYPE=$dt
LABEL="$dt default" run
for c in 8 16 24 32 48 0; do LABEL="$dt chunk=$c" TCFD_CHUNK=$c run; done
for v in 0 1; do LABEL="$dt nt_planes=$v" TCFD_NT_PLANES=$v run; done
for v in 0 2; do LABEL="$dt pair_xcd=$v" TCFD_PAIR_XCD=$v run; done
LABEL="$dt nyq_pack=0" TCFD_NYQ_PACK=0 run
done
