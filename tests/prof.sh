#!/bin/bash
# usage: tests/prof.sh <tag> [bench args...]   -- run on the GPU box via gpurun; writes gpurun_out/prof_<tag>/
tag=$1; shift
cd /root/repo
export TMPDIR=/tmp
out=/root/repo/gpurun_out/prof_$tag
mkdir -p $out
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-sfno --no-probe --no-c4 $@"
CMD=${PROF_CMD:-python bench.py $ARGS}   # PROF_CMD="python tests/bench_fno.py" profiles something else
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- $CMD > $out/trace.log 2>&1
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -o pmc -- $CMD > $out/pmc$i.log 2>&1
done
python tests/prof_summarize.py $out > $out/summary.txt 2>&1
# keep what is read afterwards (kernel stats, per-dispatch counter tables, logs); the raw traces are tens of MB per run and
# gpurun only brings back 64 MB in total
find $out -type f \( -name '*kernel_trace.csv' -o -name '*.db' -o -name '*agent_info.csv' -o -name '*.json' -o -name '*.pftrace' \) -delete
find $out -type f -name '*counter_collection.csv' -size +6M -delete
cat $out/summary.txt
