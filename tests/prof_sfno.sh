#!/bin/bash
# usage: tests/prof_sfno.sh <tag>   -- on the GPU box: rocprofv3 kernel trace + PMC passes of the SFNO config-5 forward + loss
# (tests/bench_sfno.py, TRAIN=0) -> gpurun_out/prof_sfno_<tag>/{summary.txt, sfno_traffic.json}
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_sfno_$tag
rm -rf $out; mkdir -p $out
CMD="python $R/tests/bench_sfno.py"
export TRAIN=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- $CMD > $out/trace.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -o pmc -- $CMD > $out/pmc$i.log 2>&1
done
cd $R
python tests/prof_sfno_traffic.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
