# rocprofv3 kernel stats + PMC passes of the contraction kernels (csrc/tcfd_fno.hip: k_contract_lanes, k_modes_gemm, k_contract_wgrad,
# beside k_contract_mfma) at the config-5 spectrum.  usage (on the GPU box): bash tests/micro/contract_prof.sh <tag> <widths ...>
#   -> gpurun_out/prof_<tag>/summary.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
CMD="python $R/tests/micro/contract_wide_timing.py $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- $CMD > $out/trace.log 2>&1
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -o pmc -- $CMD > $out/pmc$i.log 2>&1
done
python $R/tests/prof_summarize.py $out > $out/summary.txt 2>&1
grep -B1 -A34 "k_contract\|k_modes_gemm" $out/summary.txt | head -400
