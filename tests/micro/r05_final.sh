# end-of-round evidence (round 5): GPU tests, default bench, rocprofv3 kernel stats + PMC of the headline, traffic.json, SFNO PMC table,
# SFNO training-step kernel stats, torch ops left in a training step, contraction kernels, pointwise tiles timing, strong-scaling proxy
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05_final_tests.log
python bench.py > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
bash tests/prof.sh r05_final > gpurun_out/r05_final_prof.log 2>&1
python tests/prof_traffic.py gpurun_out/prof_r05_final 1024 64 f64 16 "r05_final (round-5 build; the solver kernels are round 4's)" > gpurun_out/r05_final_traffic.log 2>&1
cp profiles/traffic.json gpurun_out/r05_traffic.json
bash tests/prof_sfno.sh r05 > gpurun_out/r05_final_sfno_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tests/micro/strong_proxy.py > gpurun_out/r05_strong_proxy.json 2>/dev/null
TRAIN=1 ONLY_TRAIN=1 ROWS=40 bash tests/micro/sfno_profile.sh > gpurun_out/r05_final_sfno_train.txt 2>&1
cd $GRAFT_REPO_ROOT
ROWS=40 python tests/micro/train_step_ops.py 2>&1 | tail -42 > gpurun_out/r05_final_train_step_ops.txt
python tests/micro/contract_wide_timing.py 10 14 16 20 24 32 > gpurun_out/r05_final_contract_timing.json 2>/dev/null
python tests/micro/pw_bwd_wide_timing.py 4 8 10 16 20 32 > gpurun_out/r05_final_pw_tiles_timing.json 2>/dev/null
cat gpurun_out/r05_final_tests.log; tail -2 gpurun_out/r05_final_bench.err; cat gpurun_out/r05_final_traffic.log
