# end-of-round evidence: GPU tests, default bench, rocprofv3 kernel stats + PMC of the headline, traffic.json, SFNO profiles,
# the training step by operator, the differentiable solver step
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r03_final_tests.log
python bench.py > gpurun_out/r03_final3_bench.json 2> gpurun_out/r03_final_bench.err
bash tests/prof.sh r03_final > gpurun_out/r03_final_prof.log 2>&1
python tests/prof_traffic.py gpurun_out/prof_r03_final 1024 64 f64 16 "r03_final (round-3 build: same 1024^2 kernels as r02_final2)" > gpurun_out/r03_final_traffic.log 2>&1
cp profiles/traffic.json gpurun_out/r03_traffic.json
TRAIN=1 ONLY_TRAIN=1 ROWS=40 bash tests/micro/sfno_profile.sh > gpurun_out/r03_final_sfno_train.txt 2>&1
cd $GRAFT_REPO_ROOT; TRAIN=0 ROWS=30 bash tests/micro/sfno_profile.sh > gpurun_out/r03_final_sfno_fwd.txt 2>&1
cd $GRAFT_REPO_ROOT; python tests/micro/sfno_train_ops.py 45 2>/dev/null | cut -c1-62,120-260 > gpurun_out/r03_final_sfno_train_ops.txt
python tests/micro/grad_step_timing.py 2>/dev/null | tail -1 > gpurun_out/r03_final_grad_step.json
bash tests/micro/prof_cmd.sh grad 24 python $GRAFT_REPO_ROOT/tests/micro/grad_step_only.py 2>&1 | cut -c1-160 > gpurun_out/r03_final_grad_step_kernels.txt
cd $GRAFT_REPO_ROOT; cat gpurun_out/r03_final_tests.log; tail -2 gpurun_out/r03_final_bench.err; cat gpurun_out/r03_final_traffic.log; cat gpurun_out/r03_final_grad_step.json
