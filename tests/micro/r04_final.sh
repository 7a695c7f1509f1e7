# end-of-round evidence (round 4): GPU tests, default bench, rocprofv3 kernel stats + PMC of the headline, traffic.json, SFNO PMC table,
# C2 PMC, strong-scaling proxy, C4 trace
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04_final_tests.log
python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
bash tests/prof.sh r04_final > gpurun_out/r04_final_prof.log 2>&1
python tests/prof_traffic.py gpurun_out/prof_r04_final 1024 64 f64 16 "r04_final (round-4 build: row pass v7 without the mirror swizzle)" > gpurun_out/r04_final_traffic.log 2>&1
cp profiles/traffic.json gpurun_out/r04_traffic.json
bash tests/prof_sfno.sh r04 > gpurun_out/r04_final_sfno_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python tests/micro/strong_proxy.py > gpurun_out/r04_strong_proxy.json 2>/dev/null
TRAIN=1 ONLY_TRAIN=1 ROWS=30 bash tests/micro/sfno_profile.sh > gpurun_out/r04_final_sfno_train.txt 2>&1
cd $GRAFT_REPO_ROOT; cat gpurun_out/r04_final_tests.log; tail -2 gpurun_out/r04_final_bench.err; cat gpurun_out/r04_final_traffic.log
python tests/micro/gelu_bench.py > gpurun_out/r04_gelu.json 2>/dev/null
python tests/micro/subsample_bench.py > gpurun_out/r04_subsample.json 2>/dev/null
python tests/micro/dense_vs_fused_layer.py > gpurun_out/r04_layer_by_grid.json 2>/dev/null
