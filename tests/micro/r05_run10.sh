# the lanes contraction kernel and the batch-split weight gradient: tests, timing sweep
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "contraction or spectral_conv or golden" 2>&1 | tail -5
python tests/micro/contract_timing.py 10 8 > gpurun_out/r05_contract_timing.json 2> gpurun_out/r05_contract_timing.err; cat gpurun_out/r05_contract_timing.json; tail -3 gpurun_out/r05_contract_timing.err
