# round 5, second GPU call: the failing golden test with its traceback, the whole FNO GPU suite, PMC of the tiled backward
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "widths_16" 2>&1 | tail -40 > gpurun_out/r05_run2_w16.log
cat gpurun_out/r05_run2_w16.log
python -m pytest tests/test_fno_gpu.py tests/test_training_kernels_gpu.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r05_run2_fno_tests.log
tail -25 gpurun_out/r05_run2_fno_tests.log
TCFD_PW_BWD_TILES=2 ACTS=ReLU bash tests/micro/pw_bwd_wide_prof.sh r05_tiles_w10 10 > gpurun_out/r05_tiles_w10_pmc.txt 2>&1
ACTS=ReLU bash tests/micro/pw_bwd_wide_prof.sh r05_tiles_w16_32 16 32 > gpurun_out/r05_tiles_w16_32_pmc.txt 2>&1
tail -80 gpurun_out/r05_tiles_w10_pmc.txt
