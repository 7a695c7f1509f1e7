# round 5, first GPU call: the tiled pointwise backward (csrc/tcfd_fno_tiles.hip) -- tests, then timing at the config-5 grid
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "pointwise_backward or widths_16 or reference_shape_suite or fused_layer_node or training_step or pointwise_block" 2>&1 | tail -15 > gpurun_out/r05_run1_tests.log
cat gpurun_out/r05_run1_tests.log
timeout 600 python tests/micro/pw_bwd_wide_timing.py > gpurun_out/r05_pw_bwd_wide.json 2> gpurun_out/r05_pw_bwd_wide.err
cat gpurun_out/r05_pw_bwd_wide.json; tail -3 gpurun_out/r05_pw_bwd_wide.err
python -m pytest tests/test_ns2d_gpu.py -m gpu -x -q -k "subsample" 2>&1 | tail -3
