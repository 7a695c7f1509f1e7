# A/B of the tiled pointwise kernels after the two-buffer / constant-channel change: fuzz, tests, timing
cd $GRAFT_REPO_ROOT
CASES=300 python tests/micro/pw_tiles_fuzz.py 2>&1 | tail -3
python -m pytest tests/test_fno_gpu.py -m gpu -x -q -k "pointwise or tiled or wide or widths_16 or golden" 2>&1 | tail -3
python tests/micro/pw_bwd_wide_timing.py 4 8 10 16 20 32 > gpurun_out/r05_pw_tiles_timing_v2.json 2>gpurun_out/r05_pw_tiles_timing_v2.err; cat gpurun_out/r05_pw_tiles_timing_v2.json | head -60
