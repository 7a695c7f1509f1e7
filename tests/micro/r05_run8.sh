cd $GRAFT_REPO_ROOT
CASES=300 python tests/micro/pw_tiles_fuzz.py 2>&1 | tail -3
SEED=7 CASES=300 python tests/micro/pw_tiles_fuzz.py 2>&1 | tail -3
hipcc --offload-arch=gfx950 -O3 tests/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo | grep f64 > gpurun_out/r05_mfma_f64_overlap.txt; cat gpurun_out/r05_mfma_f64_overlap.txt
