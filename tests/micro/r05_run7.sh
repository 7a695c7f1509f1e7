# round 5: the whole GPU suite + smoke on the pruned library (row kernels v3 / v4 / v6 and the plane-split pass removed)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r05_run7_tests.log
cat gpurun_out/r05_run7_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --no-c4 --no-sfno --steps 20 2> gpurun_out/r05_run7_bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['kernels'])"
