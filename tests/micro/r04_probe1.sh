cd $GRAFT_REPO_ROOT
python tests/micro/strong_proxy.py > gpurun_out/r04_strong_proxy0.json 2> gpurun_out/r04_strong_proxy0.err
cat gpurun_out/r04_strong_proxy0.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b8
BS=8 STEPS=10 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b8 -o trace -- python $GRAFT_REPO_ROOT/tests/micro/strong_proxy.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tests/micro/trace_gaps.py gpurun_out/prof_b8 70 > gpurun_out/r04_b8_gaps.txt 2>&1
tail -75 gpurun_out/r04_b8_gaps.txt
