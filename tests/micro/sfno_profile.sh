# per-kernel durations of the SFNO config-5 forward + loss: rocprofv3 kernel trace of tests/bench_sfno.py (TRAIN=0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_sfno2
TRAIN=${TRAIN:-0} ONLY_TRAIN=${ONLY_TRAIN:-0} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sfno2 -o trace -- python $R/tests/bench_sfno.py 2>&1 | tail -1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_sfno2/**/*kernel_stats.csv', recursive=True)
tot=0
rows=list(csv.DictReader(open(f[0])))
for r in rows[:int(os.environ.get('ROWS', 24))]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), '%9.1f us avg'%(float(r['AverageNs'])/1e3), '%6.2f ms total'%(float(r['TotalDurationNs'])/1e6), r['Percentage'])
PY
