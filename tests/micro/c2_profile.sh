# per-kernel durations of the C2 step (256^2 x 16 fp32): rocprofv3 kernel trace of tests/micro/c2_rate.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_c2b
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2b -o trace -- python $R/tests/micro/c2_rate.py 2>&1 | tail -3
find $R/gpurun_out/prof_c2b -name "*.csv" | head
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_c2b/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
# gaps between consecutive kernels in the trace
t=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_c2b/**/*kernel_trace.csv', recursive=True)
rows=sorted(csv.DictReader(open(t[0])), key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:len(rows)//2+40]
prev=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(r['Kernel_Name'][:40].ljust(40), 'dur %6d ns'%(e-s), 'gap %6d ns'%((s-prev) if prev else 0))
    prev=e
PY
