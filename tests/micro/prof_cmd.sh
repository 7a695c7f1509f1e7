# usage: bash tests/micro/prof_cmd.sh <tag> <rows> <command...>   -> top kernels of the command (rocprofv3 kernel stats)
tag=$1; rows=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o trace -- "$@" 2>&1 | tail -1
ROWS=$rows python - <<PY
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/prof_$tag/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:int(os.environ['ROWS'])]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), '%9.1f us avg'%(float(r['AverageNs'])/1e3), '%7.2f ms total'%(float(r['TotalDurationNs'])/1e6), r['Percentage'])
PY
