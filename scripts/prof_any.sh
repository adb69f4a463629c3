cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profq; mkdir -p gpurun_out/profq
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profq -- python3 $1 > gpurun_out/profq/run.log 2>&1
echo rc=$?
grep -v amdgpu.ids gpurun_out/profq/run.log | tail -6
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/profq/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:52].ljust(53), r['Calls'], 'avg_ms=%.3f'%(float(r['AverageNs'])/1e6), r['Percentage'])
PY
