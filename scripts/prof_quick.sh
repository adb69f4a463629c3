cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profq; mkdir -p gpurun_out/profq
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/profq/bench.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/profq/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:45].ljust(46), r['Calls'], 'avg_ms=%.3f'%(float(r['AverageNs'])/1e6), r['Percentage'])
f=glob.glob('gpurun_out/profq/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'partition' in r['Kernel_Name']]
print('partition calls ms', [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,2) for r in rows])
for l in open('gpurun_out/profq/bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print('value G/s', round(d['value']/1e9,2), 'ms/step', round(d['ms_per_step'],2), 'scan', round(d['roofline']['kernel_ms'],2), 'part+build', round(d['roofline']['partition_build_ms'],2), d['config']['check'])
PY
