# Kernel trace of the N > 1 code path with a 1-rank RCCL group (bench.py --force-dist).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/proffd; mkdir -p gpurun_out/proffd
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proffd -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-dist > gpurun_out/proffd/bench.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,json
f=sorted(glob.glob('gpurun_out/proffd/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print(r['Name'][:70].ljust(71), r['Calls'], 'avg_ms=%.3f'%(float(r['AverageNs'])/1e6), r['Percentage'])
for l in open('gpurun_out/proffd/bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print('value G/s', round(d['value']/1e9,2), 'ms/step', round(d['ms_per_step'],2), d['config']['check'])
PY
