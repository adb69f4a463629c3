# One PMC pass of the default bench with the counters given in $PMC (space separated); prints per-kernel sums.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcq; mkdir -p gpurun_out/pmcq
timeout -k 10 400 rocprofv3 --pmc $PMC --output-format csv -d gpurun_out/pmcq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcq/bench.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/pmcq/*/*_counter_collection.csv'))[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    if any(s in k for s in ('partition_ring','scan_log','build_segments')):
        agg[k+'#'+r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
for k in sorted(agg): print(k, {c: '%.3g'%v for c,v in sorted(agg[k].items())})
PY
