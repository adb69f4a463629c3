#!/bin/bash
# kernel trace + stats of the default bench; prints the per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_quick; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cross-check --check-reads 50 $BENCH_ARGS > $OUT/bench.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/trace_quick/t/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:10]:
    print('%-50s calls %3s avg %.3f ms' % (r['Name'].split('(')[0][-50:], r['Calls'], float(r['AverageNs'])/1e6))
PY
