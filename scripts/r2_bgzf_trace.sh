#!/bin/bash
# Kernel trace of the BGZF entry point (scripts/r2_bgzf_timing.py): how long is inflate_members_kernel itself?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_bgzf
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/r2_bgzf_timing.py ${1:-200000} ${2:-28} > $OUT/run.log 2>&1
echo rc=$?
tail -n 3 $OUT/run.log
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-60s calls %5s  avg %10.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
