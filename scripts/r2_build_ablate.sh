#!/bin/bash
# build-kernel ablations: stage times from bench.py (no check), then a kernel trace of the default build
mkdir -p gpurun_out
for dbg in 0 2 4 6 8; do
  TSX_HIP_DEBUG=$dbg timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cross-check --check-reads 1 > gpurun_out/abl_$dbg.json 2> gpurun_out/abl_$dbg.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/abl_$dbg.json"))
    print("dbg=$dbg", round(d["ms_per_step"],2),"ms", {s:round(x["ms"],2) for s,x in d["roofline"]["stages"].items()})
except Exception as e: print("dbg=$dbg ERR", e)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cross-check --check-reads 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_r2a.log 2>&1
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_r2a -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -20 {}'
