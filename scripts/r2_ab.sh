#!/bin/bash
# A/B of env-variable variants of the default bench: scripts/r2_ab.sh tag1 "VAR=1 VAR2=2" tag2 "" ...
mkdir -p gpurun_out
while [ $# -gt 0 ]; do
  tag=$1; envs=$2; shift 2
  env $envs timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-cross-check --check-reads 200 $BENCH_ARGS > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_$tag.json"))
    print("$tag", round(d["ms_per_step"],2),"ms", d["config"]["check"], {s:round(x["ms"],2) for s,x in d["roofline"]["stages"].items()})
except Exception as e: print("$tag ERR", e)
PY
done
