#!/bin/bash
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-cross-check --check-reads 200 > gpurun_out/mx_$tag.json 2> gpurun_out/mx_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/mx_$tag.json"))
    print("$tag", round(d["ms_per_step"],2),"ms", d["config"]["check"], {s:round(x["ms"],2) for s,x in d["roofline"]["stages"].items()})
except Exception as e: print("$tag ERR", e)
PY
}
run la1 TSX_HIP_BUILD_LOOKAHEAD=1
run la2 TSX_HIP_BUILD_LOOKAHEAD=2
run la3 TSX_HIP_BUILD_LOOKAHEAD=3
run la1b TSX_HIP_BUILD_LOOKAHEAD=1
