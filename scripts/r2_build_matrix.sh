#!/bin/bash
# build-kernel variants: stage times from bench.py (no check)
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-cross-check --check-reads 200 > gpurun_out/mx_$tag.json 2> gpurun_out/mx_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/mx_$tag.json"))
    print("$tag", round(d["ms_per_step"],2),"ms", d["config"]["check"], {s:round(x["ms"],2) for s,x in d["roofline"]["stages"].items()})
except Exception as e: print("$tag ERR", e)
PY
}
run stream X=1
run stream_first TSX_HIP_DEBUG=8
run fifo TSX_HIP_BUILD_V=1
