#!/bin/bash
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-cross-check --check-reads 200 > gpurun_out/mx_$tag.json 2> gpurun_out/mx_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/mx_$tag.json"))
    print("$tag", round(d["ms_per_step"],2),"ms", d["config"]["check"], {s:round(x["ms"],2) for s,x in d["roofline"]["stages"].items()}, d["config"].get("check_detail",{}).get("failure_counters"))
except Exception as e: print("$tag ERR", e)
PY
}
run base X=1
run ring4 TSX_HIP_RING_BITS=4
run ring6 TSX_HIP_RING_BITS=6
run cpr4 TSX_HIP_CPR2=4
