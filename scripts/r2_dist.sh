#!/bin/bash
# multi-GPU path on a one-GPU box: (1) the RCCL leg through its API with a 1-rank group, (2) two gloo ranks on cuda:0
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline > gpurun_out/dist_force.json 2> gpurun_out/dist_force.err; echo "force-dist rc=$?"
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/dist_force.json"))
    print("force-dist", round(d["ms_per_step"],2),"ms", d["config"]["check"], {s:round(x["ms"],2) for s,x in d["roofline"]["stages"].items()}, "gap", round(d["roofline"]["exchange_gap_ms"],2))
except Exception as e: print("force-dist ERR", e)
PY
tail -3 gpurun_out/dist_force.err
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29688 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --reads 400000 --table-bits 29 --no-cpu-baseline > gpurun_out/dist_gloo2.json 2> gpurun_out/dist_gloo2.err; echo "gloo2 rc=$?"
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/dist_gloo2.json"))
    print("gloo2", round(d["ms_per_step"],2),"ms", d["config"]["check"], d["config"]["check_detail"])
except Exception as e: print("gloo2 ERR", e)
PY
tail -5 gpurun_out/dist_gloo2.err
