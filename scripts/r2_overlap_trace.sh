#!/bin/bash
# Two gloo ranks on cuda:0 (the exchange is staged through host memory), rank 0 under rocprofv3: does the
# exchange of window i overlap the scan of window i+1?  Kernel + memory-copy trace, no counters.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29701 WORLD_SIZE=2
OUT=gpurun_out/prof_r2_overlap_${TSX_HIP_SHARD_MODE:-desc}
rm -rf $OUT; mkdir -p $OUT
ARGS="--gpus 2 --backend gloo --steps 2 --warmup 1 --reads 400000 --table-bits 29 --no-cpu-baseline --check-reads 20"
RANK=1 LOCAL_RANK=1 timeout -k 10 400 python3 bench.py $ARGS > $OUT/rank1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/rank0 -- python3 bench.py $ARGS > $OUT/rank0.log 2>&1
echo rank0 rc=$?
wait $P1; echo rank1 rc=$?
python3 scripts/summarize_overlap.py $OUT/rank0 | tee $OUT/summary.txt
