#!/bin/bash
# One RCCL rank (RCCL refuses two ranks on one GPU) under rocprofv3: the all-to-all of window i is a device-side
# RCCL kernel / copy on the exchange stream; does it run while scan kernels of window i+1 run on the compute stream?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29703 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
OUT=gpurun_out/prof_r2_overlap_rccl
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/rank0 -- python3 bench.py --force-dist --steps 2 --warmup 1 --no-cpu-baseline --check-reads 20 --no-cross-check > $OUT/rank0.log 2>&1
echo rc=$?
python3 scripts/summarize_overlap.py $OUT/rank0 | tee $OUT/summary.txt
python3 - <<'PY'
import csv,glob
kf=sorted(glob.glob('gpurun_out/prof_r2_overlap_rccl/rank0/*/*_kernel_trace.csv'))[-1]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:60],r.get('Stream_Id', r.get('Queue_Id',''))) for r in csv.DictReader(open(kf))]
rows.sort()
# the last step: print the tail of the timeline relative to its first scan kernel
t0=[r for r in rows if 'line_count' in r[2]][-4][0]
for s,e,n,q in rows:
    if s>=t0 and ('tsx::' in n or 'ccl' in n.lower()):
        print('%9.3f %9.3f ms  q=%s  %s'%((s-t0)/1e6,(e-t0)/1e6,q,n))
PY
