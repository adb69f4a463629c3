cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 64 72 192; do
rm -rf gpurun_out/profq; mkdir -p gpurun_out/profq
TSX_HIP_DEBUG=$D timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --l 30 > gpurun_out/profq/bench.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/profq/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'partition_kernel' in r['Kernel_Name']]
print('dbg=$D partition calls ms', [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,2) for r in rows])
PY
done
