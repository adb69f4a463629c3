# Refresh of the k=63 traces after two-limb segments went to 64 KiB (two build workgroups per CU), + the bench lines kept under profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r3b
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-cross-check --check-reads 50"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k63 -- python3 bench.py --steps 5 --warmup 1 --k 63 $B > $OUT/bench_trace_k63.log 2>&1
echo trace k63 rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_zipf63 -- python3 bench.py --steps 5 --warmup 1 --workload zipf --k 63 --reads 5300000 $B > $OUT/bench_trace_zipf63.log 2>&1
echo trace zipf63 rc=$?
python3 bench.py --k 63 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_k63.json 2>/dev/null; echo k63 rc=$?
python3 bench.py --workload zipf --k 63 --reads 5300000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_zipf63.json 2>/dev/null; echo zipf rc=$?
python3 bench.py --k 127 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_k127.json 2>/dev/null; echo k127 rc=$?
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_k31.json 2>/dev/null; echo k31 rc=$?
find $OUT -name "*kernel_trace.csv" -size +2M -delete
