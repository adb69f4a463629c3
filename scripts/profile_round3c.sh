# The minimizer exchange: kernel trace of the one-GPU simulation of an 8-GPU step, the simulation's own lines for N = 8 / 4 / 2, the
# bench through RCCL with one rank and through gloo with two.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r3c
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mini8 -- python3 scripts/r3_shard_sim.py 8 check > $OUT/sim8_traced.log 2>&1
echo trace rc=$?
for n in 8 4 2; do timeout -k 10 300 python3 scripts/r3_shard_sim.py $n check > $OUT/sim$n.log 2>&1; echo sim $n rc=$?; done
SIM_WINDOWS=2 timeout -k 10 300 python3 scripts/r3_shard_sim.py 8 check > $OUT/sim8_w2.log 2>&1; echo sim 8 w2 rc=$?
timeout -k 10 300 python3 bench.py --force-dist --merge mini --steps 5 --warmup 2 --no-cpu-baseline --check-reads 200 > $OUT/bench_mini_rccl1.json 2> $OUT/bench_mini_rccl1.err; echo rccl1 rc=$?
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --merge mini --steps 2 --warmup 1 --no-cpu-baseline --check-reads 100 > $OUT/bench_mini_gloo2.json 2> $OUT/bench_mini_gloo2.err; echo gloo2 rc=$?
find $OUT -name "*kernel_trace.csv" -size +2M -delete
