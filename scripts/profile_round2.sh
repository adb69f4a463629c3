# Round-2 profiles.  Kernel trace + stats of the default bench (k=31) and of k=63, then PMC passes for k=31,
# each counter set in its own run (no trace domains besides --kernel-trace are ever combined with --pmc).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-cross-check --check-reads 50"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k31 -- python3 bench.py --steps 5 --warmup 1 $B > $OUT/bench_trace_k31.log 2>&1
echo trace k31 rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k63 -- python3 bench.py --steps 5 --warmup 1 --k 63 $B > $OUT/bench_trace_k63.log 2>&1
echo trace k63 rc=$?
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_k31_$N -- python3 bench.py --steps 1 --warmup 0 $B > $OUT/bench_pmc_$N.log 2>&1
  echo pmc $N rc=$?
done
