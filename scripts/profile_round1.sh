# Round-1 profiles.  Kernel trace + stats for both insert paths, then PMC passes for
# the default (partitioned) path, each counter set in its own run (no trace domains
# besides --kernel-trace are ever combined with --pmc).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r1
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_partitioned -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_trace_partitioned.log 2>&1
echo trace partitioned rc=$?
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_atomic -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --path atomic --l 31 > $OUT/bench_trace_atomic.log 2>&1
echo trace atomic rc=$?
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_partitioned_$N -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_pmc_$N.log 2>&1
  echo pmc $N rc=$?
done
