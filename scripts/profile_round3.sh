# Round-3 profiles.  Kernel trace + stats of the default bench (k=31), of k=63, of the Zipf workload at k=63 (BASELINE
# config 4) and of the 2^33-slot table, then PMC passes for k=31, each counter set in its own run (no trace domain besides
# --kernel-trace is ever combined with --pmc).  The program after `--` is python3 itself (no env / shell in between).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r3
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-cross-check --check-reads 50"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k31 -- python3 bench.py --steps 5 --warmup 1 $B > $OUT/bench_trace_k31.log 2>&1
echo trace k31 rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k63 -- python3 bench.py --steps 5 --warmup 1 --k 63 $B > $OUT/bench_trace_k63.log 2>&1
echo trace k63 rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_zipf63 -- python3 bench.py --steps 5 --warmup 1 --workload zipf --k 63 --reads 5300000 $B > $OUT/bench_trace_zipf63.log 2>&1
echo trace zipf63 rc=$?
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_l33 -- python3 bench.py --steps 2 --warmup 1 --l 33 --reads 8700000 $B > $OUT/bench_trace_l33.log 2>&1
echo trace l33 rc=$?
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_k31_$N -- python3 bench.py --steps 1 --warmup 0 $B > $OUT/bench_pmc_$N.log 2>&1
  echo pmc $N rc=$?
done
python3 scripts/summarize_profiles.py 3 > $OUT/summary.log 2>&1; echo summarize rc=$?
# the summaries the round is judged on go back with gpurun_out/ (profiles/ itself is not merged back)
mkdir -p $OUT/profiles_out; cp profiles/round3_* $OUT/profiles_out/ 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete     # (raw traces: keep the stats, drop what would not fit the 64 MiB merge)
