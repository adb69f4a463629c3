# experiment batch 1: atomics microbench, insert ablation, reference stability
timeout -k 10 200 ./scripts/atomics_ubench > gpurun_out/atomics_ubench.log 2>&1
echo ubench rc=$?
TSX_HIP_DEBUG=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_noinsert.log 2>&1
echo noinsert rc=$?
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --l 30 > gpurun_out/bench_l30.log 2>&1
echo l30 rc=$?
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from tsxcount_amd import synth
open('/tmp/s3000.fastq','wb').write(synth.fastq(20261004,0,3000))
PY
for t in 16 16 8; do
( time timeout 200 oracle/_ref/tsxCount_ref --input=/tmp/s3000.fastq --k=31 --l=23 --s=2 --mode=CAS --threads=$t > /tmp/ref.out 2> /tmp/ref.err; echo "threads=$t rc=$?" ) 2>&1 | grep "real\|rc="
tail -2 /tmp/ref.out | cut -c1-100
done
