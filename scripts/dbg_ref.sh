set -x
ls -la oracle/_ref/
ldd oracle/_ref/tsxCount_ref | head -20
python - <<'PY'
import sys; sys.path.insert(0,'.')
from tsxcount_amd import synth
open('/tmp/s.fastq','wb').write(synth.fastq(1,0,300))
PY
nproc
( time timeout 120 oracle/_ref/tsxCount_ref --input=/tmp/s.fastq --k=31 --l=23 --s=2 --mode=CAS --threads=16 > /tmp/ref.out 2> /tmp/ref.err ) 2>&1 | tail -4
echo rc=$?
tail -5 /tmp/ref.err; tail -3 /tmp/ref.out
