#!/bin/bash
# PMC counters of the scan kernel for the variants given as "tag ENV" pairs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/scanpmc; rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-cross-check --check-reads 20 --steps 1 --warmup 0"
while [ $# -gt 0 ]; do
  tag=$1; envs=$2; shift 2
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-30)
    export $envs >/dev/null 2>&1
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${tag}_$N -- python3 bench.py $B > $OUT/${tag}_$N.log 2>&1
    echo $tag $N rc=$?
  done
  for v in $envs; do unset ${v%%=*}; done
done
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob('gpurun_out/scanpmc/*/*/*counter_collection.csv'):
    tag=f.split('/')[2].split('_')[0]
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        if 'scan_' in kn and 'line' not in kn:
            acc[(tag,r['Counter_Name'])]+=float(r['Counter_Value'])
for k in sorted(acc): print(k, acc[k])
PY
