# A/B of library builds on one box (experiments): scripts/_libs/lib_<name>.so copied over the in-tree library in turn
cp tsxcount_amd/lib/libtsxcount_hip.so /tmp/orig.so
for v in "$@"; do
  cp scripts/_libs/lib_$v.so tsxcount_amd/lib/libtsxcount_hip.so
  echo "== $v"; timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --check-reads 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline());r=d['roofline'];print(round(d['value']/1e9,2), round(d['ms_per_step'],3), d['config']['check'], 'build', round(r['kernel_ms'],3))"
done
cp /tmp/orig.so tsxcount_amd/lib/libtsxcount_hip.so
