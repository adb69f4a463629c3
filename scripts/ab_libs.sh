# A/B of library builds on one box (experiments): scripts/_libs/lib_<name>.so copied over the in-tree library in turn.
# BENCH_EXTRA: further bench.py options (e.g. "--k 63").
cp tsxcount_amd/lib/libtsxcount_hip.so /tmp/orig.so
for v in "$@"; do
  cp scripts/_libs/lib_$v.so tsxcount_amd/lib/libtsxcount_hip.so
  echo "== $v"
  timeout -k 10 300 python3 bench.py $BENCH_EXTRA --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --check-reads 100 2>/dev/null > /tmp/ab_line.json
  python3 - <<'PY'
import json
d = json.loads(open('/tmp/ab_line.json').readline())
r = d['roofline']
print(round(d['value'] / 1e9, 2), round(d['ms_per_step'], 3), d['config']['check'], r['kernel'][:30], round(r['kernel_ms'], 3),
      'line', round(r.get('line_pass_ms', 0), 3), 'partition+build', round(r.get('partition_build_ms', 0), 3))
PY
done
cp /tmp/orig.so tsxcount_amd/lib/libtsxcount_hip.so
