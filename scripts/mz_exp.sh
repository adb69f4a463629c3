B="--steps 10 --warmup 3 --no-cpu-baseline --check-reads 300"
for v in 1 0 1 0; do echo "== FUSE_LINES=$v"; TSX_HIP_FUSE_LINES=$v timeout -k 10 300 python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline());print(round(d['value']/1e9,2), round(d['ms_per_step'],3), d['config']['check'], d['roofline'].get('stage_ms'), d['roofline'].get('line_pass_ms'))"; done
