for D in 0 2 16 32 48; do
TSX_HIP_DEBUG=$D timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --l 30 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('dbg=$D', round(d['ms_per_step'],2), 'scan', round(d['roofline']['kernel_ms'],2), 'part+build', round(d['roofline']['partition_build_ms'],2), d['config']['check'])"
done
