for C in 4 2 8 1; do
echo "== cpr2=$C"
TSX_HIP_CPR2=$C BENCH_ARGS="--l 30" bash scripts/prof_quick.sh | grep "partition calls\|build_seg\|value G"
done
