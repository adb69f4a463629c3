for fq in 2 4 1 2; do echo "== flush_q $fq"; TSX_HIP_WALK_FLUSHQ=$fq timeout -k 10 200 python3 scripts/r3_shard_sim.py 8 2>&1 | grep "minimizer exchange\|plain one" | cut -c1-140; done
