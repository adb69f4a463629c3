for mg in 1 0 1 0; do TSX_HIP_MZ_MERGE=$mg timeout -k 10 200 python3 scripts/r3_shard_sim.py 8 2>&1 | grep "minimizer exchange" | cut -c1-330; done
