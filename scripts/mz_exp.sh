# A/B of library builds on one box: each variant copied over the in-tree library in turn (experiments only)
cp tsxcount_amd/lib/libtsxcount_hip.so /tmp/orig.so
for v in round4 round5 round6 round4; do
  cp scripts/_libs/lib_$v.so tsxcount_amd/lib/libtsxcount_hip.so
  echo "== $v"; timeout -k 10 200 python3 scripts/r3_shard_sim.py 8 check 2>&1 | grep "minimizer exchange\|plain one\|check" | cut -c1-140
done
cp /tmp/orig.so tsxcount_amd/lib/libtsxcount_hip.so
