# Rehearsal of bench.py --gpus 2 on ONE GPU (gloo collective through host memory): exercises the
# N > 1 code path end to end; the numbers mean nothing (two ranks share a GPU, PCIe staging).
for MODE in shard tables; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --reads 200000 --table-bits 28 --merge $MODE > gpurun_out/rehearse_$MODE.log 2>&1
echo "$MODE rc=$?"; grep "^{" gpurun_out/rehearse_$MODE.log | cut -c1-600; tail -3 gpurun_out/rehearse_$MODE.log | cut -c1-300
done
