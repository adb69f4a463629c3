"""Sharded pipeline with ONE rank over RCCL (all-to-all with itself): kernel-side cost of scan + split + hist + build."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import torch.distributed as dist
import tsxcount_amd as T
from tsxcount_amd import distributed as TD
n_reads, k = 1087000, 31
nb, nk, _ = T.synth_sizes(20261004, 0, n_reads, k)
text = torch.empty(nb + 256, dtype=torch.uint8, device='cuda:0'); torch.cuda.synchronize()
T.synth_fastq_device(20261004, 0, n_reads, k, text.data_ptr(), nb)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29642')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
m = T.TSXHashMapHIP(30, 0, k)
sc = TD.ShardedCounter(m, nb)
for it in range(4):
    m.clear(); torch.cuda.synchronize(); t0 = time.perf_counter(); sc.step(text.data_ptr(), nb); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('sharded pipeline, 1 rank, RCCL: %.1f ms' % (dt*1e3), m.stats()['distinct'], m.stats()['kmers_added'] == nk)
dist.destroy_process_group()
