"""PCIe-inclusive rate of the host entry point and the single-GPU overhead of the sharded pipeline."""
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
host = text[:nb].cpu().numpy().tobytes()
m = T.TSXHashMapHIP(30, 0, k)
for it in range(3):
    m.clear(); t0 = time.perf_counter(); m.countFastq(host); dt = time.perf_counter() - t0
    print('host entry: %.1f ms  %.2f G k-mers/s  (%.1f GB/s of text)' % (dt*1e3, nk/dt/1e9, nb/dt/1e9), m.stats()['kmers_added'] == nk)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29641')
dist.init_process_group('gloo', rank=0, world_size=1)
sc = TD.ShardedCounter(m, nb)
for it in range(3):
    m.clear(); torch.cuda.synchronize(); t0 = time.perf_counter(); sc.step(text.data_ptr(), nb); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('sharded pipeline on one GPU (gloo staging through host!): %.1f ms' % (dt*1e3), m.stats()['distinct'])
