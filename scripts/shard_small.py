import os, sys, time
sys.path.insert(0, '.')
import torch
import torch.distributed as dist
import tsxcount_amd as T
from tsxcount_amd import distributed as TD
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29643')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
for n_reads, l in ((20000, 25), (200000, 28), (1087000, 30)):
    k = 31
    nb, nk, _ = T.synth_sizes(5, 0, n_reads, k)
    text = torch.empty(nb + 256, dtype=torch.uint8, device='cuda:0'); torch.cuda.synchronize()
    T.synth_fastq_device(5, 0, n_reads, k, text.data_ptr(), nb)
    m = T.TSXHashMapHIP(l, 0, k); m.set_path('partitioned')
    m.countFastqDevice(text.data_ptr(), nb); m.sync(); print('normal ', n_reads, l, m.stats())
    m.clear()
    sc = TD.ShardedCounter(m, nb)
    torch.cuda.synchronize(); t0 = time.perf_counter(); sc.step(text.data_ptr(), nb); torch.cuda.synchronize()
    print('sharded', n_reads, l, round((time.perf_counter()-t0)*1e3,1), 'ms', m.stats())
    m.close(); del sc
