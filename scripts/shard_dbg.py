import os, sys, time, ctypes
sys.path.insert(0, '.')
import torch, numpy as np
import torch.distributed as dist
import tsxcount_amd as T
from tsxcount_amd import distributed as TD
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29644')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
n_reads, l, k = 200000, 28, 31
nb, nk, _ = T.synth_sizes(5, 0, n_reads, k)
text = torch.empty(nb + 256, dtype=torch.uint8, device='cuda:0'); torch.cuda.synchronize()
T.synth_fastq_device(5, 0, n_reads, k, text.data_ptr(), nb)
m = T.TSXHashMapHIP(l, 0, k)
sc = TD.ShardedCounter(m, nb)
L = m._lib; vp = ctypes.c_void_p
T._check(L.tsx_hip_shard_scan_device(m.handle, vp(text.data_ptr()), nb, vp(sc.send.data_ptr()), sc.send.numel(), vp(sc.counts.data_ptr()), vp(sc.hot_k.data_ptr()), vp(sc.hot_c.data_ptr()), sc.HOT_CAP, vp(sc.hot_n.data_ptr()), None))
torch.cuda.synchronize()
n = int(sc.counts[0].item()); print('sent keys', n, 'hot', int(sc.hot_n.item()))
recv = torch.empty((n + 1000,), dtype=torch.int64, device='cuda:0')
dist.all_to_all_single(recv[:n], sc.send[:n], output_split_sizes=[n], input_split_sizes=[n])
torch.cuda.synchronize()
print('a2a equal:', bool(torch.equal(recv[:n], sc.send[:n])), 'mismatches', int((recv[:n] != sc.send[:n]).sum().item()))
sc2 = TD.ShardedCounter(m, nb)
m.clear(); sc2.step(text.data_ptr(), nb); print('step 1', m.stats())
m.clear(); sc2.step(text.data_ptr(), nb); print('step 2', m.stats())
dist.destroy_process_group()
