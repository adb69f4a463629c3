"""PCIe-inclusive rate of the host entry point (tsx_hip_count_fastq_host): the same 2.08 GB text in pageable host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tsxcount_amd as T
n_reads, k = 1087000, 31
nb, nk, _ = T.synth_sizes(20261004, 0, n_reads, k)
text = torch.empty(nb + 256, dtype=torch.uint8, device='cuda:0'); torch.cuda.synchronize()
T.synth_fastq_device(20261004, 0, n_reads, k, text.data_ptr(), nb)
host = text[:nb].cpu().numpy().tobytes()
del text
m = T.TSXHashMapHIP(30, 0, k)
for it in range(4):
    m.clear(); t0 = time.perf_counter(); m.countFastq(host); dt = time.perf_counter() - t0
    print('host entry: %.1f ms  %.2f G k-mers/s  (%.1f GB/s of text)' % (dt*1e3, nk/dt/1e9, nb/dt/1e9), m.stats()['kmers_added'] == nk, flush=True)
