"""Scale validation beyond the bench shape: several device windows (> 4 GiB of text), a table
too large for the partitioned path (atomic path), totals checked against the generator."""
import sys, time
sys.path.insert(0, '.')
import torch
import tsxcount_amd as T
n_reads, k, l = int(sys.argv[1]), 31, int(sys.argv[2])
t0 = time.time(); nb, nk, npolya = T.synth_sizes(7, 0, n_reads, k, want_polya=True); print('sizing %.1f s' % (time.time()-t0), nb, nk, npolya, flush=True)
text = torch.empty(nb + 256, dtype=torch.uint8, device='cuda:0'); torch.cuda.synchronize()
T.synth_fastq_device(7, 0, n_reads, k, text.data_ptr(), nb)
m = T.TSXHashMapHIP(l, 0, k)
print('layout', {f: getattr(m.layout, f) for f, _ in m.layout._fields_}, flush=True)
for it in range(2):
    m.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    m.countFastqDevice(text.data_ptr(), nb); m.sync(); dt = time.perf_counter() - t0
    st = m.stats()
    print('run %d: %.1f ms  %.2f G k-mers/s' % (it, dt*1e3, nk/dt/1e9), st, flush=True)
    assert st['kmers_added'] == nk and st['insert_failures'] == 0 and st['overflow_failures'] == 0
    assert m.getKmerCount(T.encode('A' * k)) == npolya
print('SCALE OK')
