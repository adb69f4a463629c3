"""How much do the stages of the partitioned path gain from running beside each other?
Two maps (2^29 slots each), half of the bench reads each: counted one after the other on
one stream vs side by side on two streams."""
import sys, time
sys.path.insert(0, '.')
import torch, tsxcount_amd as T
reads = 543500; seed = 20261004; k = 31
maps, texts, sizes, streams = [], [], [], []
for i in range(2):
    nbytes, nk, _ = T.synth_sizes(seed, i * reads, reads, k)
    t = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda')
    T.synth_fastq_device(seed, i * reads, reads, k, t.data_ptr(), nbytes, device=0)
    m = T.TSXHashMapHIP(29, 0, k); m.set_path('partitioned')
    maps.append(m); texts.append(t); sizes.append((nbytes, nk)); streams.append(torch.cuda.Stream())
torch.cuda.synchronize()
def run(concurrent):
    for m in maps: m.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, m in enumerate(maps):
        st = streams[i].cuda_stream if concurrent else streams[0].cuda_stream
        m.countFastqDevice(texts[i].data_ptr(), sizes[i][0], stream=st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for _ in range(2): run(False); run(True)
a = min(run(False) for _ in range(4)); b = min(run(True) for _ in range(4))
print('sequential %.2f ms, two streams %.2f ms, k-mers %d' % (a, b, sum(s[1] for s in sizes)))
for m in maps:
    st = m.stats(); print(st['kmers_added'], st['distinct'], st['insert_failures'])
