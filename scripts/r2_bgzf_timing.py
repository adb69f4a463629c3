"""Wall time of the BGZF entry point (inflate on the device + count) against the plain host entry point, and of
zlib on the host, for the same synthetic FASTQ."""
import ctypes, gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tsxcount_amd as T
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k, l, seed = 31, (int(sys.argv[2]) if len(sys.argv) > 2 else 28), 20261004
nb, nk, _ = T.synth_sizes(seed, 0, reads, k)
buf = torch.empty(nb + 256, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
T.synth_fastq_device(seed, 0, reads, k, buf.data_ptr(), nb)
text = bytes(buf[:nb].cpu().numpy())
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
t0 = time.perf_counter(); z = T.bgzf_compress(text, level=level); t1 = time.perf_counter()
print("text %d bytes, BGZF %d bytes (%.2fx), %d members; compress %.1f s" % (nb, len(z), nb / len(z), T.bgzf_index(z)[0], t1 - t0))
t0 = time.perf_counter(); back = gzip.decompress(z); t1 = time.perf_counter()
assert back == text
print("zlib inflate on one host core: %.3f s = %.2f GB/s of text" % (t1 - t0, nb / (t1 - t0) / 1e9))
m = T.TSXHashMapHIP(l, 0, k)
for name, fn, arg in (("plain host text", m.countFastq, text), ("BGZF, inflated on the device", m.countFastqBgzf, z)):
    best = 1e9
    for _ in range(3):
        m.clear(); m.sync()
        t0 = time.perf_counter(); fn(arg); m.sync(); best = min(best, time.perf_counter() - t0)
    st = m.stats()
    print("%-30s %.1f ms  (%d k-mers, %.2f G k-mers/s, %.2f GB/s of text)" % (name, best * 1e3, st["kmers_added"], st["kmers_added"] / best / 1e9, nb / best / 1e9))
t0 = time.perf_counter(); out = T.bgzf_inflate(z); t1 = time.perf_counter()
assert out == text
print("inflate only (device, incl. copies both ways): %.1f ms" % ((t1 - t0) * 1e3))
