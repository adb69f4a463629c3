"""One GPU playing the compute side of ONE GPU of an N-GPU step of the minimizer exchange (argv[1] = N, default 8), without
the exchange: it describes its own text (1e9 k-mer occurrences, the bench shape), splits the descriptions into N owner
lists, then walks ALL N lists -- stand-ins for the N lists it would receive, one from every GPU, each 1/N of a text --
level 2 + build, and adds the homopolymer totals.  Nothing is walked twice, so the table that comes out is the exact count
of the text: it is compared with a plain tsx_hip_count_fastq_device of the same text (argv[2] = "check").
Prints ms per step and the bytes one GPU would send."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tsxcount_amd as T
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
check = len(sys.argv) > 2 and sys.argv[2] == "check"
reads = int(os.environ.get("SIM_READS", "1087000"))
k, l, seed = int(os.environ.get("SIM_K", "31")), int(os.environ.get("SIM_L", "30")), 20261004
nb, nk, _ = T.synth_sizes(seed, 0, reads, k)
buf = torch.empty(nb + 256, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
T.synth_fastq_device(seed, 0, reads, k, buf.data_ptr(), nb)
m = T.TSXHashMapHIP(l, 0, k)
L, vp = m._lib, ctypes.c_void_p
assert L.tsx_hip_mini_supported(m.handle)
i64 = dict(dtype=torch.int64, device="cuda:0")
cnt = torch.zeros((world + 4,), **i64)
emit = torch.zeros((2,), **i64)
hom_k = torch.from_numpy(np.stack([T.encode("ACGT"[b] * k, k) for b in range(4)]).astype(np.int64)).to("cuda:0")


nwin = int(os.environ.get("SIM_WINDOWS", "4"))
pc = ctypes.c_size_t(0)
L.tsx_hip_mini_part_capacity(m.handle, nb + 256, nwin, ctypes.byref(pc))
cap = pc.value
dsc = torch.empty((2 * cap * world,), **i64)
recv = [torch.empty((2 * cap * (world if reads < 100000 else 1),), **i64) for _ in range(nwin)]      # what the all-to-all of a window would deliver: N lists back to back
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * nwin)]
copy_ms = []
side = torch.cuda.Stream()      # one stream for the library's kernels and the stand-in copies (the default stream's handle is 0 = "the map's own")


def step():
    with torch.cuda.stream(side):
        return step_on(vp(side.cuda_stream))


def step_on(cs):
    m.clear()
    emit.zero_()
    tot = [0] * (world + 4)
    ms = 0.0
    rc = L.tsx_hip_mini_describe_device(m.handle, vp(buf.data_ptr()), nb, 0, nb, vp(emit.data_ptr()), cs)   # the text is described once
    assert rc == 0, rc
    for i in range(nwin):
        rc = L.tsx_hip_mini_split_device(m.handle, i, nwin, world, vp(dsc.data_ptr()), cap, vp(cnt.data_ptr()), cs)
        assert rc == 0, rc
        c = [int(x) for x in cnt.tolist()]
        assert max(c[:world]) <= cap and 2 * sum(c[:world]) <= recv[i].numel()
        # stand-in for the all-to-all (not part of the compute side: timed and taken off)
        ev[2 * i].record()
        at = 0
        for s in range(world):
            recv[i][2 * at:2 * (at + c[s])].copy_(dsc[2 * s * cap:2 * (s * cap + c[s])])
            at += c[s]
        ev[2 * i + 1].record()
        est = int(nk * 1.1) + 65536
        rc = L.tsx_hip_shard_walk_device(m.handle, vp(recv[i].data_ptr()), at, 2, i, nwin, est, vp(emit[1:].data_ptr()), cs)
        assert rc == 0, rc
        tot = [a + b for a, b in zip(tot, c)]
    rc = L.tsx_hip_shard_build_l1_device(m.handle, cs)
    assert rc == 0, rc
    homs = torch.tensor(tot[world:], **i64)
    rc = L.tsx_hip_add_kmers_device(m.handle, vp(hom_k.data_ptr()), vp(homs.data_ptr()), 4, cs)
    assert rc == 0, rc
    m.sync()
    torch.cuda.synchronize()
    copy_ms.append(sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(nwin)))
    return tot


for _ in range(2):
    c = step()
copy_ms.clear()
t0 = time.perf_counter()
for _ in range(5):
    c = step()
dt = (time.perf_counter() - t0) / 5 - sum(copy_ms) / 5 / 1e3
nd = sum(c[:world])
em = [int(x) for x in emit.tolist()]
print("minimizer exchange, world=%d, k=%d, %d windows: %.2f ms per step (describe + split + walks + level 2 + build; the copies that stand in for the all-to-all, %.2f ms, taken off) for %d k-mers = %.1f G/s per GPU; "
      "%d descriptions (%.2f per strip with a start, %.2f B per k-mer), lists %s, homopolymers %s; described %d = walked %d + homopolymers %d: %s"
      % (world, k, nwin, dt * 1e3, sum(copy_ms) / 5, nk, nk / dt / 1e9, nd, nd / (em[0] / 16.0), nd * 16.0 / nk, c[:world], c[world:], em[0], em[1], sum(c[world:]),
         "ok" if em[0] == em[1] + sum(c[world:]) else "MISMATCH"))
print("one GPU of %d sends %.2f GB per step ((N-1)/N of its lists)" % (world, nd * 16 * (world - 1) / world / 1e9))
st = m.stats()
print("stats:", st)
# the one-GPU step on the same box (the boxes differ by a few per cent): the plain path over the same text
ref = T.TSXHashMapHIP(l, 0, k)
for _ in range(2):
    ref.clear(); ref.countFastqDevice(buf.data_ptr(), nb); ref.sync()
t0 = time.perf_counter()
for _ in range(5):
    ref.clear(); ref.countFastqDevice(buf.data_ptr(), nb); ref.sync()
plain = (time.perf_counter() - t0) / 5
share = max(c[:world]) / (sum(c[:world]) / world)
print("plain one-GPU step on this box: %.2f ms; compute-side efficiency of one GPU of %d: %.1f %% (%.1f %% if the fullest list, %.3f x the mean, "
      "sets the pace of walk + level 2 + build)" % (plain * 1e3, world, 100 * plain / dt, 100 * plain / (dt + (share - 1) * 0.75 * dt), share))
if check:
    rs = ref.stats()
    print("plain:", rs)
    assert rs["distinct"] == st["distinct"] and rs["count_sum"] == st["count_sum"] and st["insert_failures"] == 0
    # the first reads' k-mers: equal counts in both tables
    txt = bytes(buf[:200000].cpu().numpy()).split(b"\n")
    seqs = [txt[i] for i in range(1, len(txt) - 1, 4)][:40]
    kms = sorted({s[i:i + k] for s in seqs for i in range(len(s) - k + 1)})
    enc = T.encode_many(kms, k)
    a, b = m.getKmerCounts(enc), ref.getKmerCounts(enc)
    assert np.array_equal(a, b) and a.min() >= 1, "counts differ"
    print("check: %d k-mers of the first reads equal in both tables (max count %d)" % (len(kms), int(a.max())))
