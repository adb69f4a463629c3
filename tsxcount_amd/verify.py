"""--check for the synthetic workload (bench.py, BASELINE-size tests).

The reference's --check (src/mains/main.cpp:224-396) walks a `.count` file and compares
getKmerCount(kmer) for every k-mer.  No such file exists for 1e9 synthetic k-mers, so the
same question -- does the table hold the right count for every k-mer? -- is answered from
what is known about the input without counting it a second time on the CPU:

  totals     the sum of all counts read back from the table equals the number of k-mers
             in the reads (nothing lost, nothing counted twice), failure counters are 0;
  polyA      getKmerCount("A" * k) equals the count computed from the generator;
  sample     every k-mer of a deterministic sample of reads is looked up: its count must be
             >= its multiplicity inside the sample, and EQUAL to it when the window holds at
             least SAFE_RANDOM_BASES random bases (such a k-mer recurs in another of R reads
             with probability < R / 4^24: 4e-9 per window at 1.1e6 reads, 3e-3 over a whole
             sample of 1000 reads);
  cross      (one GPU) the same text is counted through the OTHER insert path into a
             second table; every entry of a sample of slot ranges of the first table must
             have the same count there, and both tables must agree on distinct and totals.

Pure numpy + the C ABI; nothing here is a CPU counting path of the product.
"""
import numpy as np

from . import synth

SAFE_RANDOM_BASES = 24


def kmers_of(seq, k):
    """All k-mers of one read as [n_windows, key_limbs] uint64 limbs (UBigInt layout: base i in
    bits 2i,2i+1; SequenceUtils.h:86-160), with the product's byte code ((b>>1)^(b>>2))&3."""
    b = np.frombuffer(seq, dtype=np.uint8).astype(np.uint64)
    wk = (2 * k + 63) // 64
    n = len(b) - k + 1
    if n <= 0:
        return np.zeros((0, wk), dtype=np.uint64)
    code = ((b >> np.uint64(1)) ^ (b >> np.uint64(2))) & np.uint64(3)
    win = np.lib.stride_tricks.sliding_window_view(code, k)       # [n, k]
    out = np.zeros((n, wk), dtype=np.uint64)
    for t in range(wk):
        cols = win[:, 32 * t: min(k, 32 * (t + 1))]
        sh = (np.arange(cols.shape[1], dtype=np.uint64) * np.uint64(2))
        out[:, t] = np.bitwise_or.reduce(cols << sh, axis=1)
    return out


def sample_expectations(seed, k, read_ids):
    """(kmers [m, wk], in-sample multiplicity [m], safe [m]) over the k-mers of the sampled reads."""
    rows, safe = [], []
    for r in read_ids:
        nrand, na = synth.read_lengths(seed, int(r), 1)
        s = synth.read_sequence(seed, int(r), int(nrand[0]), int(na[0]))
        km = kmers_of(s, k)
        rows.append(km)
        i = np.arange(km.shape[0])
        safe.append(np.minimum(k, np.maximum(0, int(nrand[0]) - i)) >= min(k, SAFE_RANDOM_BASES))
    allk = np.concatenate(rows) if rows else np.zeros((0, (2 * k + 63) // 64), dtype=np.uint64)
    alls = np.concatenate(safe) if safe else np.zeros((0,), dtype=bool)
    if k < SAFE_RANDOM_BASES:
        alls[:] = False     # short k-mers repeat by chance: only the lower bound holds
    uniq, inv, cnt = np.unique(allk, axis=0, return_inverse=True, return_counts=True)
    usafe = np.zeros(len(uniq), dtype=bool)
    np.logical_or.at(usafe, inv.ravel(), alls)
    return uniq, cnt.astype(np.uint64), usafe


def sample_read_ids(first, n_reads, n_sample):
    """Deterministic spread over the shard, both ends included."""
    n_sample = min(n_sample, n_reads)
    if n_sample <= 0:
        return np.zeros((0,), dtype=np.int64)
    return first + np.unique(np.linspace(0, n_reads - 1, n_sample).astype(np.int64))


def judge_sample(got, mult, safe):
    """-> dict with the numbers of violated lower bounds and of unequal safe k-mers."""
    got = np.asarray(got, dtype=np.uint64)
    return {"looked_up": int(len(got)), "below_sample_multiplicity": int((got < mult).sum()),
            "safe_kmers": int(safe.sum()), "safe_unequal": int((got[safe] != mult[safe]).sum())}


def cross_check(m, text_ptr, nbytes, other_path, n_ranges=16, range_slots=1 << 20, device=0):
    """Count the same device text through `other_path` into a second table and compare it with
    slot-range samples of `m`'s dump, entry by entry, on the device.  Returns a dict."""
    import torch
    from . import TSXHashMapHIP
    lay = m.layout
    m2 = TSXHashMapHIP(lay.l, lay.count_bits, m.k, device=device)   # same slot layout as m
    try:
        m2.set_path(other_path)
        m2.countFastqDevice(text_ptr, nbytes)
        m2.sync()
        s1, s2 = m.stats(), m2.stats()
        slots = int(lay.slots)
        range_slots = min(range_slots, slots)
        n_ranges = max(1, min(n_ranges, slots // range_slots))
        dev = torch.device("cuda", device)
        kbuf = torch.empty((range_slots, m.wk), dtype=torch.int64, device=dev)
        cbuf = torch.empty((range_slots,), dtype=torch.int64, device=dev)
        obuf = torch.empty((range_slots,), dtype=torch.int64, device=dev)
        nbuf = torch.zeros((1,), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        compared = unequal = 0
        for i in range(n_ranges):
            lo = (slots - range_slots) * i // max(1, n_ranges - 1) if n_ranges > 1 else 0
            m.dumpRangeDevice(lo, lo + range_slots, kbuf.data_ptr(), cbuf.data_ptr(), range_slots, nbuf.data_ptr())
            n = int(nbuf.item())
            if n == 0:
                continue
            m2.getKmerCountsDevice(kbuf.data_ptr(), n, obuf.data_ptr())
            m2.sync()
            compared += n
            unequal += int((obuf[:n] != cbuf[:n]).sum().item())
        return {"other_path": other_path, "entries_compared": compared, "entries_unequal": unequal,
                "distinct_equal": s1["distinct"] == s2["distinct"], "count_sum_equal": s1["count_sum"] == s2["count_sum"],
                "other_failures": s2["insert_failures"] + s2["overflow_failures"] + s2["lock_timeouts"],
                "ok": unequal == 0 and compared > 0 and s1["distinct"] == s2["distinct"] and
                      s1["count_sum"] == s2["count_sum"] and
                      s2["insert_failures"] + s2["overflow_failures"] + s2["lock_timeouts"] == 0}
    finally:
        m2.close()
