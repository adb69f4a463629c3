"""Synthetic reads shaped like the reference's generateFakeSequences.py
(500-1000 random bases + 100-300 'A', '@seq<i>' header, '&' qualities).

Host (numpy) twin of synth_fill_kernel in csrc/tsx_kernels.h: both produce the
same bytes for the same (seed, first_read, n_reads), so small cases can be
checked byte for byte and large cases generated straight into HBM.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_C = np.uint64(0xD1B54A32D192ED03)


def _mix(seed, i, c):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.asarray(i, dtype=np.uint64) * _G + np.asarray(c, dtype=np.uint64) * _C
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def read_lengths(seed, first_read, n_reads):
    ids = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
    nrand = 500 + (_mix(seed, ids, 0) % np.uint64(501)).astype(np.int64)
    na = 100 + (_mix(seed, ids, 1) % np.uint64(201)).astype(np.int64)
    return nrand, na


def read_sequence(seed, read_id, nrand, na):
    j = np.arange(nrand, dtype=np.uint64)
    w = _mix(seed, np.uint64(read_id), np.uint64(2) + (j >> np.uint64(5)))
    code = ((w >> (np.uint64(2) * (j & np.uint64(31)))) & np.uint64(3)).astype(np.int64)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[code]
    return bases.tobytes() + b"A" * int(na)


def fastq(seed, first_read, n_reads):
    """The FASTQ text (bytes) of reads first_read .. first_read+n_reads-1."""
    nrand, na = read_lengths(seed, first_read, n_reads)
    parts = []
    for r in range(n_reads):
        s = read_sequence(seed, first_read + r, int(nrand[r]), int(na[r]))
        parts.append(b"@seq%d\n" % (first_read + r))
        parts.append(s)
        parts.append(b"\n+\n")
        parts.append(b"&" * len(s))
        parts.append(b"\n")
    return b"".join(parts)


def repeated_reads_fastq(seed, n_reads, lo, hi, max_copies=3):
    """Random reads of lo..hi-1 bases without poly-A tails; read i is written 1 + i % max_copies times, so every
    k-mer count stays small (<= max_copies unless two random reads share a k-mer): inputs for narrow counters
    that must not overflow."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts, r = [], 0
    for i in range(n_reads):
        n = int(rng.integers(lo, hi))
        s = lut[rng.integers(0, 4, size=n)].tobytes()
        for _ in range(1 + i % max_copies):
            parts.append(b"@r%d\n" % r + s + b"\n+\n" + b"I" * n + b"\n")
            r += 1
    return b"".join(parts)


def zipf_fastq(seed, n_reads, read_len, n_templates, k, a=1.2):
    """Zipf-skewed reads: each read is a window of one of n_templates template
    sequences picked with Zipf(a) rank weights, so a few k-mers are extremely
    hot (contention / reprobe stress, BASELINE config 4)."""
    rng = np.random.default_rng(seed)
    tl = read_len * 2
    templates = rng.integers(0, 4, size=(n_templates, tl), dtype=np.int64)
    ranks = np.arange(1, n_templates + 1, dtype=np.float64)
    w = ranks ** (-a)
    w /= w.sum()
    pick = rng.choice(n_templates, size=n_reads, p=w)
    start = rng.integers(0, tl - read_len + 1, size=n_reads)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = []
    for r in range(n_reads):
        s = lut[templates[pick[r], start[r]:start[r] + read_len]].tobytes()
        parts.append(b"@z%d\n" % r + s + b"\n+\n" + b"I" * read_len + b"\n")
    return b"".join(parts)


# ---- Zipf-skewed reads, the device generator's numpy twin (synth_zipf_kernel, csrc/tsx_kernels.h) ---------------------
_ZT = np.uint64(0x5A495046)
_ZS = np.uint64(0x7E3779B97F4A7C15)


def zipf_thresholds(n_templates, a=1.2):
    """Upper ends of the templates' shares of [0, 2^64) for Zipf(a) rank weights: ascending uint64, last = 2^64 - 1."""
    w = np.arange(1, n_templates + 1, dtype=np.float64) ** (-a)
    cdf = np.cumsum(w) / w.sum()
    thr = np.minimum(cdf * 18446744073709551616.0, 18446744073709549568.0).astype(np.uint64)
    thr[-1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return np.maximum.accumulate(thr)


def zipf_reads(seed, n_reads, read_len, thr):
    """(template, start) of every read: template = first t with thr[t] >= u_r, start = mix(seed, r, 1) % (read_len + 1)."""
    r = np.arange(n_reads, dtype=np.uint64)
    pick = np.minimum(np.searchsorted(thr, _mix(seed, r, 0), side="left"), len(thr) - 1).astype(np.int64)
    start = (_mix(seed, r, 1) % np.uint64(read_len + 1)).astype(np.int64)
    return pick, start


def zipf_template(seed, t, q0, n):
    """Bases q0 .. q0 + n - 1 of template t (bytes)."""
    q = np.arange(q0, q0 + n, dtype=np.uint64)
    w = _mix(np.uint64(seed) ^ _ZS, _ZT + np.uint64(t), np.uint64(2) + (q >> np.uint64(5)))
    code = ((w >> (np.uint64(2) * (q & np.uint64(31)))) & np.uint64(3)).astype(np.int64)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[code].tobytes()


def zipf_text(seed, n_reads, read_len, thr):
    """The FASTQ text the device generator writes (small cases)."""
    pick, start = zipf_reads(seed, n_reads, read_len, thr)
    parts = []
    for r in range(n_reads):
        parts.append(b"@z%d\n" % r + zipf_template(seed, int(pick[r]), int(start[r]), read_len) + b"\n+\n" + b"I" * read_len + b"\n")
    return b"".join(parts)


class ZipfExpect:
    """What the table must hold after counting the Zipf reads, from the generator alone: every k-mer is (template t,
    position p) -- two random templates share a k-mer of k >= 24 bases with negligible probability -- and its count is
    the number of reads of template t whose window covers p .. p + k - 1."""

    def __init__(self, seed, n_reads, read_len, thr, k):
        self.seed, self.read_len, self.k = seed, read_len, k
        self.span = read_len - k            # a read that starts at s covers k-mer positions s .. s + span
        self.pick, self.start = zipf_reads(seed, n_reads, read_len, thr)
        self.total = int(n_reads) * max(0, self.span + 1)
        self.stride = read_len + 2
        self.key = np.sort(self.pick * self.stride + self.start)      # reads ordered by (template, start)
        if self.span < 0:
            self.distinct = 0
        else:
            t, s = self.key // self.stride, self.key % self.stride
            same = t[1:] == t[:-1]
            gap = np.where(same, np.minimum(s[1:] - s[:-1], self.span + 1), self.span + 1)
            self.distinct = int(gap.sum()) + (self.span + 1 if len(t) else 0)

    def count(self, t, p):
        """Occurrences of the k-mer at position p of template t."""
        lo = np.searchsorted(self.key, t * self.stride + max(0, p - self.span), side="left")
        hi = np.searchsorted(self.key, t * self.stride + min(p, self.read_len), side="right")
        return int(hi - lo)

    def sample(self, read_ids):
        """(k-mer sequences [bytes], expected counts) of every window of the given reads (duplicates merged)."""
        out = {}
        for r in read_ids:
            t, s = int(self.pick[r]), int(self.start[r])
            seq = zipf_template(self.seed, t, s, self.read_len)
            for i in range(self.span + 1):
                out[seq[i:i + self.k]] = self.count(t, s + i)
        return list(out.keys()), np.array(list(out.values()), dtype=np.uint64)

    def hottest(self):
        """(template, position, count) of the most frequent k-mer of template 0 (the Zipf head)."""
        best = max(range(0, 2 * self.read_len - self.k + 1, 1), key=lambda p: self.count(0, p))
        return 0, best, self.count(0, best)
