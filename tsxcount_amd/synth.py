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
