"""The oracle (oracle/tsx_oracle.c) against the reference's own fixture, an
independent dictionary count, and recorded runs of the real reference binary."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, python_counts
from oracle.oracle import Oracle
from tsxcount_amd import synth


@pytest.mark.parametrize("l,s", [(26, 4), (18, 4), (19, 2), (20, 1), (22, 8)])
def test_oracle_matches_reference_fixture(golden_fastq, golden_counts, l, s):
    # the reference's README test: k=14, defaults l=26 s=4 (main.cpp:411-413)
    o = Oracle(14, l, s, seed=7)
    assert o.count_fastq(golden_fastq) == 202204
    assert o.distinct() == len(golden_counts) == 194697
    bad = sum(1 for kmer, c in golden_counts.items() if o.get_count(o.encode(kmer)) != c)
    assert bad == 0
    kmers, counts = o.dump()
    assert int(counts.sum()) == 202204


@pytest.mark.parametrize("k,l,s", [(5, 9, 2), (14, 16, 3), (31, 18, 4), (32, 18, 2), (33, 18, 4),
                                   (63, 18, 4), (64, 18, 5), (127, 18, 4)])
def test_oracle_matches_dictionary_count(k, l, s):
    text = synth.fastq(3, 0, 40)
    if k == 5:
        text = text[:300] + b"\n"  # 4^5 = 1024 possible 5-mers; keep the 512-slot table under-full
    ref = python_counts(text, k)
    o = Oracle(k, l, s, seed=k)
    assert o.count_fastq(text) == sum(ref.values())
    assert o.distinct() == len(ref)
    items = list(ref.items())
    rng = np.random.default_rng(0)
    for i in rng.choice(len(items), size=min(3000, len(items)), replace=False):
        kmer, c = items[i]
        assert o.get_count(o.encode(kmer)) == c
    # the hot polyA k-mer walks the whole overflow chain
    hot = max(ref, key=ref.get)
    assert o.get_count(o.encode(hot)) == ref[hot]
    probe = (b"ACGT" * 40)[:k]
    assert o.get_count(o.encode(probe)) == ref.get(probe, 0)


def test_oracle_hash_is_bijective_and_triangular():
    o = Oracle(31, 20, 4, seed=9)
    rows = o.hash_rows()
    n = 62
    # unit upper triangular: row i has bit n-1-i set and nothing above it
    for i in range(n):
        v = int(rows[i, 0])
        assert (v >> (n - 1 - i)) & 1 == 1
        assert v >> (n - i) == 0
    rng = np.random.default_rng(1)
    for _ in range(200):
        x = rng.integers(0, 1 << 62, dtype=np.uint64).reshape(1)
        assert (o.hash_invert(o.hash_apply(x)) == x).all()
    # linear over GF(2)
    a = rng.integers(0, 1 << 62, dtype=np.uint64).reshape(1)
    b = rng.integers(0, 1 << 62, dtype=np.uint64).reshape(1)
    assert (o.hash_apply(a ^ b) == (o.hash_apply(a) ^ o.hash_apply(b))).all()


def test_oracle_record_rules():
    # empty lines are skipped, reads shorter than k give nothing, last line may lack '\n'
    text = b"@r1\n\nACGTACGT\n+\n\n!!!!!!!!\n@r2\nACG\n+\n!!!\n@r3\nTTTTTTTT\n+\n!!!!!!!!"
    o = Oracle(4, 6, 4, seed=1)
    assert o.count_fastq(text) == 10
    assert o.get_count(o.encode("ACGT")) == 2
    assert o.get_count(o.encode("TTTT")) == 5
    assert o.get_count(o.encode("!!!!")) == 0


def test_oracle_rejects_bad_geometry():
    with pytest.raises(ValueError):
        Oracle(10, 20, 4)  # 2k <= l, TSXHashMap.h:91-94


def test_oracle_against_recorded_reference_runs():
    """tests/golden/ref_runs.json holds runs of the REAL reference binary
    (oracle/_ref/tsxCount_ref --check) on .count files written from this
    oracle: the reference reported 0 errors and the same distinct count.  Here
    the oracle must still produce exactly those count files (sha256)."""
    path = os.path.join(GOLDEN, "ref_runs.json")
    runs = json.load(open(path))["runs"]
    assert len(runs) >= 3
    for r in runs:
        assert r["reference_total_errors"] == 0
        if r["input"] == "golden":
            text = open(os.path.join(GOLDEN, "small_t7.1000.fastq"), "rb").read()
        else:
            text = synth.fastq(r["seed"], 0, r["n_reads"])
        o = Oracle(r["k"], r["l"], r["s"], seed=1)
        o.count_fastq(text)
        assert o.distinct() == r["reference_distinct"] == r["oracle_distinct"]
        kmers, counts = o.dump()
        order = np.lexsort(kmers.T[::-1])
        h = hashlib.sha256()
        h.update(kmers[order].tobytes())
        h.update(counts[order].tobytes())
        assert h.hexdigest() == r["oracle_counts_sha256"]


def perf_case_text(r):
    if r["input"] == "repeated":
        return synth.repeated_reads_fastq(r["seed"], r["n_reads"], r["lo"], r["hi"])
    return synth.fastq(r["seed"], 0, r["n_reads"])


def perf_digest(pairs):
    """sha256 over the sorted `kmer<TAB>count` lines (what oracle/_ref/ref_perf_driver prints)."""
    h = hashlib.sha256()
    for kmer, cnt in sorted(pairs):
        h.update(b"%s\t%d\n" % (kmer, cnt))
    return h.hexdigest()


def test_oracle_against_the_reference_serial_table_k40_to_63():
    """ref_runs.json "perf_runs": the reference's OWN serial table (TSXHashMapPerf::addKmer / getKmerCount(kmer),
    driven by oracle/ref_perf_driver.cpp over the reference's reader and fromSequence) counted these texts at
    k = 40, 47, 55, 63 (and 33); the sha256 of its sorted `kmer<TAB>count` output is recorded.  It answers every
    k-mer whose counter has not overflowed (the few that have are listed as `reference_thrown`: the reference
    throws from its overflow walk for k >= 40).  The restatement must give exactly that output."""
    import tsxcount_amd as T
    runs = json.load(open(os.path.join(GOLDEN, "ref_runs.json")))["perf_runs"]
    assert sorted({r["k"] for r in runs}) == [33, 40, 47, 55, 63]
    for r in runs:
        o = Oracle(r["k"], r["l"], r["s"], seed=1)
        o.count_fastq(perf_case_text(r))
        kmers, counts = o.dump()
        thrown = set(r["reference_thrown"])
        pairs = [(T.decode(kmers[i], r["k"]).encode(), int(counts[i])) for i in range(len(kmers))]
        pairs = [x for x in pairs if x[0].decode() not in thrown]
        assert len(pairs) == r["reference_answered"]
        assert perf_digest(pairs) == r["reference_sha256"], r["command"]


def test_fasta_records_two_lines_per_record():
    """FASTXreader<FASTAEntry> (FastXReader.h:97-116): every two non-empty lines are a record, the second is
    the sequence.  The restatement against an independent dictionary count."""
    import numpy as np
    from conftest import python_counts
    from oracle.oracle import Oracle
    rng = np.random.default_rng(4)
    recs = []
    for r in range(60):
        n = int(rng.integers(1, 120))
        recs.append(b">r%d some description\n" % r + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)) + b"\n")
        if r % 7 == 0:
            recs.append(b"\n")
    text = b"".join(recs)
    for k in (8, 21, 33):
        exp = python_counts(text, k, 2)
        o = Oracle(k, min(16, 2 * k - 1), 4, seed=1)
        assert o.count_fastq(text, 2) == sum(exp.values())
        assert o.distinct() == len(exp)
        for kmer, c in list(exp.items())[:200]:
            assert o.get_count(o.encode(kmer)) == c

