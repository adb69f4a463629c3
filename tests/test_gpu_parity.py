"""Parity of the HIP path (through the C ABI) with the oracle, the reference's
golden fixture and size-independent properties.  Bit-exact: counts are integers."""
import ctypes
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, python_counts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tsxcount_amd
    assert os.path.exists(tsxcount_amd.LIB_PATH), "HIP extension missing: no fallback"
    assert tsxcount_amd.lib().tsx_hip_device_count() > 0, "no GPU"
    return tsxcount_amd


def oracle_for(text, k, l=20, s=4):
    from oracle.oracle import Oracle
    o = Oracle(k, l, s, seed=1)
    n = o.count_fastq(text)
    return o, n


def assert_same_as_oracle(T, text, k, l, s=0, path="auto", **kw):
    o, n = oracle_for(text, k, min(max(l, 16), 24, 2 * k - 1), 4)  # the oracle's own table size is free
    m = T.TSXHashMapHIP(l, s, k, **kw)
    m.set_path(path)
    try:
        m.countFastq(text)
        st = m.stats()
        assert st["kmers_added"] == n
        assert st["distinct"] == o.distinct()
        assert st["insert_failures"] == 0 and st["overflow_failures"] == 0 and st["lock_timeouts"] == 0
        kmers, counts = o.dump()
        if len(kmers):
            assert np.array_equal(m.getKmerCounts(kmers), counts)
        # and the other direction: everything the table holds is in the oracle
        gk, gc = m.getAllKmers()
        assert len(gk) == len(kmers)
        assert int(gc.sum()) == n
        if len(gk):
            a = np.lexsort(gk.T[::-1])
            b = np.lexsort(kmers.T[::-1])
            assert np.array_equal(gk[a], kmers[b]) and np.array_equal(gc[a], counts[b])
        return st
    finally:
        m.close()


# --- the reference's own fixture -------------------------------------------------

@pytest.mark.parametrize("l,s", [(26, 4), (20, 0), (18, 2), (18, 1)])
def test_golden_fixture(T, golden_fastq, golden_counts, l, s):
    m = T.TSXHashMapHIP(l, s, 14)
    m.countFastq(golden_fastq)
    st = m.stats()
    assert st["kmers_added"] == 202204 and st["distinct"] == 194697
    kmers = T.encode_many(list(golden_counts.keys()), 14)
    exp = np.array(list(golden_counts.values()), dtype=np.uint64)
    assert np.array_equal(m.getKmerCounts(kmers), exp)
    m.close()


def test_cli_check_passes_on_golden(T, tmp_path):
    """tsxCount --input=... --mode=HIP --check, the README's test (README.md:47-52)."""
    fq = tmp_path / "small_t7.1000.fastq"
    fq.write_bytes(open(os.path.join(GOLDEN, "small_t7.1000.fastq"), "rb").read())
    with gzip.open(os.path.join(GOLDEN, "small_t7.1000.fastq.14.count.gz"), "rb") as f:
        (tmp_path / "small_t7.1000.fastq.14.count").write_bytes(f.read())
    exe = os.path.join(ROOT, "tsxcount_amd", "bin", "tsxCount")
    p = subprocess.run([exe, "--input=%s" % fq, "--mode=HIP", "--check", "--checkabort"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert "Added a total of 194697 different kmers" in out
    assert "total errors0" in out
    assert "Reference kmer count: 194697" in out


def test_reference_binary_checks_hip_counts(T, tmp_path):
    """The REAL reference (oracle/_ref/tsxCount_ref, built from /root/reference by
    oracle/Makefile) counts the same FASTQ in CAS mode and --check's its table against a
    .count file written FROM THE HIP TABLE: 'total errors0' and equal distinct counts
    mean its getKmerCount agrees with ours for every k-mer.  threads=1, 2k+s = 64: the
    configuration in which the reference's byte-wise CAS is exact."""
    import re
    from tsxcount_amd import synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "tsxCount_ref")
    if not os.path.exists(ref_bin):
        pytest.skip("reference binary not built (oracle/Makefile needs /root/reference)")
    k = 31
    text = synth.fastq(77, 0, 40)
    m = T.TSXHashMapHIP(18, 0, k)
    m.countFastq(text)
    kmers, counts = m.getAllKmers()
    fq = tmp_path / "in.fastq"
    fq.write_bytes(text)
    with open(str(fq) + ".%d.count" % k, "w") as f:
        for i in range(len(kmers)):
            f.write("%s\t%d\n" % (T.decode(kmers[i], k), int(counts[i])))
    p = subprocess.run([ref_bin, "--input=%s" % fq, "--k=%d" % k, "--l=18", "--s=2", "--mode=CAS", "--threads=1",
                        "--check"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0
    assert int(re.search(r"total errors(\d+)", out).group(1)) == 0
    assert int(re.search(r"Added a total of (\d+) different kmers", out).group(1)) == len(kmers) == m.stats()["distinct"]
    m.close()


def test_reference_main_with_hip_subclass_check_passes(T, tmp_path):
    """The reference's OWN main.cpp -- its FASTXreader, createKMers, TSXSeqUtils::fromSequence, its
    countKMers loop and its --check loop (main.cpp:104-396) -- compiled where it lies with --mode=HIP added
    (oracle/ref_hip_mode.sed) and `class TSXHashMapHIP : public TSXHashMap`
    (tsxcount_amd/host/ref_binding/TSXHashMapHIP.h) over the C ABI: every addKmer / getKmerCount(kmer) /
    getKmerCountDebug goes through the virtual interface into the HIP table.  'total errors0' is printed by
    the reference's code, against the reference's own golden .count file."""
    import re
    exe = os.path.join(ROOT, "oracle", "_ref", "tsxCount_ref_hip")
    if not os.path.exists(exe):
        pytest.skip("reference-linked binary not built (oracle/Makefile needs /root/reference)")
    fq = tmp_path / "small_t7.1000.fastq"
    fq.write_bytes(open(os.path.join(GOLDEN, "small_t7.1000.fastq"), "rb").read())
    with gzip.open(os.path.join(GOLDEN, "small_t7.1000.fastq.14.count.gz"), "rb") as f:
        (tmp_path / "small_t7.1000.fastq.14.count").write_bytes(f.read())
    p = subprocess.run([exe, "--input=%s" % fq, "--k=14", "--l=20", "--s=4", "--mode=HIP", "--threads=2", "--check"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-2000:] + p.stderr.decode(errors="replace")[-2000:]
    assert "Creating TSXHashMap HIP" in p.stderr.decode(errors="replace")
    assert "Added a total of 194697 different kmers" in out
    assert int(re.search(r"total errors(\d+)", out).group(1)) == 0
    assert "Reference kmer count: 194697" in out
    assert "tsxCount kmer count: 194697" in out
    # every slot the check visited is a k-mer start and vice versa (main.cpp:381-388)
    assert "queried kmer count: 194697" in out and "queried (Xor) kmer count: 0" in out


@pytest.mark.parametrize("path", ["atomic", "partitioned"])
def test_hip_against_the_reference_serial_table_k40_to_63(T, path):
    """The HIP table against the recorded output of the reference's own serial table (tests/golden/ref_runs.json
    "perf_runs", oracle/ref_perf_driver.cpp) for k = 40, 47, 55, 63 (two-limb keys) and 33: the sorted
    `kmer<TAB>count` dump of the HIP table, without the k-mers the reference could not answer (its overflow walk
    throws for k >= 40), must hash to what the reference printed.  Both insert paths."""
    import json
    from test_oracle import perf_case_text, perf_digest
    runs = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_runs.json")))["perf_runs"]
    for r in runs:
        m = T.TSXHashMapHIP(r["l"], 0, r["k"])
        m.set_path(path)
        m.countFastq(perf_case_text(r))
        kmers, counts = m.getAllKmers()
        thrown = set(r["reference_thrown"])
        pairs = [(T.decode(kmers[i], r["k"]).encode(), int(counts[i])) for i in range(len(kmers))]
        pairs = [x for x in pairs if x[0].decode() not in thrown]
        assert len(pairs) == r["reference_answered"]
        assert perf_digest(pairs) == r["reference_sha256"], r["command"]
        m.close()


def test_lookup_with_slot_and_kmer_starts(T):
    """getKmerCountDebug (TSXHashMap.h:477-545) and getKmerStarts (:650-658) over the C ABI."""
    from tsxcount_amd import synth
    text = synth.fastq(21, 0, 30)
    m = T.TSXHashMapHIP(16, 0, 31)
    m.countFastq(text)
    kmers, counts = m.getAllKmers()
    got, slots = m.getKmerCountDebug(kmers)
    assert np.array_equal(got, counts)
    starts = m.getKmerStarts()
    assert starts.sum() == len(kmers) == m.stats()["distinct"]
    assert len(np.unique(slots)) == len(kmers) and starts[slots.astype(np.int64)].all()
    absent = T.encode("ACGT" * 7 + "ACG")
    c, sl = m.getKmerCountDebug(absent)
    assert int(c[0]) == 0 and int(sl[0]) == 2 ** 64 - 1
    m.close()


# --- oracle parity over k, table geometry and slot width ------------------------------

@pytest.mark.parametrize("k,l,s,reads", [
    (5, 9, 0, 0), (14, 16, 4, 40), (21, 18, 0, 150), (31, 18, 0, 150), (31, 20, 4, 400), (32, 18, 0, 150),
    (32, 12, 0, 2), (33, 18, 0, 150), (47, 18, 6, 150), (63, 18, 0, 150), (64, 19, 0, 150), (65, 18, 3, 150),
    (95, 18, 0, 150), (96, 18, 0, 150), (97, 18, 0, 150), (127, 18, 0, 150), (127, 19, 2, 300)])
def test_parity_with_oracle(T, k, l, s, reads):
    from tsxcount_amd import synth
    if reads:
        text = synth.fastq(100 + k, 0, reads)
    else:  # k = 5: at most ~390 distinct 5-mers for the 512-slot table
        text = synth.fastq(100 + k, 0, 1)[:400] + b"\n"
    assert_same_as_oracle(T, text, k, l, s)


def test_other_hash_seeds_give_the_same_counts(T):
    from tsxcount_amd import synth
    text = synth.fastq(4, 0, 80)
    for seed in (2, 12345, 2 ** 63 + 11):
        assert_same_as_oracle(T, text, 31, 18, 0, hash_seed=seed)


def test_mapping_is_bijective_linear_and_mixes_all_bits(T):
    """IBijectiveFunction contract (IBijectiveFunction.h:26-27): apply/inv_apply are inverse,
    GF(2)-linear.  Unlike the reference's unit upper triangular matrix (the oracle's family:
    output bit p ignores input bits above p) the product's dense matrix (multiplication in
    GF(2^2k) for k <= 32, L*U above) lets the top key bits reach the slot index, so k-mers that
    share a prefix do not share a home slot."""
    from oracle.oracle import Oracle
    for k in (14, 31, 63, 127):
        m = T.TSXHashMapHIP(12, 0, k, hash_seed=5)
        n = 2 * k
        rng = np.random.default_rng(k)
        top = np.uint64((1 << (n % 64 or 64)) - 1)

        def rnd():
            x = rng.integers(0, 2 ** 63, size=m.wk, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=m.wk, dtype=np.uint64)
            x[-1] &= top
            return x
        for _ in range(20):
            x, y = rnd(), rnd()
            assert np.array_equal(m.hash_invert(m.hash_apply(x)), x)
            assert np.array_equal(m.hash_apply(m.hash_invert(x)), x)
            assert np.array_equal(m.hash_apply(x ^ y), m.hash_apply(x) ^ m.hash_apply(y))
        # flipping the LAST base of the k-mer (top two key bits) moves the home slot
        moved = 0
        for _ in range(64):
            x = rnd()
            y = x.copy()
            y[(n - 1) >> 6] ^= np.uint64(1) << np.uint64((n - 1) & 63)
            moved += int((int(m.hash_apply(x)[0]) ^ int(m.hash_apply(y)[0])) & 0xFFF != 0)
        assert moved == 64  # one fixed non-zero column of the matrix: all or nothing
        # the reference's family, for contrast: the low bits never see the top bit
        o = Oracle(k, 12, 4, seed=5)
        x = rnd(); y = x.copy(); y[(n - 1) >> 6] ^= np.uint64(1) << np.uint64((n - 1) & 63)
        assert (int(o.hash_apply(x)[0]) ^ int(o.hash_apply(y)[0])) & 0xFFF == 0
        m.close()


def test_prefix_skew_does_not_exhaust_reprobes(T):
    """AT-rich reads: thousands of k-mers share their first 12-15 bases.  With the reference's
    triangular matrix they would share one home slot and one probe chain; at load 0.5 the
    8-bit reprobe field must still be enough, on both insert paths."""
    rng = np.random.default_rng(1)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    codes = rng.choice(4, size=(8000, 300), p=[0.4, 0.1, 0.1, 0.4])
    text = b"".join(b"@r%d\n" % r + lut[codes[r]].tobytes() + b"\n+\n" + b"I" * 300 + b"\n" for r in range(8000))
    for path in ("atomic", "partitioned"):
        m = T.TSXHashMapHIP(22, 0, 31)
        m.set_path(path)
        m.countFastq(text)
        st = m.stats()
        assert st["insert_failures"] == 0 and st["kmers_added"] == 8000 * 270
        assert 0.45 < st["distinct"] / (1 << 22) < 0.55
        m.close()


# --- the partitioned insert path (key log -> radix partition -> LDS segment build) ----------

@pytest.mark.parametrize("k,l,s,reads", [
    (31, 15, 0, 15), (31, 16, 0, 40), (31, 18, 0, 150), (31, 20, 4, 400), (31, 23, 0, 1500), (31, 24, 0, 2500),
    (14, 18, 4, 100), (21, 17, 2, 60), (32, 19, 0, 400), (27, 22, 0, 1000), (31, 31, 0, 2000), (27, 32, 0, 1500),
    # multi-limb keys (k > 32): 2-, 3- and 4-limb records and slots; segments of 2^13 / 2^12 slots
    (33, 18, 0, 150), (33, 30, 0, 300), (47, 18, 6, 150), (63, 18, 0, 150), (63, 23, 0, 2500), (64, 19, 0, 300),
    (65, 18, 3, 150), (95, 18, 0, 150), (96, 20, 0, 600), (97, 18, 0, 150), (127, 18, 0, 150), (127, 19, 2, 300),
    (127, 22, 0, 2500), (31, 20, 32, 300)])
def test_partitioned_path_parity_with_oracle(T, k, l, s, reads):
    """One radix level up to 2^8 segments, two levels above (l = 31, 32 at k <= 32: 512 lists per level).  Same counts as the oracle and
    as the atomic path, also when the same text is counted twice into the table
    (second pass merges into segments that already hold data)."""
    from tsxcount_amd import synth
    text = synth.fastq(200 + k + l, 0, reads)
    assert_same_as_oracle(T, text, k, l, s, path="partitioned")
    o, n = oracle_for(text, k, min(max(l, 16), 24, 2 * k - 1), 4)
    kmers, counts = o.dump()
    m = T.TSXHashMapHIP(l, s, k)
    m.set_path("partitioned")
    m.countFastq(text)
    m.set_path("atomic")
    m.countFastq(text)          # atomic inserts into segments the build wrote
    m.set_path("partitioned")
    m.countFastq(text)          # and the build over segments the atomic path touched
    assert np.array_equal(m.getKmerCounts(kmers), 3 * counts)
    st = m.stats()
    assert st["distinct"] == o.distinct() and st["kmers_added"] == 3 * n and st["insert_failures"] == 0
    m.close()


@pytest.mark.parametrize("k", list(range(8, 33)))
def test_rolling_hash_every_one_limb_k(T, k):
    """The scan kernel of the partitioned path hashes the first window of a 16-position strip with
    the LUT and every further one with the sliding update table of x -> c*x in GF(2^2k); lookups
    hash with the LUT.  Any disagreement (a wrong irreducible polynomial, a wrong roll table) loses
    k-mers.  Every k whose key fits one limb and whose table can be partitioned, several seeds;
    short ragged reads so that strips straddle line ends."""
    rng = np.random.default_rng(1000 + k)
    recs = []
    for r in range(400):
        n = int(rng.integers(1, 3 * k + 20))
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
        recs.append(b"@r%d\n" % r + seq + b"\n+\n" + b"I" * n + b"\n")
    text = b"".join(recs)
    exp = python_counts(text, k)
    kmers = T.encode_many(list(exp.keys()), k) if exp else np.zeros((0, 1), dtype=np.uint64)
    want = np.array(list(exp.values()), dtype=np.uint64)
    for seed in (1, 2, 77):
        m = T.TSXHashMapHIP(15, 0, k, hash_seed=seed)
        m.set_path("partitioned")
        m.countFastq(text)
        st = m.stats()
        assert st["distinct"] == len(exp) and st["kmers_added"] == int(want.sum()) and st["insert_failures"] == 0
        if len(exp):
            assert np.array_equal(m.getKmerCounts(kmers), want)
        m.close()


def test_log_region_overflow_takes_the_side_path(T, monkeypatch):
    """A wave's key-log region that fills up hands the rest of its keys to the atomic path
    (scan_side_insert): forced here by shrinking the regions to 64 keys."""
    from tsxcount_amd import synth
    text = synth.fastq(9, 0, 600)
    monkeypatch.setenv("TSX_HIP_LOG_CAP", "64")
    assert_same_as_oracle(T, text, 31, 20, 0, path="partitioned")
    assert_same_as_oracle(T, synth.fastq(10, 0, 100), 14, 18, 4, path="partitioned")
    monkeypatch.delenv("TSX_HIP_LOG_CAP")


def test_hot_keys_overflow_sub_lists_two_radix_levels(T):
    """Every read ends in the same 40 bases: the 10 k-mers inside that suffix occur once per read.  At
    l = 23 the table is split by two radix levels; the hot keys fill their sub-lists, go through the
    level-2 spill cache, and the ordinary keys they displace go through the overflow queues.  Every
    count must still be exact."""
    rng = np.random.default_rng(77)
    n_reads, body, k = 30000, 90, 31
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    suffix = lut[rng.integers(0, 4, 40)].tobytes()
    bodies = lut[rng.integers(0, 4, (n_reads, body))]
    qual = b"I" * (body + 40)
    text = b"".join(b"@r%d\n" % i + bodies[i].tobytes() + suffix + b"\n+\n" + qual + b"\n" for i in range(n_reads))
    st = assert_same_as_oracle(T, text, k, 23, 0, path="partitioned")
    assert st["fallback_inserts"] > 10 * n_reads // 2   # the spill path really ran
    m = T.TSXHashMapHIP(23, 0, k)
    m.set_path("partitioned")
    m.countFastq(text)
    hot = T.encode_many([suffix[i:i + k] for i in range(10)], k)
    assert (m.getKmerCounts(hot) == n_reads).all()
    m.close()


def test_fused_scan_sub_list_overflow(T, monkeypatch):
    """walk_part_kernel keeps one fixed-capacity sub-list per (workgroup, level-1 bucket).  Shrunk to 16 keys
    nearly every key finds its sub-list full and takes the spill cache / overflow queue / deferred list; the
    counts must not change.  Also the same text with the fused kernel switched off (key log + separate level 1)."""
    from tsxcount_amd import synth
    text = synth.fastq(21, 0, 6000)
    monkeypatch.setenv("TSX_HIP_CAP1", "16")
    st = assert_same_as_oracle(T, text, 31, 23, 0, path="partitioned")
    assert st["fallback_inserts"] > 1000
    assert_same_as_oracle(T, text, 20, 24, 4, path="partitioned")
    monkeypatch.delenv("TSX_HIP_CAP1")
    monkeypatch.setenv("TSX_HIP_FUSE", "0")
    assert_same_as_oracle(T, text, 31, 23, 0, path="partitioned")
    monkeypatch.delenv("TSX_HIP_FUSE")


@pytest.mark.parametrize("fuse", ["2", "2L", "0"])
def test_fused_and_unfused_scan_agree_entry_for_entry(T, monkeypatch, fuse):
    """Two radix levels, pieces of the host entry point (segments rebuilt from their previous content): the forms of the
    scan (2: strip descriptions + the walk fused with level 1, 0: strip descriptions + key log + level 1) leave the same table."""
    from tsxcount_amd import synth
    if fuse == "2L":   # the two-kernel form with four strips per description
        monkeypatch.setenv("TSX_HIP_LOCAL_LONG", "1")
        fuse = "2"
    monkeypatch.setenv("TSX_HIP_FUSE", fuse)
    monkeypatch.setenv("TSX_HIP_PIECE_BYTES", "300000")
    text = synth.fastq(33, 0, 5000)
    assert_same_as_oracle(T, text, 27, 24, 0, path="partitioned")
    monkeypatch.delenv("TSX_HIP_LOCAL_LONG", raising=False)
    monkeypatch.delenv("TSX_HIP_PIECE_BYTES")
    monkeypatch.delenv("TSX_HIP_FUSE")


def test_stage_timing_hooks(T):
    from tsxcount_amd import synth
    text = synth.fastq(5, 0, 3000)
    m = T.TSXHashMapHIP(22, 0, 31)
    m.set_path("partitioned")
    m.set_timing(True)
    m.countFastq(text)
    stage, pieces = m.get_stage_timing()
    assert pieces >= 1 and set(stage) == {"line", "scan", "level1", "level2", "build", "gap", "post"}
    assert stage["scan"] > 0 and stage["level1"] > 0 and stage["build"] > 0
    assert stage["level2"] < 0.2 * stage["level1"]   # l=22: one radix level, two events back to back
    m.set_path("atomic")
    m.countFastq(text)
    a, b, c, pieces = m.get_timing()
    assert pieces >= 1 and b > 0 and c < 0.2 * b   # atomic path: the partition events are recorded back to back
    m.set_timing(False)
    m.close()


def test_partitioned_path_golden_and_skew(T, golden_fastq, golden_counts):
    from tsxcount_amd import synth
    m = T.TSXHashMapHIP(26, 4, 14)
    m.set_path("partitioned")
    m.countFastq(golden_fastq)
    kmers = T.encode_many(list(golden_counts.keys()), 14)
    exp = np.array(list(golden_counts.values()), dtype=np.uint64)
    assert np.array_equal(m.getKmerCounts(kmers), exp)
    assert m.stats()["distinct"] == 194697
    m.close()
    # Zipf-skewed reads: a few segments receive most keys, lists overflow into the queues and the deferred list
    text = synth.zipf_fastq(7, n_reads=6000, read_len=150, n_templates=300, k=31)
    assert_same_as_oracle(T, text, 31, 18, 0, path="partitioned")
    text63 = synth.zipf_fastq(7, n_reads=6000, read_len=150, n_templates=300, k=63)
    assert_same_as_oracle(T, text63, 63, 18, 0, path="partitioned")        # BASELINE config 4 at oracle scale
    assert_same_as_oracle(T, text63, 63, 22, 3, path="partitioned")        # two radix levels, 3-bit counters
    assert_same_as_oracle(T, text, 31, 18, 2, path="partitioned", overflow_l=17)  # nearly every key carries
    # every read identical: one hot stretch of keys
    one = synth.fastq(3, 0, 1)
    assert_same_as_oracle(T, one * 300, 31, 16, 0, path="partitioned")


@pytest.mark.parametrize("name", ["empty", "short_reads", "no_trailing_newline", "empty_lines", "crlf",
                                  "trailing_partial_record"])
def test_partitioned_path_edge_cases(T, name):
    assert_same_as_oracle(T, EDGE_TEXTS[name], 8, 15, 0, path="partitioned")


# --- edge cases of the record rules ----------------------------------------------------

EDGE_TEXTS = {
    "empty": b"",
    "only_newlines": b"\n\n\n",
    "short_reads": b"@a\nACG\n+\n!!!\n@b\nAC\n+\n!!\n",
    "no_trailing_newline": b"@a\nACGTACGTAC\n+\n!!!!!!!!!!\n@b\nTTTTTTTTTT\n+\n!!!!!!!!!!",
    "empty_lines": b"\n@a\n\n\nACGTACGTACGT\n\n+\n\n!!!!!!!!!!!!\n\n\n@b\nGGGGGGGGGGGG\n+\n!!!!!!!!!!!!\n\n",
    "exactly_k": b"@a\nACGTACGT\n+\n!!!!!!!!\n",
    "with_N_and_lowercase": b"@a\nACGTNNACGTacgtNACGTTTGA\n+\n!!!!!!!!!!!!!!!!!!!!!!!!\n",
    "crlf": b"@a\r\nACGTACGTACGT\r\n+\r\n!!!!!!!!!!!!\r\n",
    "quality_starts_with_at": b"@a\nACGTACGTAAAA\n+\n@@@@@@@@@@@@\n@b\nCCCCCCCCACGT\n+\n@ACGTACGTACG\n",
    "header_looks_like_sequence": b"ACGTACGTACGT\nTTTTTTTTTTTT\nACGTACGTACGT\nACGTACGTACGT\n",
    "trailing_partial_record": b"@a\nACGTACGTACGT\n+\n!!!!!!!!!!!!\n@b\nGGGGGGGGGGGG\n",
}


@pytest.mark.parametrize("name", sorted(EDGE_TEXTS))
def test_edge_cases(T, name):
    assert_same_as_oracle(T, EDGE_TEXTS[name], 8, 12, 0)


def _fuzz_text(rng):
    """Random FASTQ-ish text: record pieces of any length (0 included), stray empty lines, CR before
    LF, bytes outside ACGT, a missing final newline, a truncated last record."""
    alpha = np.frombuffer(b"ACGTACGTACGTACGTNacgt", dtype=np.uint8)
    out = []
    for r in range(int(rng.integers(1, 60))):
        n = int(rng.choice([0, 1, 5, 9, 15, 16, 17, 31, 32, 33, 47, 48, 64, 100, 257])) if rng.random() < 0.5 \
            else int(rng.integers(0, 140))
        if rng.random() < 0.15:
            seq = bytes([int(rng.choice(alpha[:4]))]) * n          # homopolymer read
        else:
            seq = bytes(rng.choice(alpha, size=n))
        eol = b"\r\n" if rng.random() < 0.05 else b"\n"
        lines = [b"@" + b"h" * int(rng.integers(0, 40)), seq, b"+", b"I" * int(rng.integers(0, 140))]
        for ln in lines:
            if rng.random() < 0.1:
                out.append(b"\n" * int(rng.integers(1, 3)))       # empty lines are dropped by the reader
            out.append(ln + eol)
    text = b"".join(out)
    if rng.random() < 0.3:
        text = text[:max(0, len(text) - int(rng.integers(1, 40)))]  # cut inside the last record
    return text


@pytest.mark.parametrize("seed", list(range(16)))
def test_fuzzed_record_structure_wide_keys(T, seed):
    """The same ragged texts for k = 33..127: three-word newline masks, hash-recognised homopolymers,
    2/4-word records in the log, the rings and the LDS segments."""
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(33, 128))
    text = b"".join(_fuzz_text(rng) for _ in range(int(rng.integers(2, 8))))
    assert_same_as_oracle(T, text, k, 14, 0, path="partitioned")
    assert_same_as_oracle(T, text, k, 14, 2, path="partitioned", overflow_l=14)


@pytest.mark.parametrize("k", list(range(33, 128)))
def test_rolling_hash_every_multi_limb_k(T, k):
    """x -> c*x in GF(2^2k) for every multi-limb k (irreducible polynomials of degree 66..254,
    scripts/find_irreducible.py): the scan rolls the hash from window to window, lookups use the LUT; a
    wrong polynomial or roll table loses k-mers.  Ragged short reads, homopolymer reads included."""
    rng = np.random.default_rng(3000 + k)
    recs = []
    for r in range(120):
        n = int(rng.integers(1, 2 * k + 40))
        if r % 10 == 9:
            seq = bytes([int(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8)))]) * n
        else:
            seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
        recs.append(b"@r%d\n" % r + seq + b"\n+\n" + b"I" * n + b"\n")
    text = b"".join(recs)
    exp = python_counts(text, k)
    kmers = T.encode_many(list(exp.keys()), k) if exp else np.zeros((0, T.key_limbs(k)), dtype=np.uint64)
    want = np.array(list(exp.values()), dtype=np.uint64)
    m = T.TSXHashMapHIP(14, 0, k, hash_seed=k)
    m.set_path("partitioned")
    m.countFastq(text)
    st = m.stats()
    assert st["distinct"] == len(exp) and st["kmers_added"] == int(want.sum()) and st["insert_failures"] == 0
    if len(exp):
        assert np.array_equal(m.getKmerCounts(kmers), want)
        x = kmers[0]
        assert np.array_equal(m.hash_invert(m.hash_apply(x)), x)
    m.close()


@pytest.mark.parametrize("seed", list(range(8)))
def test_fuzzed_record_structure_two_radix_levels(T, seed, monkeypatch):
    """The ragged texts through the scan forms of a table split by two radix levels (l = 23): strip descriptions + the
    walk fused with level 1 (default) and the key-log form; k >= 12 keeps 2k >= l + 1 (the slot stores key bits above l)."""
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.integers(12, 33))
    text = b"".join(_fuzz_text(rng) for _ in range(int(rng.integers(2, 8))))
    assert_same_as_oracle(T, text, k, 23, 0, path="partitioned")
    monkeypatch.setenv("TSX_HIP_FUSE", "0")
    assert_same_as_oracle(T, text, k, 23, 0, path="partitioned")
    monkeypatch.delenv("TSX_HIP_FUSE")


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzzed_record_structure_both_paths(T, seed):
    """Strips of 16 start positions, half-strips, tiles of 4 KiB and line ends fall everywhere in
    these texts; both insert paths must agree with the oracle for a k drawn per seed."""
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.integers(8, 33))
    text = b"".join(_fuzz_text(rng) for _ in range(int(rng.integers(1, 6))))
    assert_same_as_oracle(T, text, k, 15, 0, path="partitioned")
    assert_same_as_oracle(T, text, k, 15, 2, path="atomic", overflow_l=15)


@pytest.mark.parametrize("k,l,path", [(21, 16, "atomic"), (21, 16, "partitioned"), (31, 20, "partitioned"),
                                      (31, 23, "partitioned"), (20, 24, "partitioned"),   # two radix levels: strip_desc + walk_part
                                      (63, 18, "partitioned"), (127, 18, "atomic"), (127, 18, "partitioned")])
def test_fasta_records(T, k, l, path):
    """tsx_hip_set_record_lines(2): FASTA as FASTXreader<FASTAEntry> reads it (FastXReader.h:97-116) -- header
    line, one sequence line, empty lines dropped; nearly every byte starts a k-mer (no quality lines), which
    also sizes the key log differently.  Against the oracle with the same record rule."""
    from oracle.oracle import Oracle
    rng = np.random.default_rng(k + l)
    recs = []
    for r in range(400):
        n = int(rng.integers(1, 4 * k + 60))
        recs.append(b">read%d len=%d\n" % (r, n) + bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n,
                                                                 p=[0.24, 0.25, 0.25, 0.25, 0.01])) + b"\n")
        if r % 11 == 0:
            recs.append(b"\n")
    text = b"".join(recs)
    o = Oracle(k, 20, 4, seed=1)
    n = o.count_fastq(text, 2)
    m = T.TSXHashMapHIP(l, 0, k)
    m.set_record_lines(2)
    m.set_path(path)
    m.countFastq(text)
    st = m.stats()
    assert st["kmers_added"] == n and st["distinct"] == o.distinct() and st["insert_failures"] == 0
    kmers, counts = o.dump()
    assert np.array_equal(m.getKmerCounts(kmers), counts)
    # the same text read as FASTQ gives something else entirely (every 4th line only)
    m2 = T.TSXHashMapHIP(l, 0, k)
    m2.set_path(path)
    m2.countFastq(text)
    assert m2.stats()["kmers_added"] == Oracle(k, 20, 4, seed=1).count_fastq(text, 4) != n
    m.close(); m2.close()
    with pytest.raises(T.TSXException):
        T.TSXHashMapHIP(l, 0, k).set_record_lines(3)


def test_long_reads_span_many_tiles(T):
    # one 50 kb read (the bundled fixture has 20 kb reads) + long header and quality lines
    rng = np.random.default_rng(3)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 50000)].tobytes()
    text = b"@" + b"h" * 9000 + b"\n" + seq + b"\n+\n" + b"I" * 50000 + b"\n"
    assert_same_as_oracle(T, text * 2, 31, 18, 0)
    assert_same_as_oracle(T, text, 127, 18, 0)


def test_piece_seams_of_the_host_path(T):
    """The host entry point streams the text in pieces; seams fall inside lines,
    inside k-mer windows and on newlines.  Small pieces force thousands of seams."""
    from tsxcount_amd import synth
    text = synth.fastq(8, 0, 60)
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; import tsxcount_amd as T;"
            "from tsxcount_amd import synth; text = synth.fastq(8, 0, 60);"
            "m = T.TSXHashMapHIP(18, 0, 31); m.countFastq(text); k, c = m.getAllKmers();"
            "o = np.lexsort(k.T[::-1]); np.save(sys.argv[1], np.concatenate([k[o].ravel(), c[o]]))" % ROOT)
    import tempfile
    outs = []
    for piece in ("4096", "5008", "65536"):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "r.npy")
            env = dict(os.environ, TSX_HIP_PIECE_BYTES=piece)
            subprocess.run([sys.executable, "-c", code, f], check=True, env=env, timeout=300)
            outs.append(np.load(f))
    o, n = oracle_for(text, 31, 18, 4)
    kmers, counts = o.dump()
    b = np.lexsort(kmers.T[::-1])
    expect = np.concatenate([kmers[b].ravel(), counts[b]])
    for got in outs:
        assert np.array_equal(got, expect)


def test_device_entry_point_equals_host_entry_point(T):
    import torch
    from tsxcount_amd import synth
    text = synth.fastq(12, 0, 300)
    dev = torch.device("cuda", 0)
    buf = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    m = T.TSXHashMapHIP(20, 0, 31)
    torch.cuda.synchronize()
    m.countFastqDevice(buf.data_ptr(), len(text))
    m.sync()
    o, n = oracle_for(text, 31, 20, 4)
    assert m.stats()["kmers_added"] == n and m.stats()["distinct"] == o.distinct()
    kmers, counts = o.dump()
    assert np.array_equal(m.getKmerCounts(kmers), counts)
    # misaligned device pointers are refused, not silently mis-read
    rc = m._lib.tsx_hip_count_fastq_device(m.handle, ctypes.c_void_p(buf.data_ptr() + 1), 100, None)
    assert rc == T.EINVAL
    m.close()


def test_device_entry_point_windows(T):
    """Device texts are processed in windows (4 GiB in production); small windows put the
    seams inside lines, inside k-mer windows and on newlines, for both insert paths."""
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, torch; import tsxcount_amd as T;"
            "from tsxcount_amd import synth; text = synth.fastq(14, 0, 500);"
            "buf = torch.frombuffer(bytearray(text), dtype=torch.uint8).to('cuda:0');"
            "m = T.TSXHashMapHIP(int(sys.argv[3]), 0, 31); m.set_path(sys.argv[2]); torch.cuda.synchronize();"
            "m.countFastqDevice(buf.data_ptr(), len(text)); m.sync(); k, c = m.getAllKmers();"
            "o = np.lexsort(k.T[::-1]); np.save(sys.argv[1], np.concatenate([k[o].ravel(), c[o]]))" % ROOT)
    from tsxcount_amd import synth
    import tempfile
    text = synth.fastq(14, 0, 500)
    o, n = oracle_for(text, 31, 20, 4)
    kmers, counts = o.dump()
    b = np.lexsort(kmers.T[::-1])
    expect = np.concatenate([kmers[b].ravel(), counts[b]])
    # l = 23: a table split by two radix levels -- the windows then go through strip_desc_kernel + walk_part_kernel
    for window, path, l in (("4096", "atomic", 20), ("100000", "partitioned", 20), ("65536", "atomic", 20),
                            ("40000", "partitioned", 23), ("8192", "partitioned", 23)):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "r.npy")
            env = dict(os.environ, TSX_HIP_DEV_WINDOW=window)
            subprocess.run([sys.executable, "-c", code, f, path, str(l)], check=True, env=env, timeout=300)
            assert np.array_equal(np.load(f), expect), (window, path)


def test_synth_kernel_matches_numpy_generator(T):
    import torch
    from tsxcount_amd import synth
    for seed, first, n in ((5, 0, 37), (6, 123456, 20)):
        text = synth.fastq(seed, first, n)
        nb, nk, _ = T.synth_sizes(seed, first, n, 31)
        buf = torch.zeros(nb + 64, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        T.synth_fastq_device(seed, first, n, 31, buf.data_ptr(), nb)
        assert bytes(buf[:nb].cpu().numpy().tobytes()) == text


# --- insert / query surface -----------------------------------------------------------

def test_add_and_get_surface(T):
    m = T.TSXHashMapHIP(16, 0, 14)
    assert m.getK() == 14 and m.getMaxElements() == 1 << 16
    assert m.getKmerCount("ACGTACGTACGTAC") == 0
    assert m.addKmer("ACGTACGTACGTAC") is True
    m.addKmer(T.encode("ACGTACGTACGTAC"))
    assert m.getKmerCount("ACGTACGTACGTAC") == 2
    assert m.getKmerCount() == 1
    # testExecution.h:406-496 (testHashMapOld): four k-mers, 196608 / 98304 / 98304 / 49152 adds
    vals = [10067, 2786, 9816, 156]
    n = 2048 * 4 * 24
    seq = []
    for i in range(n):
        seq.append(vals[0])
        seq.append(vals[1] if i % 2 == 0 else vals[2])
        if i % 4 == 0:
            seq.append(vals[3])
    m.addKmers(np.array(seq, dtype=np.uint64))
    got = m.getKmerCounts(np.array(vals, dtype=np.uint64))
    assert list(map(int, got)) == [n, n // 2, n // 2, n // 4]
    # weighted adds (merge path) and zero weights
    m.addKmers(np.array([vals[0], 777], dtype=np.uint64), counts=np.array([5, 0], dtype=np.uint64))
    assert m.getKmerCount(np.array([vals[0]], dtype=np.uint64)) == n + 5
    assert m.getKmerCount(np.array([777], dtype=np.uint64)) == 0
    m.clear()
    assert m.getKmerCount() == 0 and m.stats()["kmers_added"] == 0
    m.close()


@pytest.mark.parametrize("k,s", [(14, 1), (14, 3), (31, 2), (63, 2), (127, 4)])
def test_count_overflow_into_secondary_array(T, k, s):
    """Hot k-mers far beyond 2^s: the in-slot counter wraps, carries go to the
    secondary array (the reference chains overflow slots, TSXHashMapPerf.h:699-881)."""
    rng = np.random.default_rng(k)
    wk = T.key_limbs(k)
    hot = rng.integers(0, 2 ** 62, size=(7, wk), dtype=np.uint64)
    hot[:, -1] &= np.uint64((1 << ((2 * k) % 64 or 64)) - 1)
    reps = [1, 2 ** s - 1, 2 ** s, 2 ** s + 1, 1000, 65537, 300000]
    batch = np.concatenate([np.repeat(hot[i:i + 1], r, axis=0) for i, r in enumerate(reps)])
    rng.shuffle(batch, axis=0)
    m = T.TSXHashMapHIP(14, s, k)
    assert m.layout.count_bits == s
    m.addKmers(batch)
    m.addKmers(hot, counts=np.array([3] * 7, dtype=np.uint64))
    assert list(map(int, m.getKmerCounts(hot))) == [r + 3 for r in reps]
    st = m.stats()
    assert st["overflow_used"] >= 4 and st["overflow_failures"] == 0
    gk, gc = m.getAllKmers()
    assert sorted(map(int, gc)) == sorted(r + 3 for r in reps)
    m.close()


def test_secondary_array_full_is_reported(T):
    m = T.TSXHashMapHIP(14, 1, 14, overflow_l=4)
    kmers = np.arange(1, 2001, dtype=np.uint64)
    with pytest.raises(T.TSXException) as e:
        m.addKmers(np.repeat(kmers, 3))
    assert e.value.code == T.EOVERFLOW
    assert m.stats()["overflow_failures"] > 0
    m.close()


def test_table_full_is_reported_like_exit_42(T):
    # more distinct k-mers than slots: reference prints "Could not insert kmer" and exit(42)
    m = T.TSXHashMapHIP(8, 0, 14)
    with pytest.raises(T.TSXException) as e:
        m.addKmers(np.arange(1, 600, dtype=np.uint64))
    assert e.value.code == T.EFULL
    st = m.stats()
    assert st["insert_failures"] > 0 and st["distinct"] <= 256
    m.close()


def test_high_load_factor(T):
    """Load factor ~0.8 (BASELINE config 5 shape, scaled down): every key must
    still be found with the 8-bit reprobe field."""
    k = 127
    n = int(0.8 * (1 << 16))
    rng = np.random.default_rng(1)
    kmers = rng.integers(0, 2 ** 62, size=(n, 4), dtype=np.uint64)
    m = T.TSXHashMapHIP(16, 2, k)
    m.addKmers(kmers)
    m.addKmers(kmers[: n // 2])
    got = m.getKmerCounts(kmers)
    assert np.array_equal(got[: n // 2], np.full(n // 2, 2, dtype=np.uint64))
    assert np.array_equal(got[n // 2:], np.full(n - n // 2, 1, dtype=np.uint64))
    assert m.getKmerCount() == n
    m.close()


def test_zipf_skewed_reads_k63(T):
    """BASELINE config 4 at oracle scale: Zipf-skewed reads, 2-limb keys, heavy
    contention on a few k-mers and long reprobe chains."""
    from tsxcount_amd import synth
    text = synth.zipf_fastq(7, n_reads=3000, read_len=150, n_templates=400, k=63)
    st = assert_same_as_oracle(T, text, 63, 17, 0)
    assert st["kmers_added"] == 3000 * (150 - 63 + 1)
    assert_same_as_oracle(T, text, 63, 17, 3)


def test_dump_partition_by_owner(T):
    import torch
    from tsxcount_amd import synth
    text = synth.fastq(31, 0, 100)
    m = T.TSXHashMapHIP(18, 0, 31)
    m.countFastq(text)
    n = m.stats()["distinct"]
    world = 4
    dev = torch.device("cuda", 0)
    kmers = torch.zeros((n, 1), dtype=torch.int64, device=dev)
    counts = torch.zeros((n,), dtype=torch.int64, device=dev)
    seg = torch.zeros((world,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    T._check(m._lib.tsx_hip_partition_device(m.handle, world, ctypes.c_void_p(kmers.data_ptr()),
                                             ctypes.c_void_p(counts.data_ptr()), n,
                                             ctypes.c_void_p(seg.data_ptr()), None))
    seg = seg.cpu().numpy()
    assert int(seg.sum()) == n and (seg > 0).all()
    k_np = kmers.cpu().numpy().view(np.uint64)
    c_np = counts.cpu().numpy().view(np.uint64)
    bounds = np.concatenate([[0], np.cumsum(seg)])
    for r in range(world):
        for i in range(bounds[r], bounds[r + 1], max(1, seg[r] // 50)):
            assert m.owner(k_np[i], world) == r
    o, _ = oracle_for(text, 31, 18, 4)
    ok, oc = o.dump()
    a, b = np.lexsort(k_np.T[::-1]), np.lexsort(ok.T[::-1])
    assert np.array_equal(k_np[a], ok[b]) and np.array_equal(c_np[a], oc[b])
    m.close()


# --- BASELINE-size properties (no oracle at this size) ------------------------------------

@pytest.mark.parametrize("path", ["atomic", "partitioned"])
def test_full_size_properties(T, path):
    """The bench shape itself (BASELINE config 2): 1,087,000 reads = 1e9 k-mers, k=31, 2^30 slots, load 0.75,
    where the level-2 spill cache, the overflow queues and the 255-probe limit actually bite.  No CPU
    count exists at this size, so: totals read back from the table, the analytically known polyA count,
    EXACT counts for every k-mer of 1000 sampled reads (tsxcount_amd/verify.py), idempotence (counting
    the same text twice doubles every count, distinct unchanged)."""
    import torch
    from tsxcount_amd import verify
    n_reads, k, seed = 1087000, 31, 20261004
    nb, nk, npolya = T.synth_sizes(seed, 0, n_reads, k, want_polya=True)
    buf = torch.empty(nb + 64, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    T.synth_fastq_device(seed, 0, n_reads, k, buf.data_ptr(), nb)
    m = T.TSXHashMapHIP(30, 0, k)
    m.set_path(path)
    m.countFastqDevice(buf.data_ptr(), nb)
    m.sync()
    st = m.stats()
    assert st["kmers_added"] == nk and st["count_sum"] == nk
    assert st["insert_failures"] == 0 and st["overflow_failures"] == 0 and st["lock_timeouts"] == 0
    polya = T.encode("A" * k)
    assert m.getKmerCount(polya) == npolya
    d1 = st["distinct"]
    # everything but polyA is unique except the ~10 windows per read that hold <= 9
    # random bases in front of the A-tail (4^9 < n_reads, so they repeat across reads)
    assert d1 < nk - npolya and d1 > nk - npolya - 12 * n_reads
    assert 0.74 < d1 / (1 << 30) < 0.76
    # sampled reads: every k-mer at least its in-sample multiplicity, exactly it when >= 24 random bases
    kmers, mult, safe = verify.sample_expectations(seed, k, verify.sample_read_ids(0, n_reads, 1000))
    res = verify.judge_sample(m.getKmerCounts(kmers), mult, safe)
    assert res["looked_up"] > 700000 and res["safe_kmers"] > 600000
    assert res["below_sample_multiplicity"] == 0 and res["safe_unequal"] == 0, res
    # idempotence
    m.countFastqDevice(buf.data_ptr(), nb)
    m.sync()
    st2 = m.stats()
    assert st2["distinct"] == d1 and st2["kmers_added"] == 2 * nk and st2["count_sum"] == 2 * nk
    assert m.getKmerCount(polya) == 2 * npolya
    res = verify.judge_sample(m.getKmerCounts(kmers), 2 * mult, safe)
    assert res["below_sample_multiplicity"] == 0 and res["safe_unequal"] == 0, res
    m.close()


def test_both_insert_paths_build_the_same_table_at_the_bench_shape(T):
    """bench.py's cross check: the partitioned path's table vs a second table filled through the atomic
    path from the same 1e9 k-mers -- 16 slot ranges of 2^20 slots of the first table's dump, entry by entry."""
    import torch
    from tsxcount_amd import verify
    n_reads, k, seed = 1087000, 31, 20261004
    nb, nk, _ = T.synth_sizes(seed, 0, n_reads, k)
    buf = torch.empty(nb + 64, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    T.synth_fastq_device(seed, 0, n_reads, k, buf.data_ptr(), nb)
    m = T.TSXHashMapHIP(30, 0, k)
    m.set_path("partitioned")
    m.countFastqDevice(buf.data_ptr(), nb)
    m.sync()
    res = verify.cross_check(m, buf.data_ptr(), nb, "atomic")
    assert res["ok"] and res["entries_compared"] > 10_000_000 and res["entries_unequal"] == 0, res
    m.close()


def test_load_factor_0p9_with_255_segment_confined_probes(T):
    """The probe sequence stops after 255 probes and wraps inside the 16 Ki-slot segment of the home slot
    (the reference probes up to 2^l - 1 times over the whole table, TSXHashMap.h:86): INTEGRATION.md
    section 5 states 0.9 as the highest load this is tested at.  Both insert paths, 2^20 slots."""
    rng = np.random.default_rng(9)
    n = int(0.9 * (1 << 20))
    kmers = np.unique(rng.integers(0, 2 ** 62, size=n + n // 50, dtype=np.uint64))[:n].reshape(-1, 1)
    rng.shuffle(kmers)
    m = T.TSXHashMapHIP(20, 0, 31)
    m.addKmers(kmers)
    st = m.stats()
    assert st["insert_failures"] == 0 and st["distinct"] == n and st["count_sum"] == n
    assert (m.getKmerCounts(kmers) == 1).all()
    m.close()
    # the same load through the partitioned path: a FASTQ whose reads are the k-mers themselves
    seqs = [T.decode(kmers[i], 31).encode() for i in range(0, n, 4)]
    text = b"".join(b"@r\n" + s_ + b"\n+\n" + b"I" * 31 + b"\n" for s_ in seqs)
    m = T.TSXHashMapHIP(18, 0, 31)
    m.set_path("partitioned")
    m.countFastq(text)
    st = m.stats()
    assert st["insert_failures"] == 0 and st["distinct"] == len(seqs) and 0.89 < len(seqs) / (1 << 18) < 0.91
    m.close()


def test_count_fastq_refuses_a_shard(T):
    """A map created with shard_bits > 0 holds one slot range of a larger table: the single-table entry
    points would insert other owners' keys with their owner bits dropped."""
    from tsxcount_amd import synth
    m = T.TSXHashMapHIP(16, 0, 31, shard_bits=1, shard_index=0)
    with pytest.raises(T.TSXException) as e:
        m.countFastq(synth.fastq(1, 0, 3))
    assert e.value.code == T.EINVAL
    m.close()


def test_two_rank_merge_on_one_gpu():
    """Two processes share cuda:0, count disjoint read shards into their own
    tables and merge (gloo collective staged through host memory).  RCCL itself
    needs one GPU per rank, which only the driver's multi-GPU node has."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(ROOT, "tests", "_merge_worker.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "MERGE OK" in o


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_counting_on_one_gpu(world):
    """The multi-GPU counting pipeline (keys travel to the shard that owns their slot range,
    then are built there) with 2 and 4 ranks on cuda:0 and a gloo all-to-all staged through host memory."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(ROOT, "tests", "_shard_worker.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "SHARD OK" in o


@pytest.mark.parametrize("world", [2, 3, 4])
def test_minimizer_counting_on_one_gpu(world):
    """The minimizer exchange (owner of a k-mer = f(its minimizer), every rank a whole table of what it owns) with 2, 3 and
    4 ranks on cuda:0 and gloo collectives: every k-mer of the oracle's dump on exactly the rank tsx_hip_mini_owner_host
    names, with the oracle's count; k = 20 .. 32, FASTA, empty and uneven shards, repeated and cleared steps."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(ROOT, "tests", "_mini_worker.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "MINI OK" in o


def test_minimizer_counting_world_8_in_one_process(T):
    """Eight ranks of the minimizer exchange as eight THREADS on cuda:0 (collectives = device copies behind a barrier,
    tests/_thread_comm.py): the world size of the 8-GPU node.  Sum over ranks == oracle, every k-mer on the rank its
    minimizer names and only there."""
    import threading
    import torch
    from _thread_comm import ThreadComm, ThreadWorld
    from oracle.oracle import Oracle
    from tsxcount_amd import distributed as TD
    from tsxcount_amd import synth
    world, k, l, n_reads = 8, 31, 23, 640
    tw = ThreadWorld(world)
    whole = Oracle(k, 21, 4, seed=1)
    whole.count_fastq(synth.fastq(68, 0, n_reads))
    kmers, counts = whole.dump()
    owner = TD.owner_of(kmers, k, world)
    got = [None] * world
    errs = []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            first, cnt = TD.shard_reads(n_reads, rank, world)
            text = synth.fastq(68, first, cnt)
            buf = torch.frombuffer(bytearray(text + b"\n" * 64), dtype=torch.uint8).to("cuda:0")
            m = T.TSXHashMapHIP(l, 0, k, device=0)
            mc = TD.MinimizerCounter(m, len(text), group=ThreadComm(tw, rank), windows=3)
            torch.cuda.synchronize()
            mc.step(buf.data_ptr(), len(text))
            assert mc.last["key_sum_diff"] == 0 and mc.last["mode"] == "minimizer"
            st = m.stats()
            assert st["insert_failures"] == 0
            got[rank] = (m.getKmerCounts(kmers), st)
            m.close()
        except BaseException as e:   # noqa: BLE001 -- a dead rank must not leave the others in a barrier
            errs.append((rank, repr(e)))
            tw.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(got[r][0], np.where(owner == r, counts, 0).astype(np.uint64)), "rank %d" % r
    assert sum(g[1]["distinct"] for g in got) == len(kmers)
    assert sum(g[1]["count_sum"] for g in got) == int(counts.sum())


def test_minimizer_entry_points_refuse_what_they_cannot_do(T):
    """k outside 20..32, a shard of a slot-range table, more than 16 owners, a list capacity below what
    tsx_hip_mini_part_capacity asks for, the walk flag on a sharded map: error codes, no launch.  A text described in
    pieces and split in shares gives the same lists' content as one call (counts of valid starts)."""
    import ctypes
    import torch
    from tsxcount_amd import synth
    vp = ctypes.c_void_p
    text = synth.fastq(71, 0, 300)
    buf = torch.frombuffer(bytearray(text + b"\n" * 64), dtype=torch.uint8).to("cuda:0")
    i64 = dict(dtype=torch.int64, device="cuda:0")
    cnt = torch.zeros((20,), **i64)
    emit = torch.zeros((2,), **i64)
    small = T.TSXHashMapHIP(23, 0, 19)
    assert not small._lib.tsx_hip_mini_supported(small.handle)
    shard = T.TSXHashMapHIP(23, 0, 31, shard_bits=1, shard_index=0)
    assert not shard._lib.tsx_hip_mini_supported(shard.handle)
    dsc = torch.empty((2 * 4096,), **i64)
    for bad in (small, shard):
        assert bad._lib.tsx_hip_mini_describe_device(bad.handle, vp(buf.data_ptr()), len(text), 0, len(text), vp(emit.data_ptr()), None) == T.EINVAL
        assert bad._lib.tsx_hip_mini_window_device(bad.handle, vp(buf.data_ptr()), len(text), 0, len(text), 2, vp(dsc.data_ptr()), 4096,
                                                   vp(cnt.data_ptr()), vp(emit.data_ptr()), None) == T.EINVAL
    assert shard._lib.tsx_hip_shard_walk_device(shard.handle, vp(dsc.data_ptr()), 16, 2, 0, 1, 1000, vp(emit[1:].data_ptr()), None) == T.EINVAL
    m = T.TSXHashMapHIP(23, 0, 31)
    L = m._lib
    cap = ctypes.c_size_t(0)
    assert L.tsx_hip_mini_capacity(m.handle, len(text), 17, ctypes.byref(cap)) == T.EINVAL
    assert L.tsx_hip_mini_window_device(m.handle, vp(buf.data_ptr()), len(text), 0, len(text), 17, vp(dsc.data_ptr()), 4096,
                                        vp(cnt.data_ptr()), vp(emit.data_ptr()), None) == T.EINVAL
    assert L.tsx_hip_mini_window_device(m.handle, vp(buf.data_ptr()), len(text), 0, len(text), 4, vp(dsc.data_ptr()), 16,
                                        vp(cnt.data_ptr()), vp(emit.data_ptr()), None) == T.ERANGE
    assert L.tsx_hip_mini_describe_device(m.handle, vp(buf.data_ptr()), len(text), 8, 16, vp(emit.data_ptr()), None) == T.EINVAL   # offset not a multiple of 16
    assert L.tsx_hip_mini_split_device(m.handle, 3, 3, 4, vp(dsc.data_ptr()), 4096, vp(cnt.data_ptr()), None) == T.EINVAL
    # one call against describe + three shares: the same valid starts per owner, the same homopolymer totals
    world = 4
    assert L.tsx_hip_mini_part_capacity(m.handle, len(text) + 256, 1, ctypes.byref(cap)) == 0
    c1 = cap.value
    d1 = torch.empty((2 * c1 * world,), **i64)
    assert L.tsx_hip_mini_window_device(m.handle, vp(buf.data_ptr()), len(text), 0, len(text), world, vp(d1.data_ptr()), c1,
                                        vp(cnt.data_ptr()), vp(emit.data_ptr()), None) == 0
    m.sync()
    one = [int(x) for x in cnt[:world + 4].tolist()]

    def starts(d, c, counts):   # valid start positions per owner
        out = []
        for o in range(world):
            lst = d[2 * o * c:2 * (o * c + counts[o])].view(-1, 2)
            w = (lst[:, 1] >> 32) & 0xFFFF
            out.append(int(sum(bin(int(x)).count("1") for x in w.cpu().tolist())))
        return out
    s_one = starts(d1, c1, one)
    assert L.tsx_hip_mini_part_capacity(m.handle, len(text) + 256, 3, ctypes.byref(cap)) == 0
    c3 = cap.value
    assert L.tsx_hip_mini_split_device(m.handle, 0, 3, world, vp(d1.data_ptr()), 16, vp(cnt.data_ptr()), None) == T.ERANGE
    assert L.tsx_hip_mini_describe_device(m.handle, vp(buf.data_ptr()), len(text), 0, len(text), vp(emit.data_ptr()), None) == 0
    s_three, hom = [0] * world, [0] * 4
    d3 = torch.empty((2 * c3 * world,), **i64)
    for part in range(3):
        assert L.tsx_hip_mini_split_device(m.handle, part, 3, world, vp(d3.data_ptr()), c3, vp(cnt.data_ptr()), None) == 0
        m.sync()
        c = [int(x) for x in cnt[:world + 4].tolist()]
        s_three = [a + b for a, b in zip(s_three, starts(d3, c3, c))]
        hom = [a + b for a, b in zip(hom, c[world:])]
    # (runs merged across strips may be cut differently at a share's border: the starts are the same, the descriptions need not be)
    assert s_three == s_one and hom == one[world:] and sum(hom) > 0
    assert int(emit[0].item()) == 2 * (sum(s_one) + sum(hom))


@pytest.mark.parametrize("seed", list(range(10)))
def test_minimizer_exchange_fuzzed_record_structure(T, seed):
    """The ragged texts (records of any length, empty lines, CR LF, N and lower case, homopolymer reads, a truncated last
    record) through the minimizer exchange: the C++ group with 2..6 ranks on cuda:0 cuts the text into record shards (some
    empty), every rank describes, splits, trades lists, walks; sum over ranks and the owner of every k-mer against the
    oracle, for a k drawn per seed from 20..32."""
    from oracle.oracle import Oracle
    from tsxcount_amd import distributed as TD
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(20, 33))
    ranks = int(rng.integers(2, 7))
    text = b"".join(_fuzz_text(rng) for _ in range(int(rng.integers(3, 10))))
    o = Oracle(k, 20, 4, seed=1)
    n = o.count_fastq(text)
    kmers, counts = o.dump()
    g = T.TSXHashMapHIPGroup(ranks, 23, 0, k, devices=[0] * ranks, comm="copy", exchange="mini")
    g.countFastq(text)
    st = g.stats()
    assert st["count_sum"] == n and st["distinct"] == len(kmers) and st["insert_failures"] == 0
    if len(kmers):
        assert np.array_equal(g.getKmerCounts(kmers), counts)
        owner = TD.owner_of(kmers, k, ranks)
        for r in range(ranks):
            assert g.rank_stats(r)["distinct"] == int((owner == r).sum())
    g.close()


@pytest.mark.parametrize("k,world", [(20, 2), (23, 7), (26, 16), (31, 8), (32, 5)])
def test_minimizer_split_lists_hold_what_the_host_function_says(T, k, world):
    """desc_owner_split_kernel against tsx_hip_mini_owner_host: list o of a text's split, walked alone into an empty table,
    must yield exactly the k-mers (with their counts) whose owner the host function says is o -- homopolymers excepted,
    which are counted on the side."""
    import ctypes
    import torch
    from oracle.oracle import Oracle
    from tsxcount_amd import distributed as TD
    from tsxcount_amd import synth
    text = synth.fastq(69 + k, 0, 500)
    ora = Oracle(k, 21, 4, seed=1)
    ora.count_fastq(text)
    kmers, counts = ora.dump()
    owner = TD.owner_of(kmers, k, world)
    hom = np.array([T.decode(km, k) in ("A" * k, "C" * k, "G" * k, "T" * k) for km in kmers])
    buf = torch.frombuffer(bytearray(text + b"\n" * 64), dtype=torch.uint8).to("cuda:0")
    m = T.TSXHashMapHIP(23, 0, k)
    L, vp = m._lib, ctypes.c_void_p
    assert L.tsx_hip_mini_supported(m.handle)
    cap = ctypes.c_size_t(0)
    assert L.tsx_hip_mini_capacity(m.handle, len(text) + 256, world, ctypes.byref(cap)) == 0
    cap = cap.value
    i64 = dict(dtype=torch.int64, device="cuda:0")
    dsc = torch.empty((2 * cap * world,), **i64)
    cnt = torch.zeros((world + 4,), **i64)
    emit = torch.zeros((2,), **i64)
    assert L.tsx_hip_mini_window_device(m.handle, vp(buf.data_ptr()), len(text), 0, len(text), world, vp(dsc.data_ptr()), cap,
                                        vp(cnt.data_ptr()), vp(emit.data_ptr()), None) == 0
    m.sync()
    c = [int(x) for x in cnt.tolist()]
    assert sum(c[world:]) == int(counts[hom].sum()) and all(x % 64 == 0 and x <= cap for x in c[:world])
    walked = 0
    for o in range(world):
        m.clear()
        emit.zero_()
        assert L.tsx_hip_shard_walk_device(m.handle, vp(dsc.data_ptr() + o * cap * 16), c[o], 2, 0, 1, len(text), vp(emit[1:].data_ptr()), None) == 0
        assert L.tsx_hip_shard_build_l1_device(m.handle, None) == 0
        m.sync()
        want = np.where((owner == o) & ~hom, counts, 0).astype(np.uint64)
        assert np.array_equal(m.getKmerCounts(kmers), want), "list %d" % o
        assert m.stats()["distinct"] == int(((owner == o) & ~hom).sum())
        walked += int(emit[1].item())
    assert walked + sum(c[world:]) == int(counts.sum())


@pytest.mark.parametrize("world", [2, 4])
def test_config5_k127_load_0p8_overflow_merge(world):
    """BASELINE config 5 at oracle scale: k = 127 (4-limb keys), --s=2 counters so that counts carry into the
    secondary array, every rank's table at load ~0.8 before AND after the merge, 2 and 4 ranks merged with
    merge_tables (gloo staged through the host; ranks share cuda:0).  k >= 40 is not pinned by the reference
    itself (it aborts there): the expectation is the C restatement."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(ROOT, "tests", "_merge_worker.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(port), "config5"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "MERGE OK" in o


@pytest.mark.parametrize("l,mode", [(16, "auto"), (23, "keys"), (23, "desc"), (23, "desc16"), (23, "desc_walk")])
def test_sharded_counting_world_8_in_one_process(T, monkeypatch, l, mode):
    """(l = 23: a table split by two radix levels -- "keys": level 1 window by window as the keys arrive, "desc": strip
    descriptions all-gathered, every shard walks all of them and keeps what it owns.)
    shard_bits = 3, the value the 8-GPU node uses: eight shards of one table, eight ShardedCounter
    pipelines (3 windows each), run as eight THREADS of this process on cuda:0 with the collectives replaced
    by device copies behind a barrier (tests/_thread_comm.py) -- a one-GPU box admits at most 6 processes on
    the card.  Same checks as the multi-process test: sum over shards == oracle, every k-mer on one shard."""
    import threading
    import torch
    from _thread_comm import ThreadComm, ThreadWorld
    from oracle.oracle import Oracle
    from tsxcount_amd import distributed as TD
    from tsxcount_amd import synth
    if mode == "desc16":   # one strip per description (16 bytes) instead of four (32 bytes)
        monkeypatch.setenv("TSX_HIP_SHARD_LONG", "0")
        mode = "desc"
    if mode == "desc_walk":   # the fused walk (a ring flush per batch at this world size) instead of the filtered key log
        monkeypatch.setenv("TSX_HIP_SHARD_FILTER", "0")
        mode = "desc"
    monkeypatch.setenv("TSX_HIP_SHARD_MODE", mode)
    world, k, n_reads = 8, 31, 400
    tw = ThreadWorld(world)
    whole = Oracle(k, 21, 4, seed=1)
    whole.count_fastq(synth.fastq(66, 0, n_reads))
    kmers, counts = whole.dump()
    got = [None] * world
    errs = []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            first, cnt = TD.shard_reads(n_reads, rank, world)
            text = synth.fastq(66, first, cnt)
            buf = torch.frombuffer(bytearray(text + b"\n" * 64), dtype=torch.uint8).to("cuda:0")
            m = T.TSXHashMapHIP(l, 0, k, device=0, shard_bits=3, shard_index=rank)
            sc = TD.ShardedCounter(m, len(text), group=ThreadComm(tw, rank), windows=3)
            torch.cuda.synchronize()
            sc.step(buf.data_ptr(), len(text))
            assert sc.last["key_sum_diff"] == 0
            assert sc.last.get("mode", "keys") == ("desc" if mode == "desc" else "keys")
            st = m.stats()
            assert st["insert_failures"] == 0
            got[rank] = (m.getKmerCounts(kmers), st)
            m.close()
        except BaseException as e:   # noqa: BLE001 -- a dead rank must not leave the others in a barrier
            errs.append((rank, repr(e)))
            tw.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errs, errs
    total = sum(g[0].astype(np.int64) for g in got)
    assert np.array_equal(total.astype(np.uint64), counts), "sum over the 8 shards != oracle"
    owners = sum((g[0] > 0).astype(np.int64) for g in got)
    assert (owners == 1).all(), "every k-mer must live on exactly one shard"
    assert sum(g[1]["distinct"] for g in got) == len(kmers)
    assert sum(g[1]["count_sum"] for g in got) == int(counts.sum())
    fr = np.array([(g[0] > 0).mean() for g in got])
    assert (fr > 0.4 / world).all() and (fr < 1.6 / world).all()



@pytest.mark.parametrize("l,segbits,k,reads", [(25, 9, 31, 3000), (26, 10, 21, 2500), (25, 10, 32, 2000)])
def test_tables_built_slab_by_slab(T, l, segbits, k, reads):
    """Tables above 2^32 slots (l - S > 18) are built slab by slab: every window of the text is described once, each
    slab walks all descriptions with the owner filter and runs its own level 2 + build (count_slabs, tsxcount_hip.hip).
    TSX_HIP_SLAB_SEGBITS lowers the limit so that a 2^25-slot table takes that route with 4 (l - S - segbits = 2) or 2
    slabs; several text windows; against the oracle, against the atomic path slot range by slot range, and a second
    count into the same (now dirty) table.  Runs in a process of its own: the limit is read once per process."""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import tsxcount_amd as T
from oracle.oracle import Oracle
from tsxcount_amd import synth
l, k, reads = %d, %d, %d
text = synth.fastq(77, 0, reads)
o = Oracle(k, 24, 4, seed=1)
n = o.count_fastq(text)
kmers, counts = o.dump()
m = T.TSXHashMapHIP(l, 0, k)
m.set_path("partitioned")
for rep in (1, 2):
    m.countFastq(text)
    st = m.stats()
    assert st["kmers_added"] == rep * n and st["distinct"] == len(kmers) and st["insert_failures"] == 0, st
    assert st["count_sum"] == rep * n
    assert np.array_equal(m.getKmerCounts(kmers), rep * counts)
a = T.TSXHashMapHIP(l, 0, k)
a.set_path("atomic")
a.countFastq(text); a.countFastq(text)
ka, ca = a.getAllKmers()
km, cm = m.getAllKmers()
ia, im = np.lexsort(ka.T[::-1]), np.lexsort(km.T[::-1])
assert np.array_equal(ka[ia], km[im]) and np.array_equal(ca[ia], cm[im])
m.clear()
m.countFastq(text)
assert np.array_equal(m.getKmerCounts(kmers), counts) and m.stats()["distinct"] == len(kmers)
print("SLABS OK")
''' % (ROOT, l, k, reads)
    env = dict(os.environ, TSX_HIP_SLAB_SEGBITS=str(segbits), TSX_HIP_DEV_WINDOW=str(1 << 20))
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"SLABS OK" in p.stdout, p.stdout.decode()[-3000:]


@pytest.mark.parametrize("l,k", [(26, 31), (25, 21)])
def test_radix_levels_of_512_lists(T, monkeypatch, l, k):
    """2^17 and 2^18 segments: one or both radix levels split 512 ways, their rings leave room for ONE workgroup per CU and
    the walk / level-2 kernels run with 1024 threads (walk_part_kernel<1024>, partition_ring_kernel<1, 1024>).  Reached
    here with 256-slot segments (TSX_HIP_SEG_BITS=8) in a 2^25 / 2^26-slot table; against the oracle."""
    from oracle.oracle import Oracle
    from tsxcount_amd import synth
    monkeypatch.setenv("TSX_HIP_SEG_BITS", "8")
    text = synth.fastq(81, 0, 2500)
    o = Oracle(k, 23, 4, seed=1)
    n = o.count_fastq(text)
    kmers, counts = o.dump()
    m = T.TSXHashMapHIP(l, 0, k)
    m.set_path("partitioned")
    for rep in (1, 2):
        m.countFastq(text)
        st = m.stats()
        assert st["kmers_added"] == rep * n and st["distinct"] == len(kmers) and st["insert_failures"] == 0
        assert np.array_equal(m.getKmerCounts(kmers), rep * counts)
    m.close()


@pytest.mark.parametrize("k,path", [(63, "partitioned"), (63, "atomic"), (31, "partitioned")])
def test_zipf_device_generator_and_analytic_counts(T, k, path):
    """BASELINE config 4's input at test size: the device generator writes the text of tsxcount_amd.synth.zipf_text byte
    for byte, and the table holds, for every k-mer, the count synth.ZipfExpect derives from the generator alone (reads of
    the k-mer's template whose window covers it) -- the hottest k-mer occurs in a third of all reads.  ZipfExpect itself
    is held to a dictionary count on the CPU."""
    import torch
    from tsxcount_amd import synth
    seed, n_reads, rl, nt = 13, 6000, 150, 300
    thr = synth.zipf_thresholds(nt, 1.2)
    nb = T.synth_zipf_device(seed, n_reads, rl, thr)
    buf = torch.empty(nb + 256, dtype=torch.uint8, device="cuda:0")
    buf[nb:] = 10
    torch.cuda.synchronize()
    assert T.synth_zipf_device(seed, n_reads, rl, thr, buf.data_ptr(), nb) == nb
    text = bytes(buf[:nb].cpu().numpy())
    assert text == synth.zipf_text(seed, n_reads, rl, thr)
    ex = synth.ZipfExpect(seed, n_reads, rl, thr, k)
    ref = python_counts(text, k)
    assert ex.total == sum(ref.values()) and ex.distinct == len(ref)
    m = T.TSXHashMapHIP(22, 0, k)
    m.set_path(path)
    m.countFastqDevice(buf.data_ptr(), nb)
    m.sync()
    st = m.stats()
    assert st["kmers_added"] == ex.total and st["distinct"] == ex.distinct and st["count_sum"] == ex.total
    assert st["insert_failures"] == 0 and st["overflow_failures"] == 0
    seqs, cnt = ex.sample(range(0, n_reads, 40))
    assert np.array_equal(m.getKmerCounts(T.encode_many(seqs, k)), cnt)
    assert all(ref[sq] == int(c) for sq, c in zip(seqs, cnt))
    t, p, c = ex.hottest()
    hot = synth.zipf_template(seed, t, p, k)
    assert c > n_reads // 8 and int(m.getKmerCounts(T.encode_many([hot], k))[0]) == c == ref[hot]
    m.close()


@pytest.mark.parametrize("k,l", [(33, 23), (47, 24), (63, 23), (64, 23), (96, 23), (127, 22)])
def test_wide_keys_two_radix_levels(T, k, l):
    """k > 32 on a table split by two radix levels (strip_desc_wide_kernel + walk_log_wide_kernel, level 1 by exact offsets,
    level 2 into sub-lists, build_segments_wide_stream_kernel): against the oracle, and a second count into the same table.
    (A walk fused with level 1 for multi-limb keys was built and measured in round 3 -- rings of 2-word records are half as
    deep, a third of the records took the direct route, 8.1 ms against 4.5 + 5.9 -- and dropped: DESIGN.md section 4.)"""
    from oracle.oracle import Oracle
    from tsxcount_amd import synth
    text = synth.fastq(55 + k, 0, 1500)
    o = Oracle(k, 22, 4, seed=1)
    n = o.count_fastq(text)
    kmers, counts = o.dump()
    m = T.TSXHashMapHIP(l, 0, k)
    m.set_path("partitioned")
    for rep in (1, 2):
        m.countFastq(text)
        st = m.stats()
        assert st["kmers_added"] == rep * n and st["distinct"] == len(kmers) and st["insert_failures"] == 0
        assert np.array_equal(m.getKmerCounts(kmers), rep * counts)
    m.close()
