"""Blocked gzip (BGZF) input inflated on the device (tsxcount_amd/csrc/tsx_inflate.h).  The reference reads `.gz`
through zlib (FastXReader.h:178-206), which takes BGZF as ordinary multi-member gzip: the expectation here is
zlib's own output (Python's gzip module) and the counts of the plain text."""
import gzip
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _texts():
    from tsxcount_amd import synth
    rng = np.random.default_rng(5)
    fastq = synth.fastq(17, 0, 400)
    return {
        "fastq": fastq,
        "random bytes (stored or barely compressed blocks)": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
        "one byte repeated (distance 1 matches of length 258)": b"A" * 300000,
        "short period": (b"ACGTTGCA" * 7 + b"\n") * 6000,
        "tiny (fixed Huffman codes)": b"ACGT\n",
        "empty": b"",
        "exactly one block": bytes(rng.integers(65, 70, 65280, dtype=np.uint8)),
        "one byte over a block": bytes(rng.integers(65, 70, 65281, dtype=np.uint8)),
    }


def test_bgzf_index_accepts_bgzf_and_nothing_else():
    import tsxcount_amd as T
    d = b"@r\nACGT\n+\nIIII\n" * 9000
    z = T.bgzf_compress(d)
    assert gzip.decompress(z) == d                      # zlib reads it as multi-member gzip
    assert T.bgzf_index(z) == (len(d) // 65280 + 2, len(d))   # the data members + the empty end-of-file member
    assert T.bgzf_index(gzip.compress(d)) is None       # a single-stream gzip file has no BC field
    assert T.bgzf_index(z[:-5]) is None                 # truncated
    assert T.bgzf_index(b"") is None and T.bgzf_index(d) is None


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_device_inflate_equals_zlib(level):
    import tsxcount_amd as T
    for name, text in _texts().items():
        z = T.bgzf_compress(text, level=level)
        assert gzip.decompress(z) == text
        assert T.bgzf_inflate(z) == text, name


@pytest.mark.gpu
def test_device_inflate_small_blocks_and_fixed_codes():
    """Blocks of 1..300 bytes: zlib emits fixed-Huffman blocks for them; thousands of members per launch."""
    import tsxcount_amd as T
    from tsxcount_amd import synth
    text = synth.fastq(3, 0, 300)
    for block in (1, 7, 64, 300):
        part = text[: 400 * block]
        z = T.bgzf_compress(part, level=6, block=block)
        assert T.bgzf_inflate(z) == part


@pytest.mark.gpu
def test_count_fastq_from_bgzf_equals_plain_text():
    import tsxcount_amd as T
    text = open(os.path.join(ROOT, "tests", "golden", "small_t7.1000.fastq"), "rb").read()
    z = T.bgzf_compress(text)
    a, b = T.TSXHashMapHIP(20, 0, 14), T.TSXHashMapHIP(20, 0, 14)
    a.countFastq(text)
    b.countFastqBgzf(z)
    ka, ca = a.getAllKmers()
    kb, cb = b.getAllKmers()
    ia, ib = np.lexsort(ka.T[::-1]), np.lexsort(kb.T[::-1])
    assert np.array_equal(ka[ia], kb[ib]) and np.array_equal(ca[ia], cb[ib])
    assert a.stats()["kmers_added"] == b.stats()["kmers_added"]
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,block", [(31, 65280), (127, 4000), (21, 700)])
def test_bgzf_is_inflated_and_counted_in_batches(monkeypatch, k, block):
    """A large .fastq.gz is inflated a batch of members at a time into two text buffers (3 GiB of text per batch by
    default; the smallest batch, 128 KiB, here): lines, records and k-mers run across the batch seams, members of 4000
    and 700 bytes put dozens to hundreds of members into a batch.  Counts must equal those of the plain text, and the
    host copy of the inflated text zlib's."""
    import tsxcount_amd as T
    from tsxcount_amd import synth
    text = synth.fastq(23, 0, 700)           # 1.2 MB: ten batches
    z = T.bgzf_compress(text, level=1, block=block)
    monkeypatch.setenv("TSX_HIP_BGZF_BATCH", "1")      # clamped to the minimum
    assert T.bgzf_inflate(z) == text
    a, b = T.TSXHashMapHIP(21, 0, k), T.TSXHashMapHIP(21, 0, k)
    a.countFastq(text)
    b.countFastqBgzf(z)
    sa, sb = a.stats(), b.stats()
    assert sa["kmers_added"] == sb["kmers_added"] > 0 and sa["distinct"] == sb["distinct"]
    ka, ca = a.getAllKmers()
    assert np.array_equal(b.getKmerCounts(ka), ca)
    a.close(); b.close()


@pytest.mark.gpu
def test_damaged_members_are_refused():
    import tsxcount_amd as T
    from tsxcount_amd import synth
    text = synth.fastq(9, 0, 300)
    z = bytearray(T.bgzf_compress(text))
    m = T.TSXHashMapHIP(18, 0, 21)
    # a flipped byte in the deflate data of the first member: a decode error or, at the latest, the CRC-32
    for pos in (30, 200, 2000):
        bad = bytearray(z)
        bad[pos] ^= 0x5A
        with pytest.raises(T.TSXException):
            m.countFastqBgzf(bytes(bad))
    # a wrong CRC-32 in the trailer of the first member
    first = int.from_bytes(z[16:18], "little") + 1
    bad = bytearray(z)
    bad[first - 8] ^= 1
    with pytest.raises(T.TSXException):
        m.countFastqBgzf(bytes(bad))
    m.clear()
    m.countFastqBgzf(bytes(z))          # and the intact file still counts
    assert m.stats()["kmers_added"] > 0
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("blocked", [True, False])
def test_cli_reads_gz_input(tmp_path, blocked):
    """tsxCount --input=reads.fastq.gz --check: a BGZF file is inflated on the device, any other gzip stream by zlib on
    the host (FastXReader.h:178-206 reads both through zlib); the counts are those of the golden fixture either way."""
    import subprocess
    import tsxcount_amd as T
    golden = os.path.join(ROOT, "tests", "golden")
    text = open(os.path.join(golden, "small_t7.1000.fastq"), "rb").read()
    fq = tmp_path / "small_t7.1000.fastq.gz"
    fq.write_bytes(T.bgzf_compress(text) if blocked else gzip.compress(text))
    with gzip.open(os.path.join(golden, "small_t7.1000.fastq.14.count.gz"), "rb") as f:
        (tmp_path / "small_t7.1000.fastq.gz.14.count").write_bytes(f.read())
    exe = os.path.join(ROOT, "tsxcount_amd", "bin", "tsxCount")
    p = subprocess.run([exe, "--input=%s" % fq, "--mode=HIP", "--check", "--checkabort"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, out + err
    assert "Added a total of 194697 different kmers" in out and "total errors0" in out
    assert ("inflated on the device" in err) == blocked
