"""The C++ multi-GPU host (csrc/tsx_multi.cpp, tsx_hip_group_*): record cuts on the CPU; on the GPU the group with
its collective as device copies (2 and 8 ranks sharing cuda:0 -- RCCL wants one GPU per rank) and through the RCCL
API with the one rank a one-GPU box can give it; the CLI's --gpus."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, python_counts


def _is_record_boundary(text, cut, lines_per_record):
    """cut is 0, len(text), or right behind the terminator of the last line of a record (reference rules:
    empty lines do not count, FastXReader.h:365-370)."""
    if cut in (0, len(text)):
        return True
    if text[cut - 1:cut] != b"\n":
        return False
    lines = [l for l in text[:cut].split(b"\n") if l]
    return len(lines) % lines_per_record == 0


@pytest.mark.parametrize("lines", [4, 2])
def test_record_cuts_are_record_boundaries(lines):
    import tsxcount_amd as T
    from tsxcount_amd import synth
    rng = np.random.default_rng(3)
    fq = synth.fastq(7, 0, 61)
    if lines == 2:
        ls = fq.split(b"\n")
        fq = b"".join(b">" + ls[i][1:] + b"\n" + ls[i + 1] + b"\n" for i in range(0, len(ls) - 1, 4))
    # empty lines sprinkled in (they are dropped by the reader and must not shift the record count), no final newline
    parts = fq.split(b"\n")
    for _ in range(25):
        parts.insert(int(rng.integers(0, len(parts))), b"")
    messy = b"\n".join(parts).rstrip(b"\n")
    for text in (fq, messy, b"", b"\n\n\n", fq[:200]):
        for n in (1, 2, 3, 8, 64):
            cuts = T.cut_records(text, n, lines)
            assert cuts[0] == 0 and cuts[-1] == len(text) and cuts == sorted(cuts)
            assert all(_is_record_boundary(text, c, lines) for c in cuts), (n, cuts)
            # the shards together hold every k-mer of the text exactly once
            whole = python_counts(text, 21, lines)
            got = sum((python_counts(text[cuts[i]:cuts[i + 1]], 21, lines) for i in range(n)), type(whole)())
            assert got == whole
    # a balanced text gives balanced shards
    cuts = T.cut_records(fq, 8, lines)
    sizes = np.diff(cuts)
    assert sizes.min() > 0.5 * len(fq) / 8 and sizes.max() < 1.5 * len(fq) / 8


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,k,l,s", [(2, 31, 18, 0), (8, 31, 17, 0), (4, 63, 17, 0), (8, 127, 16, 2), (3, 21, 17, 4)])
def test_group_counts_equal_the_oracle(ranks, k, l, s):
    """N tables on cuda:0, the collective as device copies behind a barrier: record shards, per-GPU counts, the merge
    (partition by owner, all-to-all, clear, re-insert with counts), lookups at the owner.  Any N, any k (config 5's
    k = 127 with --s=2 counters: every hot k-mer carries into the secondary array before and after the merge)."""
    import tsxcount_amd as T
    from oracle.oracle import Oracle
    from tsxcount_amd import synth
    text = synth.fastq(91, 0, 90 if k < 100 else 40)
    o = Oracle(k, 20, 4, seed=1)
    n = o.count_fastq(text)
    kmers, counts = o.dump()
    g = T.TSXHashMapHIPGroup(ranks, l, s, k, devices=[0] * ranks, comm="copy")
    assert g.comm_name() == "copy"
    for rep in (1, 2):      # a second count after clear(): the same tables, the same answer
        g.countFastq(text)
        st = g.stats()
        assert st["distinct"] == len(kmers) and st["count_sum"] == n and st["insert_failures"] == 0
        assert np.array_equal(g.getKmerCounts(kmers), counts)
        per = [g.rank_stats(r)["distinct"] for r in range(ranks)]
        assert sum(per) == len(kmers) and min(per) > 0.5 * len(kmers) / ranks      # every k-mer on one GPU, spread evenly
        assert 0 < g.exchanged_entries() <= sum(per) * 2
        g.clear()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,k,l", [(2, 31, 23), (3, 20, 23), (8, 31, 23), (5, 32, 23), (1, 26, 23)])
def test_group_minimizer_exchange_equals_the_oracle(ranks, k, l):
    """The C++ group with the minimizer exchange (tsx_hip_group_set_exchange 1): record shards, every GPU describes and
    splits its own, the lists travel (device copies behind a barrier here), every GPU walks what it owns, one build; the
    homopolymer totals on their owners.  Every k-mer on the rank tsx_hip_mini_owner_host names and only there; a second
    count doubles everything; FASTA; outside 20 <= k <= 32 the mode is refused."""
    import tsxcount_amd as T
    from oracle.oracle import Oracle
    from tsxcount_amd import distributed as TD
    from tsxcount_amd import synth
    text = synth.fastq(93, 0, 700)
    o = Oracle(k, 21, 4, seed=1)
    n = o.count_fastq(text)
    kmers, counts = o.dump()
    owner = TD.owner_of(kmers, k, ranks)
    g = T.TSXHashMapHIPGroup(ranks, l, 0, k, devices=[0] * ranks, comm="copy", exchange="mini")
    for rep in (1, 2):
        g.countFastq(text)
        st = g.stats()
        assert st["distinct"] == len(kmers) and st["count_sum"] == rep * n and st["insert_failures"] == 0
        assert np.array_equal(g.getKmerCounts(kmers), rep * counts)
        for r in range(ranks):
            assert g.rank_stats(r)["distinct"] == int((owner == r).sum())
    g.clear()
    g.set_record_lines(2)
    lines = text.split(b"\n")
    fasta = b"".join(b">" + lines[i][1:] + b"\n" + lines[i + 1] + b"\n" for i in range(0, len(lines) - 1, 4))
    g.countFastq(fasta)
    assert np.array_equal(g.getKmerCounts(kmers), counts)
    g.close()
    with pytest.raises(T.TSXException):
        T.TSXHashMapHIPGroup(2, 19, 0, 63, devices=[0, 0], comm="copy", exchange="mini")


@pytest.mark.gpu
def test_group_of_one_through_the_rccl_api():
    """ncclCommInitAll with the one GPU of this box, the merge's all-to-all as grouped ncclSend/ncclRecv from rank 0
    to rank 0: the RCCL leg of the C++ host through its API (N > 1 needs one GPU per rank)."""
    import tsxcount_amd as T
    from oracle.oracle import Oracle
    from tsxcount_amd import synth
    text = synth.fastq(92, 0, 120)
    o = Oracle(31, 20, 4, seed=1)
    n = o.count_fastq(text)
    kmers, counts = o.dump()
    g = T.TSXHashMapHIPGroup(1, 19, 0, 31, comm="rccl")
    assert g.comm_name() == "rccl"
    g.countFastq(text)
    st = g.stats()
    assert st["distinct"] == len(kmers) and st["count_sum"] == n
    assert np.array_equal(g.getKmerCounts(kmers), counts)
    g.close()
    # the minimizer exchange's lists (pieces that are not neighbours in memory) through the same grouped send / recv
    g = T.TSXHashMapHIPGroup(1, 23, 0, 31, comm="rccl", exchange="mini")
    g.countFastq(text)
    st = g.stats()
    assert st["distinct"] == len(kmers) and st["count_sum"] == n
    assert np.array_equal(g.getKmerCounts(kmers), counts)
    g.close()
    with pytest.raises(T.TSXException):       # RCCL refuses two ranks on one GPU: reported, not attempted
        T.TSXHashMapHIPGroup(2, 19, 0, 31, devices=[0, 0], comm="rccl")


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--gpus=2", "--comm=copy", "--devices=0,0"], ["--gpus=8", "--comm=copy", "--devices=0,0,0,0,0,0,0,0", "--exchange=merge"],
                                  ["--gpus=1"]])
def test_cli_gpus_check_passes_on_golden(tmp_path, args):
    """tsxCount --mode=HIP --gpus=N --check on the reference's own fixture: one command runs the job
    (src/mains/main.cpp:404-507), the reference's console lines, `total errors0`."""
    text = open(os.path.join(GOLDEN, "small_t7.1000.fastq"), "rb").read()
    fq = tmp_path / "small_t7.1000.fastq"
    fq.write_bytes(text)
    with gzip.open(os.path.join(GOLDEN, "small_t7.1000.fastq.14.count.gz"), "rb") as f:
        (tmp_path / "small_t7.1000.fastq.14.count").write_bytes(f.read())
    exe = os.path.join(ROOT, "tsxcount_amd", "bin", "tsxCount")
    p = subprocess.run([exe, "--input=%s" % fq, "--mode=HIP", "--l=20", "--check", "--checkabort"] + args,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, out + err
    assert "Added a total of 194697 different kmers" in out and "total errors0" in out
    assert "entries moved between GPUs by the merge" in err and "exchange: per-GPU tables merged" in err


@pytest.mark.gpu
@pytest.mark.parametrize("args,how", [(["--gpus=4", "--comm=copy", "--devices=0,0,0,0"], "minimizer"), (["--gpus=2", "--comm=copy", "--devices=0,0"], "merged"),
                                      (["--gpus=3", "--comm=copy", "--devices=0,0,0", "--exchange=mini"], "minimizer")])
def test_cli_gpus_exchange_choice(tmp_path, args, how):
    """tsxCount --mode=HIP --gpus=N at k = 31: the minimizer exchange from 4 GPUs on (or on request), the table merge below;
    the number of different k-mers it reports is the oracle's."""
    from oracle.oracle import Oracle
    from tsxcount_amd import synth
    text = synth.fastq(95, 0, 600)
    o = Oracle(31, 21, 4, seed=1)
    o.count_fastq(text)
    kmers, _ = o.dump()
    fq = tmp_path / "reads.fastq"
    fq.write_bytes(text)
    exe = os.path.join(ROOT, "tsxcount_amd", "bin", "tsxCount")
    p = subprocess.run([exe, "--input=%s" % fq, "--mode=HIP", "--k=31", "--l=23"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, out + err
    assert "Added a total of %d different kmers" % len(kmers) in out, out + err
    assert ("exchange: minimizer owners" in err) == (how == "minimizer"), err
