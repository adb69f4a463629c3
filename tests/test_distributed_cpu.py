"""N > 1 path on CPU: two gloo ranks shard the reads, count their shard into a
local table (the oracle stands in for the GPU table here), and run the SAME
exchange_segments() the GPU path uses; the merged result must equal a single
table over all reads."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _owner(kmer_row, nranks):
    # same function as owner_of()/tsx_hip_owner_host (splitmix finaliser chain)
    M = (1 << 64) - 1
    z = 0x243F6A8885A308D3
    for x in kmer_row:
        z ^= int(x)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z = z ^ (z >> 31)
    return ((z >> 32) * nranks) >> 32


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from tsxcount_amd import synth
    from tsxcount_amd.distributed import exchange_segments, shard_reads
    k, n_reads = 21, 48
    first, cnt = shard_reads(n_reads, rank, world)
    o = Oracle(k, 18, 4, seed=1)
    o.count_fastq(synth.fastq(77, first, cnt))
    kmers, counts = o.dump()
    owner = np.array([_owner(r, world) for r in kmers], dtype=np.int64)
    order = np.argsort(owner, kind="stable")
    seg = np.bincount(owner, minlength=world).astype(np.int64)
    rk, rc = exchange_segments(torch.from_numpy(kmers[order].view(np.int64)),
                               torch.from_numpy(counts[order].view(np.int64)), torch.from_numpy(seg))
    merged = Oracle(k, 18, 4, seed=1)
    rk = rk.numpy().view(np.uint64)
    rc = rc.numpy().view(np.uint64)
    got = {}
    for i in range(rk.shape[0]):
        assert _owner(rk[i], world) == rank
        key = rk[i].tobytes()
        got[key] = got.get(key, 0) + int(rc[i])
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_merge_equals_single_table(world):
    """world = 8 is the size the target node runs: the exchange (sizes, rows, checksums, the agreement
    all-reduce) with eight real processes, on CPU."""
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle
    from tsxcount_amd import synth
    from tsxcount_amd.distributed import shard_reads
    assert shard_reads(10, 0, 3) == (0, 4) and shard_reads(10, 1, 3) == (4, 3) and shard_reads(10, 2, 3) == (7, 3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = Oracle(21, 18, 4, seed=1)
    whole.count_fastq(synth.fastq(77, 0, 48))
    kmers, counts = whole.dump()
    expect = {kmers[i].tobytes(): int(counts[i]) for i in range(len(kmers))}
    merged = {}
    for r in range(world):
        assert not (set(results[r]) & set(merged))  # owners are disjoint
        merged.update(results[r])
    assert merged == expect


def _geometry_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsxcount_amd.distributed import TorchComm, agreed_max, window_geometry, window_of
    comm = TorchComm()
    mine = [0, 40 << 20, 150 << 20][rank]          # rank 0 has NO reads; the ranks' texts straddle the 32 MiB window steps
    big = agreed_max(mine, comm, torch.device("cpu"))
    windows, win_bytes = window_geometry(big)
    wins = [window_of(i, mine, win_bytes) for i in range(windows)]
    # every window of every rank is a round of collectives: count them with a real one
    for off, ln in wins:
        t = torch.tensor([ln], dtype=torch.int64)
        dist.all_reduce(t)
    q.put((rank, big, windows, win_bytes, wins))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_uneven_and_empty_shards_agree_on_the_windows():
    """ShardedCounter's window geometry: derived from the LARGEST text of any rank, so that a rank with a
    short or empty shard runs the same number of rounds of collectives (empty windows at the 16-byte-aligned end of its
    text) -- three gloo ranks with 0, 40 MiB and 150 MiB of text."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_geometry_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({(r[1], r[2], r[3]) for r in res}) == 1, "every rank must hold the same geometry"
    big, windows, win_bytes = res[0][1:4]
    assert big == 150 << 20 and windows == 4 and win_bytes % 4096 == 0 and windows * win_bytes >= big
    for rank, _, _, _, wins in res:
        mine = [0, 40 << 20, 150 << 20][rank]
        assert len(wins) == windows and sum(ln for _, ln in wins) == mine
        assert all(off % 16 == 0 and off + ln <= mine for off, ln in wins)
        assert [ln for _, ln in wins] == sorted((ln for _, ln in wins), reverse=True)   # full windows first, then empty ones


def _mz_owner_py(x, k, nranks):
    """The minimizer owner rule of csrc/tsx_minimizer.h, restated: m = min(11, k - 15); value of an m-mer = bit 31 if it
    starts or ends with AAA, below it ((x * 0x9E3779 + 0x2B5A3D) mod 4^m) top-aligned in 31 bits; the minimizer is the m-mer
    of the k-mer with the smallest value; owner = (bits 8..23 of (value >> (31 - 2m)) * 0xC2B2AF) * nranks >> 16."""
    m = min(11, k - 15)
    mb = 2 * m
    mask = (1 << mb) - 1
    best = 0xFFFFFFFF
    for j in range(k - m + 1):
        v = (x >> (2 * j)) & mask
        pen = 0x80000000 if ((v & 63) == 0 or (v >> (mb - 6)) == 0) else 0
        key = pen | ((((v * 0x9E3779 + 0x2B5A3D) & 0xFFFFFFFF) & mask) << (31 - mb))
        best = min(best, key)
    t = best >> (31 - mb)
    u = (((t * 0xC2B2AF) & 0xFFFFFFFF) >> 8) & 0xFFFF
    return (u * nranks) >> 16


def test_minimizer_owner_function():
    """tsx_hip_mini_owner_host against its restatement, and what the exchange relies on: owners in range, consecutive k-mers
    of a random sequence mostly share one (runs of about 12), every rank gets its share."""
    import numpy as np
    import tsxcount_amd as T
    from tsxcount_amd import distributed as TD
    rng = np.random.default_rng(5)
    for k in (20, 21, 25, 26, 27, 31, 32):
        for world in (1, 2, 3, 8, 16):
            xs = rng.integers(0, 1 << 62, size=300, dtype=np.uint64) & np.uint64((1 << (2 * k)) - 1)
            xs[:4] = [int(T.encode(b * k, k)[0]) for b in "ACGT"]
            got = TD.owner_of(xs, k, world)
            assert got.max() < world
            assert [int(g) for g in got] == [_mz_owner_py(int(x), k, world) for x in xs], (k, world)
    k, world, n = 31, 8, 60000
    codes = rng.integers(0, 4, size=n + k).astype(np.uint64)
    km = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        km |= codes[j:j + n] << np.uint64(2 * j)
    own = TD.owner_of(km, k, world)
    runs = 1 + int((own[1:] != own[:-1]).sum())
    assert 9 < n / runs < 16, n / runs
    share = np.bincount(own, minlength=world) / n
    assert share.min() > 0.08 and share.max() < 0.17, share
    import pytest
    with pytest.raises(T.TSXException):
        TD.owner_of(km[:4], 19, 8)      # k < 20: the 16 windows of a strip would not share a core


def _lists_worker(rank, world, port, q):
    """The minimizer exchange's collectives on CPU: every rank groups the k-mer occurrences of ITS reads by
    owner = f(minimizer) (tsx_hip_mini_owner_host), the groups -- pieces that are not neighbours in memory, some empty --
    travel through TorchComm.all_to_all_lists after their sizes through all_to_all, and the receiver counts what it got."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tsxcount_amd as T
    from tsxcount_amd import distributed as TD
    from tsxcount_amd import synth
    comm = TD.TorchComm()
    k, n_reads = 31, 10
    first, cnt = (0, 0) if rank == 0 else TD.shard_reads(n_reads, rank - 1, world - 1)   # rank 0 has no reads at all
    text = synth.fastq(78, first, cnt)
    lines = text.split(b"\n")
    occ = [T.encode(s[i:i + k], k)[0] for s in lines[1::4] for i in range(len(s) - k + 1)]
    occ = np.array(occ, dtype=np.uint64)
    owner = TD.owner_of(occ, k, world) if len(occ) else np.zeros(0, dtype=np.uint32)
    cap = len(occ) + 8      # the lists sit cap apart, as the split kernel leaves them
    buf = torch.zeros((world * cap,), dtype=torch.int64)
    sizes = []
    for o in range(world):
        mine = occ[owner == o].view(np.int64)
        buf[o * cap:o * cap + len(mine)] = torch.from_numpy(mine.copy())
        sizes.append(len(mine))
    meta_in = torch.tensor(sizes, dtype=torch.int64).view(world, 1)
    meta_out = torch.empty_like(meta_in)
    comm.all_to_all(meta_out, meta_in)
    rs = [int(x) for x in meta_out.view(-1).tolist()]
    recv = torch.empty((sum(rs),), dtype=torch.int64)
    outs, at = [], 0
    for p in range(world):
        outs.append(recv[at:at + rs[p]])
        at += rs[p]
    comm.all_to_all_lists(outs, [buf[o * cap:o * cap + sizes[o]] for o in range(world)])
    got = recv.numpy().view(np.uint64)
    assert len(got) == 0 or (TD.owner_of(got, k, world) == rank).all()
    keys, counts = np.unique(got, return_counts=True)
    q.put((rank, dict(zip(keys.tolist(), counts.tolist())), len(occ)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_minimizer_lists_exchange_on_cpu(world):
    import tsxcount_amd as T
    from tsxcount_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lists_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every occurrence arrived exactly once, on the owner of its k-mer
    k = 31
    text = synth.fastq(78, 0, 10)
    lines = text.split(b"\n")
    want = {}
    for sq in lines[1::4]:
        for i in range(len(sq) - k + 1):
            key = int(T.encode(sq[i:i + k], k)[0])
            want[key] = want.get(key, 0) + 1
    merged = {}
    for _, got, _ in res:
        for key, c in got.items():
            assert key not in merged, "a k-mer on two ranks"
            merged[key] = c
    assert merged == want
    assert sum(n for _, _, n in res) == sum(want.values())
