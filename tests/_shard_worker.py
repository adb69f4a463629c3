"""Worker of test_sharded_counting_on_one_gpu (not a pytest file): 2 or 4 ranks share cuda:0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import tsxcount_amd as T  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tsxcount_amd import distributed as TD  # noqa: E402
from tsxcount_amd import synth  # noqa: E402

dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
bits = world.bit_length() - 1


def shard(n_reads, shape):
    """even: contiguous equal shards.  skew: rank 0 has NO reads at all, the last rank twice the share of the others
    (the ranks' texts differ in size by more than a window: every rank must still run the same collectives)."""
    if shape != "skew":
        return TD.shard_reads(n_reads, rank, world)
    if rank == 0:
        return 0, 0
    unit = n_reads // world       # ranks 1 .. world-2 take one unit each, the last rank the rest
    first = (rank - 1) * unit
    return first, (n_reads - first) if rank == world - 1 else unit


def fasta_of(fastq_text):
    """The same reads as FASTA records (header line, one sequence line)."""
    lines = fastq_text.split(b"\n")
    return b"".join(b">" + lines[i][1:] + b"\n" + lines[i + 1] + b"\n" for i in range(0, len(lines) - 1, 4))


# windows > 1: the text is cut at multiples of 4 KiB, inside lines and records; every window's keys take
# their own exchange while the next window is scanned
# l = 23: a table split by two radix levels -- level 1 of the received keys then runs window by window
# (tsx_hip_shard_l1_window_device), level 2 + build once at the end
# (there: description exchange at world sizes <= 4 -- TSX_HIP_SHARD_MODE=keys runs the key exchange on the same table)
CASES = ((31, 17, 240, 3, "auto", "even"), (21, 15, 30, 1, "auto", "even"), (32, 19, 700, 5, "auto", "even"),
         (31, 23, 1500, 3, "auto", "even"), (31, 23, 1500, 3, "keys", "even"),
         (31, 23, 1500, 3, "auto", "skew"), (31, 23, 1500, 3, "keys", "skew"), (25, 17, 120, 4, "auto", "skew"),
         (31, 23, 900, 3, "auto", "fasta"), (31, 23, 900, 3, "keys", "fasta"), (27, 16, 60, 2, "auto", "fasta"))
for k, l, n_reads, windows, mode, shape in CASES:
    os.environ["TSX_HIP_SHARD_MODE"] = mode
    first, cnt = shard(n_reads, shape)
    text = synth.fastq(66, first, cnt)
    whole_text = synth.fastq(66, 0, n_reads)
    lines = 4
    if shape == "fasta":
        text, whole_text, lines = fasta_of(text), fasta_of(whole_text), 2
    buf = torch.frombuffer(bytearray(text + b"\n" * 64), dtype=torch.uint8).to("cuda:0")
    m = T.TSXHashMapHIP(l, 0, k, device=0, shard_bits=bits, shard_index=rank)
    sc = TD.ShardedCounter(m, len(text), windows=windows)     # every rank passes ITS size: the counter agrees on the largest
    if shape == "fasta":
        m.set_record_lines(2)      # after the counter was made: the key buffers are sized at the first step
    torch.cuda.synchronize()
    whole = Oracle(k, 21, 4, seed=1)
    whole.count_fastq(whole_text, lines)
    kmers, counts = whole.dump()

    def check(times):
        assert sc.last["key_sum_diff"] == 0 and sc.last["windows"] == sc.windows
        got = m.getKmerCounts(kmers)
        tot = torch.from_numpy(got.astype(np.int64))
        dist.all_reduce(tot)
        assert np.array_equal(tot.numpy().astype(np.uint64), times * counts), "sum over shards != oracle"
        owned = got > 0
        others = torch.from_numpy(owned.astype(np.int64))
        dist.all_reduce(others)
        assert (others.numpy() == 1).all(), "every k-mer must live on exactly one shard"
        st = m.stats()
        assert st["insert_failures"] == 0 and st["distinct"] == int(owned.sum())
        return got, owned

    for rep in (1, 2):  # second pass merges into segments that already hold data
        sc.step(buf.data_ptr(), len(text))
        got, owned = check(rep)
        # the dump of a shard reconstructs full k-mers (owner bits included)
        dk, dc = m.getAllKmers()
        assert np.array_equal(np.sort(dk[:, 0]), np.sort(kmers[owned][:, 0]))
    for rep in range(3):   # clear() works on the map's own stream, the step on the counter's: three times, exact each time
        m.clear()
        sc.step(buf.data_ptr(), len(text))
        check(1)
    frac = owned.mean()
    assert 0.4 / world < frac < 1.6 / world, "slot-range ownership should split the keys roughly evenly"
    m.close()
dist.barrier()
dist.destroy_process_group()
print("SHARD OK rank", rank)
