"""Worker of test_minimizer_counting_on_one_gpu (not a pytest file): 2, 3 or 4 ranks share cuda:0, gloo collectives."""
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


def shard(n_reads, shape):
    """even: contiguous equal shards.  skew: rank 0 has NO reads at all, the last rank the rest of what the others leave."""
    if shape != "skew":
        return TD.shard_reads(n_reads, rank, world)
    if rank == 0:
        return 0, 0
    unit = n_reads // world
    first = (rank - 1) * unit
    return first, (n_reads - first) if rank == world - 1 else unit


def fasta_of(fastq_text):
    lines = fastq_text.split(b"\n")
    return b"".join(b">" + lines[i][1:] + b"\n" + lines[i + 1] + b"\n" for i in range(0, len(lines) - 1, 4))


# (k, l, reads, windows, shape): l = 23: two radix levels, which the exchange needs; windows > 1: the text is cut inside lines and
# records; k = 20 / 26 / 32: the ends of the supported range and the switch of the minimizer length (m = k - 15 below 26)
CASES = ((31, 23, 1500, 3, "even"), (31, 23, 1500, 1, "skew"), (20, 23, 900, 2, "even"), (26, 23, 900, 4, "skew"),
         (32, 23, 1200, 3, "fasta"), (27, 23, 600, 2, "even"), (31, 23, 1500, 2, "pieces"), (24, 23, 700, 1, "pieces"),
         (31, 23, 6000, 2, "zipf"))
for k, l, n_reads, windows, shape in CASES:
    # "pieces": the text is described in several pieces (as a text above 2 GiB would be), cut at multiples of 4 KiB inside
    # lines and records; ranks with uneven shards run the same number of pieces all the same
    TD.MinimizerCounter.PIECE = (128 << 10) if shape == "pieces" else (2 << 30)
    if shape == "pieces":
        shape = "skew"
    first, cnt = shard(n_reads, shape)
    if shape == "zipf":   # BASELINE config 4 in small: windows of 40 templates picked with Zipf(1.2) weights -- k-mers with hundreds of
        whole_text = synth.zipf_fastq(67, n_reads, 150, 40, k)   # occurrences, all of one template's on few owners
        recs = whole_text.split(b"\n")
        text = b"".join(x + b"\n" for x in recs[4 * first:4 * (first + cnt)])
    else:
        text = synth.fastq(67, first, cnt)
        whole_text = synth.fastq(67, 0, n_reads)
    lines = 4
    if shape == "fasta":
        text, whole_text, lines = fasta_of(text), fasta_of(whole_text), 2
    buf = torch.frombuffer(bytearray(text + b"\n" * 64), dtype=torch.uint8).to("cuda:0")
    m = T.TSXHashMapHIP(l, 0, k, device=0)
    if shape == "fasta":
        m.set_record_lines(2)
    mc = TD.MinimizerCounter(m, len(text), windows=windows)
    assert mc.windows == mc.pieces * mc.parts and (TD.MinimizerCounter.PIECE > (1 << 20) or mc.pieces > 1)
    torch.cuda.synchronize()
    whole = Oracle(k, 21, 4, seed=1)
    whole.count_fastq(whole_text, lines)
    kmers, counts = whole.dump()
    owner = TD.owner_of(kmers, k, world)

    def check(times):
        assert mc.last["key_sum_diff"] == 0 and mc.last["windows"] == mc.windows
        got = m.getKmerCounts(kmers)
        # every k-mer lives on the rank its minimizer names, with the whole count, and nowhere else
        assert np.array_equal(got, np.where(owner == rank, times * counts, 0).astype(np.uint64)), "counts on rank %d" % rank
        st = m.stats()
        assert st["insert_failures"] == 0 and st["distinct"] == int((owner == rank).sum())
        assert st["count_sum"] == int(times * counts[owner == rank].sum())

    for rep in (1, 2):      # the second pass merges into segments that already hold data
        mc.step(buf.data_ptr(), len(text))
        check(rep)
    dk, dc = m.getAllKmers()
    assert np.array_equal(np.sort(dk[:, 0]), np.sort(kmers[owner == rank][:, 0]))
    for rep in range(2):
        m.clear()
        mc.step(buf.data_ptr(), len(text))
        check(1)
    share = (owner == rank).mean()
    assert shape == "zipf" or 0.5 / world < share < 1.6 / world, "minimizer owners should split the distinct k-mers roughly evenly"
    m.close()
dist.barrier()
dist.destroy_process_group()
print("MINI OK rank", rank)
