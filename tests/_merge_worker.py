"""Worker of test_two_rank_merge_on_one_gpu (not a pytest file)."""
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
config5 = len(sys.argv) > 4 and sys.argv[4] == "config5"
if config5:
    # k = 127, 2-bit in-slot counters, ~138 reads per rank = ~100,000 distinct 127-mers in 2^17 slots (load ~0.8)
    # on every rank before the merge and again after it; each shard is counted 5 times, so every count is
    # >= 5 > 2^2 - 1 and EVERY k-mer carries into the secondary array (sized to hold them all).
    k, l, s, per_rank, reps = 127, 17, 2, 138, 5
    n_reads = per_rank * world
    first, cnt = TD.shard_reads(n_reads, rank, world)
    m = T.TSXHashMapHIP(l, s, k, device=0, overflow_l=18)
    assert m.layout.count_bits == 2 and m.layout.entry_limbs == 4
    text = synth.fastq(55, first, cnt)
    for _ in range(reps):
        m.countFastq(text)
    st = m.stats()
    assert 0.7 < st["distinct"] / (1 << l) < 0.9, st
    assert st["overflow_used"] == st["distinct"] and st["overflow_failures"] == 0 and st["insert_failures"] == 0
    received = TD.merge_tables(m)
    whole = Oracle(k, 21, 4, seed=1)
    whole.count_fastq(synth.fastq(55, 0, n_reads))
    kmers, counts = whole.dump()
    mine = np.array([m.owner(kmers[i], world) == rank for i in range(len(kmers))])
    got = m.getKmerCounts(kmers)
    assert np.array_equal(got[mine], reps * counts[mine]), "owned k-mers must carry the merged count"
    assert (got[~mine] == 0).all(), "foreign k-mers must be gone after the merge"
    st = m.stats()
    assert st["distinct"] == int(mine.sum()) and received >= st["distinct"]
    assert 0.6 < st["distinct"] / (1 << l) < 0.95, st   # load ~0.8 again: every rank owns 1/world of world x as many
    assert st["count_sum"] == reps * int(counts[mine].sum())
    assert st["overflow_failures"] == 0 and st["insert_failures"] == 0 and st["lock_timeouts"] == 0
    dk, dc = m.getAllKmers()
    a, b = np.lexsort(dk.T[::-1]), np.lexsort(kmers[mine].T[::-1])
    assert np.array_equal(dk[a], kmers[mine][b]) and np.array_equal(dc[a], reps * counts[mine][b])
    tot = torch.tensor([st["distinct"]], dtype=torch.int64)
    dist.all_reduce(tot)
    assert int(tot.item()) == len(kmers)
    m.close()
else:
    for k, l in ((31, 18), (63, 18)):
        n_reads = 160
        first, cnt = TD.shard_reads(n_reads, rank, world)
        m = T.TSXHashMapHIP(l, 0, k, device=0)
        m.countFastq(synth.fastq(55, first, cnt))
        received = TD.merge_tables(m)
        whole = Oracle(k, 20, 4, seed=1)
        whole.count_fastq(synth.fastq(55, 0, n_reads))
        kmers, counts = whole.dump()
        mine = np.array([m.owner(kmers[i], world) == rank for i in range(len(kmers))])
        got = m.getKmerCounts(kmers)
        assert np.array_equal(got[mine], counts[mine]), "owned k-mers must carry the merged count"
        assert (got[~mine] == 0).all(), "foreign k-mers must be gone after the merge"
        assert m.stats()["distinct"] == int(mine.sum())
        tot = torch.tensor([m.stats()["distinct"]], dtype=torch.int64)
        dist.all_reduce(tot)
        assert int(tot.item()) == len(kmers)
        m.close()
dist.barrier()
dist.destroy_process_group()
print("MERGE OK rank", rank)
