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
