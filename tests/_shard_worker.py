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
# windows > 1: the text is cut at multiples of 4 KiB, inside lines and records; every window's keys take
# their own exchange while the next window is scanned
# l = 23: a table split by two radix levels -- level 1 of the received keys then runs window by window
# (tsx_hip_shard_l1_window_device), level 2 + build once at the end
# (there: description exchange at world sizes <= 4 -- TSX_HIP_SHARD_MODE=keys runs the key exchange on the same table)
for k, l, n_reads, windows, mode in ((31, 17, 240, 3, "auto"), (21, 15, 30, 1, "auto"), (32, 19, 700, 5, "auto"),
                                     (31, 23, 1500, 3, "auto"), (31, 23, 1500, 3, "keys")):
    os.environ["TSX_HIP_SHARD_MODE"] = mode
    first, cnt = TD.shard_reads(n_reads, rank, world)
    text = synth.fastq(66, first, cnt)
    buf = torch.frombuffer(bytearray(text + b"\n" * 64), dtype=torch.uint8).to("cuda:0")
    m = T.TSXHashMapHIP(l, 0, k, device=0, shard_bits=bits, shard_index=rank)
    sc = TD.ShardedCounter(m, len(text), windows=windows)
    torch.cuda.synchronize()
    for rep in (1, 2):  # second pass merges into segments that already hold data
        sc.step(buf.data_ptr(), len(text))
        assert sc.last["key_sum_diff"] == 0 and sc.last["windows"] == min(windows, -(-len(text) // sc.win_bytes))
        whole = Oracle(k, 21, 4, seed=1)
        whole.count_fastq(synth.fastq(66, 0, n_reads))
        kmers, counts = whole.dump()
        got = m.getKmerCounts(kmers)
        tot = torch.from_numpy(got.astype(np.int64))
        dist.all_reduce(tot)
        assert np.array_equal(tot.numpy().astype(np.uint64), rep * counts), "sum over shards != oracle"
        owned = got > 0
        others = torch.from_numpy(owned.astype(np.int64))
        dist.all_reduce(others)
        assert (others.numpy() == 1).all(), "every k-mer must live on exactly one shard"
        st = m.stats()
        assert st["insert_failures"] == 0 and st["distinct"] == int(owned.sum())
        # the dump of a shard reconstructs full k-mers (owner bits included)
        dk, dc = m.getAllKmers()
        assert np.array_equal(np.sort(dk[:, 0]), np.sort(kmers[owned][:, 0]))
    frac = owned.mean()
    assert 0.4 / world < frac < 1.6 / world, "slot-range ownership should split the keys roughly evenly"
    m.close()
dist.barrier()
dist.destroy_process_group()
print("SHARD OK rank", rank)
