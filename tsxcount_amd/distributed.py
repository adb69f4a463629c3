"""Multi-GPU plumbing for --mode=HIP: reads shard across ranks (one process per
GPU), every rank counts its shard into its own table with no collective on the
data path, then ONE exchange step merges the per-GPU tables: each rank groups
its table entries by owner rank (tsx_hip_partition_device), the groups travel
through an all-to-all (RCCL over xGMI on GPUs, gloo on CPU in the tests), and
the owner re-inserts what it receives.  After the merge rank r holds the
complete counts of every k-mer with owner(kmer) == r.

torch.distributed is used for the collective only.
"""
import ctypes

import torch
import torch.distributed as dist


def shard_reads(n_reads_total, rank, world):
    """Contiguous read shard [first, first+count) of rank `rank`."""
    base, rem = divmod(n_reads_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def exchange_segments(kmers, counts, seg_counts, group=None):
    """All-to-all of owner-grouped table entries.

    kmers  [total, wk] int64 (uint64 bit patterns), rows grouped by destination rank
    counts [total] int64
    seg_counts [world] int64 -- rows destined to each rank
    Returns (recv_kmers, recv_counts) with everything this rank owns.
    """
    world = dist.get_world_size(group)
    wk = kmers.shape[1]
    send_sizes = seg_counts.to(torch.int64).contiguous()
    recv_sizes = torch.empty_like(send_sizes)
    dist.all_to_all_single(recv_sizes, send_sizes, group=group)
    ss = [int(x) for x in send_sizes.tolist()]
    rs = [int(x) for x in recv_sizes.tolist()]
    assert len(ss) == world and sum(ss) == kmers.shape[0]
    recv_k = torch.empty((sum(rs), wk), dtype=kmers.dtype, device=kmers.device)
    recv_c = torch.empty((sum(rs),), dtype=counts.dtype, device=counts.device)
    dist.all_to_all_single(recv_k, kmers.contiguous(), output_split_sizes=rs, input_split_sizes=ss, group=group)
    dist.all_to_all_single(recv_c, counts.contiguous(), output_split_sizes=rs, input_split_sizes=ss, group=group)
    return recv_k, recv_c


def merge_tables(hmap, group=None):
    """Merge the per-GPU tables in place (see module docstring).  Returns the
    number of entries this rank received."""
    from . import _check
    world = dist.get_world_size(group)
    dev = torch.device("cuda", hmap.device)
    n = hmap.stats()["distinct"]
    kmers = torch.empty((max(n, 1), hmap.wk), dtype=torch.int64, device=dev)
    counts = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    seg = torch.zeros((world,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    _check(hmap._lib.tsx_hip_partition_device(hmap.handle, world, ctypes.c_void_p(kmers.data_ptr()),
                                              ctypes.c_void_p(counts.data_ptr()), max(n, 1),
                                              ctypes.c_void_p(seg.data_ptr()), None))
    if dist.get_backend(group) == "gloo":
        # CPU collective (tests: several ranks sharing one GPU): stage through host memory
        rk, rc = exchange_segments(kmers[:n].cpu(), counts[:n].cpu(), seg.cpu(), group)
        recv_k, recv_c = rk.to(dev), rc.to(dev)
    else:
        recv_k, recv_c = exchange_segments(kmers[:n], counts[:n], seg, group)
    torch.cuda.synchronize(dev)
    hmap.clear()
    if recv_k.shape[0]:
        _check(hmap._lib.tsx_hip_add_kmers_device(hmap.handle, ctypes.c_void_p(recv_k.data_ptr()),
                                                  ctypes.c_void_p(recv_c.data_ptr()), recv_k.shape[0], None))
    hmap.sync()
    return int(recv_k.shape[0])
