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


A2A_CHUNK = 16 << 20  # elements (128 MiB of int64) per pair and round


def exchange_rows(inp, in_sizes, out_sizes, group=None, chunk=A2A_CHUNK):
    """All-to-all of int64 rows, robust against large messages.

    `inp` is 1-D int64, grouped by destination rank (in_sizes[p] elements for rank p);
    the result is grouped by source rank (out_sizes[p] elements from rank p).

    Measured on this stack (ROCm 7.2 RCCL, torch 2.10): one all_to_all_single of
    >= 1.2 GB delivers only half of the payload intact (scripts/a2a_test.py; 128 MiB is
    fine, all_gather is fine at 6.4 GB).  So: the part a rank keeps never goes through
    the collective, peers are served in rounds of at most `chunk` elements per pair,
    and a per-pair checksum (sum modulo 2^64) travels separately and is verified.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gloo = dist.get_backend(group) == "gloo"
    dev = inp.device
    in_off = [0] * (world + 1)
    out_off = [0] * (world + 1)
    for p in range(world):
        in_off[p + 1] = in_off[p] + in_sizes[p]
        out_off[p + 1] = out_off[p] + out_sizes[p]
    out = torch.empty((out_off[world],), dtype=inp.dtype, device=dev)
    out[out_off[rank]:out_off[rank + 1]] = inp[in_off[rank]:in_off[rank + 1]]
    peers = [p for p in range(world) if p != rank]
    most = max([max(in_sizes[p], out_sizes[p]) for p in peers], default=0)
    r = torch.tensor([(most + chunk - 1) // chunk], dtype=torch.int64, device="cpu" if gloo else dev)
    if world > 1:
        dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
    for t in range(int(r.item())):
        ss = [0 if p == rank else max(0, min(chunk, in_sizes[p] - t * chunk)) for p in range(world)]
        rs = [0 if p == rank else max(0, min(chunk, out_sizes[p] - t * chunk)) for p in range(world)]
        send = torch.cat([inp[in_off[p] + t * chunk: in_off[p] + t * chunk + ss[p]] for p in range(world)])
        recv = torch.empty((sum(rs),), dtype=inp.dtype, device="cpu" if gloo else dev)
        dist.all_to_all_single(recv, send.cpu() if gloo else send, output_split_sizes=rs, input_split_sizes=ss,
                               group=group)
        at = 0
        for p in range(world):
            if rs[p]:
                out[out_off[p] + t * chunk: out_off[p] + t * chunk + rs[p]] = recv[at:at + rs[p]].to(dev)
                at += rs[p]
    # integrity: checksums of what was meant for each peer vs what arrived from each peer
    mine = torch.stack([inp[in_off[p]:in_off[p + 1]].sum() for p in range(world)]).to(torch.int64)
    theirs = torch.empty_like(mine)
    if gloo:
        t_cpu = torch.empty((world,), dtype=torch.int64)
        dist.all_to_all_single(t_cpu, mine.cpu(), group=group)
        theirs = t_cpu.to(dev)
    else:
        dist.all_to_all_single(theirs, mine, group=group)
    got = torch.stack([out[out_off[p]:out_off[p + 1]].sum() for p in range(world)]).to(torch.int64)
    bad = [p for p in range(world) if int(got[p]) != int(theirs[p])]
    ok = torch.tensor([0 if bad else 1], dtype=torch.int64, device="cpu" if gloo else dev)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)  # every rank fails together, nobody hangs
    if int(ok.item()) == 0:
        raise RuntimeError("all-to-all payload corrupted in transit (rank %d: checksum mismatch from ranks %s)"
                           % (rank, bad))
    return out


def exchange_segments(kmers, counts, seg_counts, group=None):
    """All-to-all of owner-grouped table entries.

    kmers  [total, wk] int64 (uint64 bit patterns), rows grouped by destination rank
    counts [total] int64
    seg_counts [world] int64 -- rows destined to each rank
    Returns (recv_kmers, recv_counts) with everything this rank owns.
    """
    world = dist.get_world_size(group)
    wk = kmers.shape[1]
    gloo = dist.get_backend(group) == "gloo"
    send_sizes = seg_counts.to(torch.int64).contiguous()
    recv_sizes = torch.empty_like(send_sizes)
    if gloo and send_sizes.is_cuda:
        r_cpu = torch.empty((world,), dtype=torch.int64)
        dist.all_to_all_single(r_cpu, send_sizes.cpu(), group=group)
        recv_sizes = r_cpu
    else:
        dist.all_to_all_single(recv_sizes, send_sizes, group=group)
    ss = [int(x) for x in send_sizes.tolist()]
    rs = [int(x) for x in recv_sizes.tolist()]
    assert len(ss) == world and sum(ss) == kmers.shape[0]
    recv_k = exchange_rows(kmers.contiguous().view(-1), [x * wk for x in ss], [x * wk for x in rs], group).view(-1, wk)
    recv_c = exchange_rows(counts.contiguous(), ss, rs, group)
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
    recv_k, recv_c = exchange_segments(kmers[:n], counts[:n], seg, group)
    torch.cuda.synchronize(dev)
    hmap.clear()
    if recv_k.shape[0]:
        _check(hmap._lib.tsx_hip_add_kmers_device(hmap.handle, ctypes.c_void_p(recv_k.data_ptr()),
                                                  ctypes.c_void_p(recv_c.data_ptr()), recv_k.shape[0], None))
    hmap.sync()
    return int(recv_k.shape[0])


class ShardedCounter:
    """Multi-GPU counting into ONE table sharded by slot range (k <= 32).

    Rank r holds home slots [r << l, (r+1) << l) of a table with 2^(l + log2 world)
    slots (TSXHashMapHIP(..., shard_bits=log2 world, shard_index=r)).  Every rank
    scans its own reads; the hashed keys travel to their owners through ONE
    all-to-all BEFORE they are built into a table, so nothing is inserted twice and
    no table is torn down and re-inserted (merge_tables() does that).  Hot k-mers
    that the scan merged on chip travel as a small (key, count) list through an
    all-gather.  After step() rank r answers getKmerCount for the k-mers it owns.
    """

    HOT_CAP = 1 << 20

    def __init__(self, hmap, max_text_bytes, group=None):
        from . import _check
        self.m, self.group = hmap, group
        self.world = dist.get_world_size(group)
        assert self.world == 1 << hmap.layout.shard_bits, "world size must equal 2^shard_bits"
        self.dev = torch.device("cuda", hmap.device)
        cap = ctypes.c_size_t(0)
        _check(hmap._lib.tsx_hip_shard_send_capacity(hmap.handle, max_text_bytes, ctypes.byref(cap)))
        self.send = torch.empty((cap.value,), dtype=torch.int64, device=self.dev)
        self.counts = torch.zeros((self.world,), dtype=torch.int64, device=self.dev)
        self.hot_k = torch.zeros((self.HOT_CAP,), dtype=torch.int64, device=self.dev)
        self.hot_c = torch.zeros((self.HOT_CAP,), dtype=torch.int64, device=self.dev)
        self.hot_n = torch.zeros((1,), dtype=torch.int64, device=self.dev)
        self.recv = torch.empty((0,), dtype=torch.int64, device=self.dev)
        self.gloo = dist.get_backend(group) == "gloo"

    def _a2a(self, out, inp, out_sizes=None, in_sizes=None):
        if self.gloo:  # CPU collective (tests: several ranks sharing one GPU)
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_sizes, input_split_sizes=in_sizes,
                                   group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, output_split_sizes=out_sizes, input_split_sizes=in_sizes,
                                   group=self.group)

    def step(self, text_ptr, nbytes):
        """Count one FASTQ text (device pointer) of this rank's reads into the sharded table."""
        from . import _check
        m, L = self.m, self.m._lib
        vp = ctypes.c_void_p
        _check(L.tsx_hip_shard_scan_device(m.handle, vp(text_ptr), nbytes, vp(self.send.data_ptr()),
                                           self.send.numel(), vp(self.counts.data_ptr()), vp(self.hot_k.data_ptr()),
                                           vp(self.hot_c.data_ptr()), self.HOT_CAP, vp(self.hot_n.data_ptr()), None))
        torch.cuda.synchronize(self.dev)
        ss = [int(x) for x in self.counts.tolist()]
        recv_sizes = torch.empty_like(self.counts)
        self._a2a(recv_sizes, self.counts)
        rs = [int(x) for x in recv_sizes.tolist()]
        n_recv = sum(rs)
        self.recv = exchange_rows(self.send[:sum(ss)], ss, rs, self.group)
        # hot (key, count) lists: pad to the longest, gather everywhere, owners pick theirs
        nh = torch.tensor([min(int(self.hot_n.item()), self.HOT_CAP)], dtype=torch.int64,
                          device="cpu" if self.gloo else self.dev)
        dist.all_reduce(nh, op=dist.ReduceOp.MAX, group=self.group)
        nh = int(nh.item())
        hk = hc = None
        if nh:
            mine = min(int(self.hot_n.item()), self.HOT_CAP)
            self.hot_c[mine:nh].zero_()   # entries past this rank's own list carry count 0 = ignored
            src_k, src_c = self.hot_k[:nh], self.hot_c[:nh]
            if self.gloo:
                gk = [torch.empty((nh,), dtype=torch.int64) for _ in range(self.world)]
                gc = [torch.empty((nh,), dtype=torch.int64) for _ in range(self.world)]
                dist.all_gather(gk, src_k.cpu(), group=self.group)
                dist.all_gather(gc, src_c.cpu(), group=self.group)
                hk, hc = torch.cat(gk).to(self.dev), torch.cat(gc).to(self.dev)
            else:
                hk = torch.empty((nh * self.world,), dtype=torch.int64, device=self.dev)
                hc = torch.empty((nh * self.world,), dtype=torch.int64, device=self.dev)
                dist.all_gather_into_tensor(hk, src_k.contiguous(), group=self.group)
                dist.all_gather_into_tensor(hc, src_c.contiguous(), group=self.group)
        torch.cuda.synchronize(self.dev)
        _check(L.tsx_hip_shard_build_device(m.handle, vp(self.recv.data_ptr()), n_recv, None))
        if nh:
            _check(L.tsx_hip_add_hashed_device(m.handle, vp(hk.data_ptr()), vp(hc.data_ptr()), hk.numel(), None))
        m.sync()
        return n_recv
