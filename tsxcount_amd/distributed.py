"""Multi-GPU plumbing for --mode=HIP: one process per GPU, reads shard across ranks.

Two ways to combine the ranks' work:

MinimizerCounter one whole table per rank, owner of a k-mer = f(its minimizer): strip descriptions
                 masked per owner travel (about 1.8 B per k-mer occurrence), every rank walks only what
                 it owns.  Any world size <= 16, 20 <= k <= 32.  The better exchange from 4 ranks on.
ShardedCounter   ONE table sharded by slot range.  Every rank scans its own reads in
                 windows; the hashed keys of a window travel to the rank that owns their
                 slot range through ONE all-to-all (RCCL over xGMI) and are built there.
                 The exchange of window i runs on its own stream while the GPU scans
                 window i+1 and builds window i-1; the keys a rank owns itself never
                 enter the collective (the scan writes them straight into the receive
                 buffer).  No table is extracted or re-inserted.
merge_tables     the literal "merge of per-GPU tables": every rank counts into its own
                 table, entries are grouped by owner rank, exchanged and re-inserted.
                 Any k, any world size; one extraction + one atomic insert per entry.

torch.distributed is used for the collectives only (backend "nccl" = RCCL on device
tensors; "gloo" stages through host memory and exists for tests that put several ranks
on one GPU).  A rank whose local step fails does not leave its peers inside a
collective: the status travels with the sizes, and every rank raises together.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def shard_reads(n_reads_total, rank, world):
    """Contiguous read shard [first, first+count) of rank `rank`."""
    base, rem = divmod(n_reads_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class TorchComm:
    """The three collectives the multi-GPU path needs, over a torch.distributed group."""

    def __init__(self, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.gloo = dist.get_backend(group) == "gloo"

    def all_to_all(self, out, inp, out_sizes=None, in_sizes=None):
        """all_to_all_single on the current stream; sizes count rows of dim 0."""
        if self.gloo:   # CPU collective: tests with several ranks on one GPU
            torch.cuda.current_stream(out.device).synchronize() if out.is_cuda else None
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_sizes, input_split_sizes=in_sizes,
                                   group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, output_split_sizes=out_sizes, input_split_sizes=in_sizes,
                                   group=self.group)

    def all_to_all_lists(self, outs, inps):
        """outs[p] <- what rank p has in its inps[my rank]: the pieces need not be neighbours in memory (RCCL: grouped
        send / recv straight from and into them, no packing on either side)."""
        if self.gloo:
            dev = next((t.device for t in outs if t.is_cuda), None)
            torch.cuda.current_stream(dev).synchronize() if dev is not None else None
            flat = torch.cat([t.reshape(-1).cpu() for t in inps])
            o = torch.empty((sum(t.numel() for t in outs),), dtype=flat.dtype)
            dist.all_to_all_single(o, flat, output_split_sizes=[t.numel() for t in outs],
                                   input_split_sizes=[t.numel() for t in inps], group=self.group)
            at = 0
            for t in outs:
                t.copy_(o[at:at + t.numel()].view(t.shape))
                at += t.numel()
        else:
            # (a pair with nothing to exchange -- both sides know: the sizes travelled before -- trades a dummy element
            # instead of an empty tensor)
            dev = next((t.device for t in list(outs) + list(inps) if t.is_cuda), None)
            dummy = lambda t: torch.zeros((1,), dtype=t.dtype, device=dev if dev is not None else t.device)
            dist.all_to_all([t if t.numel() else dummy(t) for t in outs], [t if t.numel() else dummy(t) for t in inps],
                            group=self.group)

    def all_gather(self, out, inp):
        if self.gloo:
            torch.cuda.current_stream(out.device).synchronize() if out.is_cuda else None
            parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(self.world)]
            dist.all_gather(parts, inp.cpu(), group=self.group)
            out.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)

    def all_reduce(self, t, op="sum"):
        """In place; returns t.  Small tensors only."""
        rop = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op]
        if self.gloo and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, op=rop, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=rop, group=self.group)
        return t


def _comm(group_or_comm):
    return group_or_comm if hasattr(group_or_comm, "all_to_all") else TorchComm(group_or_comm)


def agree(ok, comm, device):
    """True iff `ok` holds on every rank (one tiny MIN all-reduce): call it before a collective
    whose peers would otherwise wait for a rank that has already failed."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cpu" if comm.gloo else device)
    comm.all_reduce(t, "min")
    return bool(int(t.item()))


def agreed_max(value, comm, device):
    """The largest `value` of any rank (one tiny MAX all-reduce): what every rank must derive its window geometry from."""
    if comm.world == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if comm.gloo else device)
    comm.all_reduce(t, "max")
    return int(t.item())


def window_geometry(max_text_bytes, windows=None, min_window=32 << 20):
    """(windows, win_bytes) of a step over texts of at most max_text_bytes: at most 4 windows of at least min_window
    bytes (TSX_HIP_SHARD_WINDOWS overrides the count), window length a multiple of 4 KiB."""
    if windows is None:
        windows = int(os.environ.get("TSX_HIP_SHARD_WINDOWS", "0")) or max(1, min(4, max_text_bytes // min_window))
    windows = max(1, int(windows))
    return windows, max(4096, ((max_text_bytes + windows - 1) // windows + 4095) & ~4095)


def window_of(i, nbytes, win_bytes):
    """(offset, length) of window i of a text of nbytes: windows past the end of a short (or empty) text are empty, their
    offset the end of the text rounded down to the 16 bytes the entry points ask for."""
    off = i * win_bytes
    if off >= nbytes:
        return nbytes & ~15, 0
    return off, min(win_bytes, nbytes - off)


A2A_CHUNK = 128 << 20  # elements (1 GiB of int64) per pair and collective


def exchange_rows(inp, in_sizes, out_sizes, group=None, chunk=A2A_CHUNK):
    """All-to-all of 1-D int64 data grouped by destination rank (in_sizes[p] elements for rank p);
    the result is grouped by source rank (out_sizes[p] elements from rank p).

    When every pair's message fits `chunk` elements (the usual case) it is ONE all_to_all_single
    straight from `inp` into the result; larger messages go in rounds (the part a rank keeps is then
    copied on the device).  A per-pair checksum (sum modulo 2^64) travels separately and is
    verified: a single all_to_all_single of >= 1.2 GB was seen to deliver half of its payload on
    this stack with ONE rank (scripts/a2a_test.py); whether that also holds between real peers
    is logged by the checksum here rather than assumed."""
    comm = _comm(group)
    world, rank = comm.world, comm.rank
    dev = inp.device
    in_off = [0] * (world + 1)
    out_off = [0] * (world + 1)
    for p in range(world):
        in_off[p + 1] = in_off[p] + in_sizes[p]
        out_off[p + 1] = out_off[p] + out_sizes[p]
    out = torch.empty((out_off[world],), dtype=inp.dtype, device=dev)
    most = max([max(in_sizes[p], out_sizes[p]) for p in range(world)], default=0)
    r = torch.tensor([(most + chunk - 1) // chunk], dtype=torch.int64, device="cpu" if comm.gloo else dev)
    if world > 1:
        comm.all_reduce(r, "max")
    rounds = int(r.item())
    if rounds <= 1:
        comm.all_to_all(out, inp, list(out_sizes), list(in_sizes))   # no staging copy on either side
    else:
        out[out_off[rank]:out_off[rank + 1]] = inp[in_off[rank]:in_off[rank + 1]]
        for t in range(rounds):
            ss = [0 if p == rank else max(0, min(chunk, in_sizes[p] - t * chunk)) for p in range(world)]
            rs = [0 if p == rank else max(0, min(chunk, out_sizes[p] - t * chunk)) for p in range(world)]
            send = torch.cat([inp[in_off[p] + t * chunk: in_off[p] + t * chunk + ss[p]] for p in range(world)])
            recv = torch.empty((sum(rs),), dtype=inp.dtype, device=dev)
            comm.all_to_all(recv, send, rs, ss)
            at = 0
            for p in range(world):
                if rs[p]:
                    out[out_off[p] + t * chunk: out_off[p] + t * chunk + rs[p]] = recv[at:at + rs[p]]
                    at += rs[p]
    # integrity: checksums of what was meant for each peer vs what arrived from each peer
    mine = torch.stack([inp[in_off[p]:in_off[p + 1]].sum() for p in range(world)]).to(torch.int64)
    theirs = torch.empty_like(mine)
    comm.all_to_all(theirs, mine)
    got = torch.stack([out[out_off[p]:out_off[p + 1]].sum() for p in range(world)]).to(torch.int64)
    bad = [p for p in range(world) if int(got[p]) != int(theirs[p])]
    if not agree(not bad, comm, dev):   # every rank fails together, nobody hangs
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
    comm = _comm(group)
    world = comm.world
    wk = kmers.shape[1]
    send_sizes = seg_counts.to(torch.int64).contiguous()
    recv_sizes = torch.empty_like(send_sizes)
    comm.all_to_all(recv_sizes, send_sizes)
    ss = [int(x) for x in send_sizes.tolist()]
    rs = [int(x) for x in recv_sizes.tolist()]
    assert len(ss) == world and sum(ss) == kmers.shape[0]
    recv_k = exchange_rows(kmers.contiguous().view(-1), [x * wk for x in ss], [x * wk for x in rs], comm).view(-1, wk)
    recv_c = exchange_rows(counts.contiguous(), ss, rs, comm)
    return recv_k, recv_c


def merge_tables(hmap, group=None):
    """Merge the per-GPU tables in place (see module docstring).  Returns the
    number of entries this rank received."""
    from . import OK, TSXException, _check
    comm = _comm(group)
    world = comm.world
    dev = torch.device("cuda", hmap.device)
    n = hmap.stats()["distinct"]
    kmers = torch.empty((max(n, 1), hmap.wk), dtype=torch.int64, device=dev)
    counts = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    seg = torch.zeros((world,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    rc = hmap._lib.tsx_hip_partition_device(hmap.handle, world, ctypes.c_void_p(kmers.data_ptr()),
                                            ctypes.c_void_p(counts.data_ptr()), max(n, 1),
                                            ctypes.c_void_p(seg.data_ptr()), None)
    if not agree(rc == OK, comm, dev):
        _check(rc)
        raise TSXException(rc, "merge_tables: another rank failed to group its table by owner")
    recv_k, recv_c = exchange_segments(kmers[:n], counts[:n], seg, comm)
    torch.cuda.synchronize(dev)
    hmap.clear()
    rc = OK
    if recv_k.shape[0]:
        rc = hmap._lib.tsx_hip_add_kmers_device(hmap.handle, ctypes.c_void_p(recv_k.data_ptr()),
                                                ctypes.c_void_p(recv_c.data_ptr()), recv_k.shape[0], None)
    if rc == OK:
        rc = hmap._lib.tsx_hip_sync(hmap.handle)
    if not agree(rc == OK, comm, dev):
        _check(rc)
        raise TSXException(rc, "merge_tables: another rank failed to insert what it received")
    return int(recv_k.shape[0])


class ShardedCounter:
    """Multi-GPU counting into ONE table sharded by slot range (k <= 32).

    Rank r holds home slots [r << l, (r+1) << l) of a table with 2^(l + log2 world)
    slots (TSXHashMapHIP(..., shard_bits=log2 world, shard_index=r)).  step() counts one
    device text of this rank's reads, cut into W windows:

        compute stream   scan(0) scan(1) l1(0) scan(2) l1(1) ... l1(W-1)  build
        exchange stream        sizes(0) a2a(0)  sizes(1) a2a(1) ... a2a(W-1)

    scan(i)   tsx_hip_shard_scan_window_device: window i of the text -> keys grouped by owner
              (own keys straight into the receive buffer), per-owner counts, hot (key, count) list
    sizes(i)  ONE small all-to-all carries, per pair: keys to come, this rank's status, the
              length of its hot list.  Its result is the only thing the host waits for, and it
              waits while scan(i+1) is already queued on the GPU
    a2a(i)    ONE all_to_all_single with split sizes, from the send buffer into window i's part of
              the receive buffer, behind the own keys; hot lists by all-gather (owners pick theirs)
    l1(i)     tsx_hip_shard_l1_window_device: radix level 1 over window i's received keys as soon as they
              are here (compute stream, behind scan(i+1)); tables that need one level only: not at all
    build     ONE tsx_hip_shard_build_l1_device (level 2 + build) over all windows' keys (a build costs a
              pass over the whole slot range, however few keys it brings), then the hot lists

    The exchange of window i overlaps the scan of window i+1; two send and hot buffers alternate.
    Integrity: every scan adds the sum of the keys it wrote, the build the sum of the keys it
    read; the two totals must agree over all ranks (one all-reduce per step).  After step() rank r
    answers getKmerCount for the k-mers it owns.
    """

    HOT_CAP = 1 << 18
    MIN_WINDOW = 32 << 20      # bytes of text below which a step is not split further

    def __init__(self, hmap, max_text_bytes, group=None, windows=None):
        self.m = hmap
        self.comm = _comm(group)
        self.world, self.rank = self.comm.world, self.comm.rank
        assert self.world == 1 << hmap.layout.shard_bits, "world size must equal 2^shard_bits"
        self.dev = torch.device("cuda", hmap.device)
        # Every rank must run the SAME number of windows (each window is a round of collectives) over the same
        # window length, whatever its own text size: both are derived from the largest text of any rank.
        self.max_text_bytes = agreed_max(max_text_bytes, self.comm, self.dev)
        self.windows, self.win_bytes = window_geometry(self.max_text_bytes, windows, self.MIN_WINDOW)
        i64 = dict(dtype=torch.int64, device=self.dev)
        self.send_cap = 0                         # key-exchange buffers: allocated by the first keys-mode step
        self.sums = torch.zeros((2,), **i64)      # [0] += keys written by the scans, [1] += keys read by the build
        self.cs = torch.cuda.Stream(self.dev)     # every kernel of a step
        self.xs = torch.cuda.Stream(self.dev)     # the collectives
        self.ev_scan = [torch.cuda.Event() for _ in range(2)]
        self.ev_exch = [torch.cuda.Event() for _ in range(2)]
        self.last = {}

    def _window(self, i, nbytes):
        """(offset, length) of window i of a text of nbytes: windows past the end of a short (or empty) text are
        empty, their offset the end of the text rounded down to the 16 bytes the entry points ask for."""
        return window_of(i, nbytes, self.win_bytes)

    def _ensure_key_buffers(self):
        """Send / receive / hot-list buffers of the key exchange, sized by the map's CURRENT record format (a FASTA
        text holds up to twice the k-mers per byte): allocated on the first keys-mode step, again when the
        capacity the library asks for has grown."""
        from . import _check
        cap = ctypes.c_size_t(0)
        _check(self.m._lib.tsx_hip_shard_send_capacity(self.m.handle, self.win_bytes + 256, ctypes.byref(cap)))
        if cap.value <= self.send_cap:
            return
        i64 = dict(dtype=torch.int64, device=self.dev)
        self.send_cap = cap.value
        self.send = [torch.empty((self.send_cap,), **i64) for _ in range(2)]
        # receive buffer: one part per window -- own keys in front (any number up to the window's total),
        # the peers' behind them
        self.part = 2 * self.send_cap
        self.recv = torch.empty((self.windows * self.part,), **i64)
        self.counts = [torch.zeros((self.world,), **i64) for _ in range(2)]
        self.hot_k = [torch.zeros((self.HOT_CAP,), **i64) for _ in range(2)]
        self.hot_c = [torch.zeros((self.HOT_CAP,), **i64) for _ in range(2)]
        self.hot_n = [torch.zeros((1,), **i64) for _ in range(2)]
        self.hot_all_k = [torch.zeros((0,), **i64) for _ in range(self.windows)]
        self.hot_all_c = [torch.zeros((0,), **i64) for _ in range(self.windows)]

    def _scan(self, i, text_ptr, nbytes):
        m, L, vp, b = self.m, self.m._lib, ctypes.c_void_p, i & 1
        off, ln = self._window(i, nbytes)
        own_ptr = self.recv.data_ptr() + i * self.part * 8
        rc = L.tsx_hip_shard_scan_window_device(
            m.handle, vp(text_ptr), nbytes, off, ln, vp(self.send[b].data_ptr()), self.send_cap,
            vp(own_ptr), self.part, vp(self.counts[b].data_ptr()),
            vp(self.hot_k[b].data_ptr()), vp(self.hot_c[b].data_ptr()), self.HOT_CAP, vp(self.hot_n[b].data_ptr()),
            vp(self.sums.data_ptr()), vp(self.cs.cuda_stream))
        self.ev_scan[b].record(self.cs)
        return rc

    def _mode(self):
        """"keys": every key travels to its owner (8 B per k-mer occurrence, (N-1)/N of them);  "desc": strip
        descriptions (32 B per up-to-64 occurrences) are all-gathered and every GPU walks all of them, keeping what it
        owns: N x the rolling work (about 1.4 ms per 1e9 positions) for N/16 of the bytes.  At N = 8 that is 3.9 GB
        received instead of 5.6 GB sent and a step bound by the walks (about 23 ms) rather than by the links; at N = 2
        a sixth of the bytes over the one link there is.  Default: "desc" wherever the table is split by two radix
        levels and the keys fit one limb (TSX_HIP_SHARD_MODE overrides)."""
        want = os.environ.get("TSX_HIP_SHARD_MODE", "auto")
        ok = bool(self.m._lib.tsx_hip_shard_l1_supported(self.m.handle))
        if want == "keys" or not ok:
            return "keys"
        if want == "desc":
            return "desc"
        return "desc"

    def _step_desc(self, text_ptr, nbytes):
        """step() by description exchange.

            compute stream   desc(0) desc(1) walk(0,*) desc(2) walk(1,*) ...   level 2 + build
            exchange stream        gather(0)       gather(1)      ...
        """
        from . import OK, TSXException, _check
        m, L, vp = self.m, self.m._lib, ctypes.c_void_p
        world, comm = self.world, self.comm
        if nbytes > self.max_text_bytes:
            raise ValueError("sharded step: text of %d bytes, the counter was made for %d" % (nbytes, self.max_text_bytes))
        nwin = self.windows      # the same on every rank: a window is a round of collectives (short texts: empty windows)
        nslots = nwin            # one walk launch (= one set of level-1 lists) per window, over the descriptions of ALL GPUs
        i64 = dict(dtype=torch.int64, device=self.dev)
        # long descriptions (four strips in 32 bytes: half the bytes per start position) unless TSX_HIP_SHARD_LONG=0
        lng = 0 if os.environ.get("TSX_HIP_SHARD_LONG", "1") == "0" else 1
        w8 = 4 if lng else 2      # int64 words per description
        if getattr(self, "dsc_long", None) != lng:
            cap = ctypes.c_size_t(0)
            _check(L.tsx_hip_shard_desc_capacity(m.handle, self.win_bytes + 256, lng, ctypes.byref(cap)))
            self.dsc_cap, self.dsc_long = cap.value, lng
            self.dsc = [torch.empty((w8 * self.dsc_cap,), **i64) for _ in range(2)]
            self.dsc_n = [torch.zeros((1,), **i64) for _ in range(2)]
            self.dsc_all = [torch.zeros((0,), **i64) for _ in range(self.windows)]
            self.emit = torch.zeros((2,), **i64)     # [0] += k-mer occurrences described, [1] += occurrences kept by the walks
        self.cs.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.cs):
            self.emit.zero_()
        late = None      # a failure only this rank has seen: it stays in the collectives, all raise after the last all-reduce

        def desc(i):
            b = i & 1
            off, ln = self._window(i, nbytes)
            rc = L.tsx_hip_shard_desc_window_device(m.handle, vp(text_ptr), nbytes, off, ln, lng,
                                                    vp(self.dsc[b].data_ptr()), self.dsc_cap, vp(self.dsc_n[b].data_ptr()),
                                                    vp(self.emit.data_ptr()), vp(self.cs.cuda_stream))
            self.ev_scan[b].record(self.cs)
            return rc

        rc_next = desc(0)
        est_total, total_desc = 0, 0
        for i in range(nwin):
            b = i & 1
            rc_this = rc_next
            if i + 1 < nwin:
                if i >= 1:
                    self.cs.wait_event(self.ev_exch[(i + 1) & 1])   # gather i-1 has read the buffer desc(i+1) writes
                rc_next = desc(i + 1)                               # queued before the host waits for window i
            with torch.cuda.stream(self.xs):
                self.xs.wait_event(self.ev_scan[b])
                mine = torch.stack([self.dsc_n[b][0], torch.tensor(int(rc_this), **i64)])
                allm = torch.empty((2 * world,), **i64)
                comm.all_gather(allm, mine)
                host = allm.cpu().view(world, 2)                    # the one host wait of the window
                counts = [int(x) for x in host[:, 0].tolist()]
                status = [int(x) for x in host[:, 1].tolist()]
                if any(s != OK for s in status):                    # every rank sees the same statuses: all leave together
                    late = late or (i, status)
                    break
                nmax = max(max(counts), 1)
                if nmax > self.dsc_cap:                             # TSX_HIP_ERANGE; the same counts on every rank
                    late = late or (i, [-7])
                    break
                if self.dsc_all[i].numel() < w8 * nmax * world:
                    self.dsc_all[i] = torch.empty((w8 * nmax * world,), **i64)
                # every GPU sends nmax descriptions: what lies behind its own count is zeroed (validity bits 0 = a strip
                # without k-mer starts), so that the receivers can walk the gathered array as ONE packed run
                if counts[self.rank] < nmax:
                    self.dsc[b][w8 * counts[self.rank]:w8 * nmax].zero_()
                comm.all_gather(self.dsc_all[i][:w8 * nmax * world], self.dsc[b][:w8 * nmax])
                self.ev_exch[b].record(self.xs)
            total_desc += sum(counts)
            self.cs.wait_event(self.ev_exch[b])
            if i == 0:
                est_total = int(sum(counts) * (64 if lng else 16) * nwin / world * 1.1) + 65536
            if late is None:
                # a GPU that keeps one key in >= 4: owner-filtered walk into key logs + level 1 (two kernels, no rings to
                # hold while it mostly rolls); else the fused walk (TSX_HIP_SHARD_FILTER=0|1 overrides)
                flt = os.environ.get("TSX_HIP_SHARD_FILTER", "auto")
                walk = L.tsx_hip_shard_filter_device if (flt == "1" or (flt == "auto" and world >= 4)) else L.tsx_hip_shard_walk_device
                rc = walk(m.handle, vp(self.dsc_all[i].data_ptr()), nmax * world, lng, i, nslots,
                          est_total, vp(self.emit[1:].data_ptr()), vp(self.cs.cuda_stream))
                if rc != OK:
                    late = (i, [rc])
        self.cs.wait_stream(self.xs)
        failure = late
        if failure is None and total_desc:
            rc = L.tsx_hip_shard_build_l1_device(m.handle, vp(self.cs.cuda_stream))
            if rc != OK:
                failure = (nwin, [rc])
        self.cs.synchronize()
        self.xs.synchronize()
        em = [int(x) for x in self.emit.tolist()]
        fin = torch.tensor([em[0] - em[1], 0 if failure is None else 1], dtype=torch.int64,
                           device="cpu" if comm.gloo else self.dev)
        comm.all_reduce(fin, "sum") if world > 1 else None
        diff, nfail = int(fin[0].item()), int(fin[1].item())
        self.last = {"windows": nwin, "received_keys": em[1], "key_sum_diff": diff, "mode": "desc"}
        if nfail:
            if failure is not None:
                bad = [s for s in failure[1] if s != OK]
                _check(bad[0])
            raise TSXException(-4, "sharded step: another rank failed")
        if diff != 0:
            raise RuntimeError("sharded step: k-mer occurrences scanned - kept, over all ranks = %d" % diff)
        m.sync()
        return em[1]

    def step(self, text_ptr, nbytes):
        """Count one FASTQ text (device pointer, 16-byte aligned) of this rank's reads into the sharded table."""
        from . import OK, TSXException, _check
        if self._mode() == "desc":
            return self._step_desc(text_ptr, nbytes)
        m, L, vp = self.m, self.m._lib, ctypes.c_void_p
        world, rank, comm = self.world, self.rank, self.comm
        if nbytes > self.max_text_bytes:
            raise ValueError("sharded step: text of %d bytes, the counter was made for %d" % (nbytes, self.max_text_bytes))
        nwin = self.windows      # the same on every rank (see _step_desc)
        self._ensure_key_buffers()
        # the caller's text may have been produced on another stream
        self.cs.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.cs):
            self.sums.zero_()
        failure = None     # seen by every rank in the same window (statuses travel with the sizes): all leave together
        late = None        # seen by this rank only: it stays in the collectives, everyone raises after the last all-reduce
        piece_off, piece_cnt, hot_g = [], [], []
        l1_windows = bool(L.tsx_hip_shard_l1_supported(m.handle)) and os.environ.get("TSX_HIP_SHARD_L1_WINDOWS", "1") != "0"
        est_total = 0
        rc_scan = self._scan(0, text_ptr, nbytes)
        for i in range(nwin):
            b = i & 1
            rc_this = rc_scan
            if i + 1 < nwin:
                if i >= 1:
                    self.cs.wait_event(self.ev_exch[(i + 1) & 1])   # exchange i-1 has finished with the buffers scan i+1 reuses
                rc_scan = self._scan(i + 1, text_ptr, nbytes)       # queued before the host waits for window i
            with torch.cuda.stream(self.xs):
                self.xs.wait_event(self.ev_scan[b])
                # sizes(i): row p = [keys for p, my status, my hot keys]
                meta_in = torch.empty((world, 3), dtype=torch.int64, device=self.dev)
                meta_in[:, 0] = self.counts[b]
                meta_in[:, 1] = int(rc_this)
                meta_in[:, 2] = self.hot_n[b]
                meta_out = torch.empty_like(meta_in)
                comm.all_to_all(meta_out, meta_in)
                host = torch.cat([meta_out.view(-1), self.counts[b], self.hot_n[b]]).cpu()   # the one host wait
                mo = host[:3 * world].view(world, 3)
                mine = [int(x) for x in host[3 * world:4 * world].tolist()]
                my_hot = min(int(host[4 * world]), self.HOT_CAP)
                status = [int(x) for x in mo[:, 1].tolist()]
                if any(s != OK for s in status):
                    failure = (i, status)
                    break
                own = mine[rank]
                ss = [0 if p == rank else mine[p] for p in range(world)]
                rs = [0 if p == rank else int(mo[p, 0]) for p in range(world)]
                n_recv = own + sum(rs)
                part = self.recv[i * self.part:(i + 1) * self.part]
                if n_recv > self.part:
                    # TSX_HIP_ERANGE: ownership far more skewed than the receive buffer allows.  A LOCAL failure: the
                    # peers are about to enter the collective, so this rank takes part in it (into a scratch buffer)
                    # and in every later one, and all ranks raise together after the step's last all-reduce.
                    late = late or (i, [-7])
                    if world > 1:
                        comm.all_to_all(torch.empty((sum(rs),), dtype=torch.int64, device=self.dev),
                                        self.send[b][:sum(ss)], rs, ss)
                    n_recv = 0
                elif world > 1:
                    comm.all_to_all(part[own:n_recv], self.send[b][:sum(ss)], rs, ss)
                # hot lists: padded to the longest, gathered everywhere, owners pick theirs
                nh = min(max(max(int(x) for x in mo[:, 2].tolist()), 0), self.HOT_CAP)
                g = 0
                if nh:
                    g = 1 << (nh - 1).bit_length()
                    self.hot_c[b][my_hot:g].zero_()   # entries past this rank's own list carry count 0 = ignored
                    if self.hot_all_k[i].numel() < g * world:
                        self.hot_all_k[i] = torch.empty((g * world,), dtype=torch.int64, device=self.dev)
                        self.hot_all_c[i] = torch.empty((g * world,), dtype=torch.int64, device=self.dev)
                    comm.all_gather(self.hot_all_k[i][:g * world], self.hot_k[b][:g])
                    comm.all_gather(self.hot_all_c[i][:g * world], self.hot_c[b][:g])
                self.ev_exch[b].record(self.xs)
            piece_off.append(i * self.part)
            piece_cnt.append(n_recv if (failure is None and late is None) else 0)
            hot_g.append(g)
            if l1_windows and failure is None and late is None:
                # level 1 of window i as soon as its keys are here, behind the scan of window i+1 on the compute stream
                self.cs.wait_event(self.ev_exch[b])
                if i == 0:
                    est_total = int(n_recv * nwin * 1.1)
                rc = L.tsx_hip_shard_l1_window_device(m.handle, vp(self.recv.data_ptr() + i * self.part * 8), n_recv, i, nwin,
                                                      est_total, vp(self.sums[1:].data_ptr()), vp(self.cs.cuda_stream))
                if rc != OK:
                    late = (i, [rc])   # local: keep taking part in the later windows' collectives (see above)
        # ---- ONE partition + build over everything this rank owns, then the hot (key, count) lists
        n_recv_total = sum(piece_cnt)
        self.cs.wait_stream(self.xs)
        if failure is None and late is not None:
            failure = late
        if failure is None:
            po = (ctypes.c_uint64 * len(piece_off))(*piece_off)
            pc = (ctypes.c_uint64 * len(piece_cnt))(*piece_cnt)
            if l1_windows:
                rc = L.tsx_hip_shard_build_l1_device(m.handle, vp(self.cs.cuda_stream)) if n_recv_total else OK
            else:
                rc = L.tsx_hip_shard_build_pieces_device(m.handle, vp(self.recv.data_ptr()), po, pc, len(piece_off),
                                                         vp(self.sums[1:].data_ptr()), vp(self.cs.cuda_stream))
            for i, g in enumerate(hot_g):
                if rc == OK and g:
                    rc = L.tsx_hip_add_hashed_device(m.handle, vp(self.hot_all_k[i].data_ptr()),
                                                     vp(self.hot_all_c[i].data_ptr()), g * world, vp(self.cs.cuda_stream))
            if rc != OK:
                failure = (nwin, [rc])
        self.cs.synchronize()
        self.xs.synchronize()
        # integrity of the exchange + agreement on failures: one small all-reduce per step
        fin = torch.stack([self.sums[0] - self.sums[1],
                           torch.tensor(0 if failure is None else 1, dtype=torch.int64, device=self.dev)])
        red = fin.cpu() if comm.gloo else fin
        comm.all_reduce(red, "sum") if world > 1 else None
        diff, nfail = int(red[0].item()), int(red[1].item())
        self.last = {"windows": nwin, "received_keys": n_recv_total, "key_sum_diff": diff}
        if nfail:
            if failure is not None:
                bad = [s for s in failure[1] if s != OK]
                _check(bad[0])
            raise TSXException(-4, "sharded step: another rank failed")
        if diff != 0:
            raise RuntimeError("sharded step: keys lost or corrupted in the exchange (sum of keys scanned - built, "
                               "over all ranks = %d)" % diff)
        m.sync()
        return n_recv_total


class MinimizerCounter:
    """Multi-GPU counting with owner(k-mer) = f(minimizer of the k-mer) (csrc/tsx_minimizer.h; 20 <= k <= 32, world <= 16,
    any world size).  Every rank holds a WHOLE table (TSXHashMapHIP without shard_bits) of the k-mers it owns; nothing is
    merged at the end, and a lookup goes to rank owner_of(kmers).  step() counts one device text of this rank's reads, cut
    into W windows:

        compute stream   describe split(0) .. split(W-1) walk(0) walk(1) ...  walk(W-1)  level 2 + build + homopolymers
        exchange stream           sizes(0) a2a(0) sizes(1) a2a(1) ...

    split(i)  the text is described ONCE (tsx_hip_mini_describe_device: line pass + strip descriptions; texts above 2 GiB in
              pieces); tsx_hip_mini_split_device then hands out share i of the described strips as one packed list per owner
              rank (consecutive k-mers mostly share a minimizer, so a strip of 16 starts ends up in about 1.7 lists);
              homopolymer k-mers (poly-A tails: ONE key, one owner) are counted here instead and leave the descriptions
    sizes(i)  one small all-to-all: per pair {descriptions to come, the sender's status}; the only host wait of the window,
              while the other shares' splits and the earlier walks are queued
    a2a(i)    grouped send / recv straight from the lists into window i's receive buffer (back to back by source rank)
    walk(i)   tsx_hip_shard_walk_device (flag 2): first window + rolls + radix level 1 over what arrived, every key kept
    build     ONE tsx_hip_shard_build_l1_device; the homopolymer totals (summed over ranks and windows in the step's final
              all-reduce) are added on their owners

    Integrity: sum over ranks of (k-mer occurrences described - walked - homopolymers) must be 0."""

    MIN_WINDOW = 32 << 20
    PIECE = 2 << 30          # bytes of text described at once (the entry point takes less than 4 GiB)

    def __init__(self, hmap, max_text_bytes, group=None, windows=None):
        from . import _check
        self.m = hmap
        self.comm = _comm(group)
        self.world, self.rank = self.comm.world, self.comm.rank
        L = hmap._lib
        if hmap.layout.shard_bits != 0 or not L.tsx_hip_mini_supported(hmap.handle):
            raise ValueError("minimizer exchange: 20 <= k <= 32, one-limb slots, a table split by two radix levels, shard_bits = 0")
        if not 1 <= self.world <= 16:
            raise ValueError("minimizer exchange: at most 16 ranks")
        self.dev = torch.device("cuda", hmap.device)
        self.max_text_bytes = agreed_max(max_text_bytes, self.comm, self.dev)
        # a text is described in pieces of at most PIECE bytes (one line pass + one strip_desc_kernel launch each); every piece
        # is split and exchanged in `parts` shares.  Every rank runs pieces x parts rounds of collectives, whatever its own text.
        self.piece_bytes = min(self.PIECE, max(4096, (self.max_text_bytes + 4095) & ~4095))
        self.pieces = max(1, -(-self.max_text_bytes // self.piece_bytes))
        self.parts, _ = window_geometry(self.piece_bytes, windows, self.MIN_WINDOW)
        self.windows = self.pieces * self.parts
        cap = ctypes.c_size_t(0)
        _check(L.tsx_hip_mini_part_capacity(hmap.handle, self.piece_bytes + 256, self.parts, ctypes.byref(cap)))
        self.cap = cap.value
        i64 = dict(dtype=torch.int64, device=self.dev)
        # one set of lists per share of a piece: all shares are split before the first is walked (the exchange of share 0 then
        # runs under the splits of the others, and every later one under the walks in front of it)
        self.dsc = [torch.empty((2 * self.cap * self.world,), **i64) for _ in range(self.parts)]   # [owner][cap] descriptions of 16 bytes
        self.cnt = [torch.zeros((self.world + 4,), **i64) for _ in range(self.parts)]              # list lengths, then homopolymers per base
        self.recv = [torch.zeros((0,), **i64) for _ in range(self.windows)]
        self.emit = torch.zeros((2,), **i64)      # [0] += occurrences described, [1] += occurrences walked
        self.hom = torch.zeros((4,), **i64)
        self.cs = torch.cuda.Stream(self.dev)
        self.xs = torch.cuda.Stream(self.dev)
        self.ev_scan = [torch.cuda.Event() for _ in range(self.parts)]
        self.ev_exch = [torch.cuda.Event() for _ in range(self.parts)]
        from . import encode
        import numpy as np
        self.hom_kmers = np.stack([encode("ACGT"[b] * hmap.k, hmap.k) for b in range(4)])
        self.hom_owner = owner_of(self.hom_kmers, hmap.k, self.world)
        self.hom_dev = torch.from_numpy(self.hom_kmers.astype(np.int64)).to(self.dev)
        self.last = {}

    def step(self, text_ptr, nbytes):
        from . import OK, TSXException, _check
        m, L, vp = self.m, self.m._lib, ctypes.c_void_p
        world, rank, comm = self.world, self.rank, self.comm
        if nbytes > self.max_text_bytes:
            raise ValueError("minimizer step: text of %d bytes, the counter was made for %d" % (nbytes, self.max_text_bytes))
        nwin = self.windows
        i64 = dict(dtype=torch.int64, device=self.dev)
        self.cs.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.cs):
            self.emit.zero_()
            self.hom.zero_()
        late = None      # a failure only this rank has seen: it stays in the collectives, all raise after the last all-reduce

        def split(i):
            piece, part = divmod(i, self.parts)
            b = part
            rc = OK
            if part == 0:     # the piece is described once; its shares follow window by window
                off, ln = window_of(piece, nbytes, self.piece_bytes)
                rc = L.tsx_hip_mini_describe_device(m.handle, vp(text_ptr), nbytes, off, ln, vp(self.emit.data_ptr()),
                                                    vp(self.cs.cuda_stream))
            rc2 = L.tsx_hip_mini_split_device(m.handle, part, self.parts, world, vp(self.dsc[b].data_ptr()), self.cap,
                                              vp(self.cnt[b].data_ptr()), vp(self.cs.cuda_stream))
            rc = rc if rc != OK else rc2
            with torch.cuda.stream(self.cs):
                self.hom += self.cnt[b][world:]
            self.ev_scan[b].record(self.cs)
            return rc

        est_total, n_recv_total, n_sent = 0, 0, 0
        rcs = []
        for i in range(nwin):
            b = i % self.parts
            if b == 0:     # a new piece: described once, all its shares split (queued) before the host waits for the first
                if i:
                    for e in self.ev_exch:
                        self.cs.wait_event(e)       # the last piece's exchanges have read the lists these splits overwrite
                rcs = [split(i + j) for j in range(self.parts)]
            rc_this = rcs[b]
            with torch.cuda.stream(self.xs):
                self.xs.wait_event(self.ev_scan[b])
                meta_in = torch.empty((world, 2), **i64)
                meta_in[:, 0] = self.cnt[b][:world]
                meta_in[:, 1] = int(rc_this)
                meta_out = torch.empty_like(meta_in)
                comm.all_to_all(meta_out, meta_in)
                host = torch.cat([meta_out.view(-1), self.cnt[b][:world]]).cpu()   # the one host wait of the window
                mo = host[:2 * world].view(world, 2)
                mine = [min(int(x), self.cap) for x in host[2 * world:].tolist()]
                status = [int(x) for x in mo[:, 1].tolist()]
                if any(s != OK for s in status):     # every rank sees the same statuses (a rank's own travels to all): all leave together
                    late = late or (i, status)
                    break
                rs = [min(int(x), self.cap) for x in mo[:, 0].tolist()]
                n_recv = sum(rs)
                if self.recv[i].numel() < 2 * n_recv:
                    self.recv[i] = torch.empty((2 * n_recv + 4096,), **i64)
                outs, at = [], 0
                for p in range(world):
                    outs.append(self.recv[i][2 * at:2 * (at + rs[p])])
                    at += rs[p]
                inps = [self.dsc[b][2 * p * self.cap:2 * (p * self.cap + mine[p])] for p in range(world)]
                comm.all_to_all_lists(outs, inps)
                self.ev_exch[b].record(self.xs)
            n_recv_total += n_recv
            n_sent += sum(mine) - mine[rank]
            self.cs.wait_event(self.ev_exch[b])
            if i == 0:
                # k-mer occurrences this rank will own: its share of everybody's text (a description stands for about 7)
                est_total = int(max(n_recv, 4096) * 10 * nwin * 1.2) + 65536
            if late is None:
                rc = L.tsx_hip_shard_walk_device(m.handle, vp(self.recv[i].data_ptr()), n_recv, 2, i, nwin, est_total,
                                                 vp(self.emit[1:].data_ptr()), vp(self.cs.cuda_stream))
                if rc != OK:
                    late = (i, [rc])
        self.cs.wait_stream(self.xs)
        failure = late
        if failure is None and n_recv_total:
            rc = L.tsx_hip_shard_build_l1_device(m.handle, vp(self.cs.cuda_stream))
            if rc != OK:
                failure = (nwin, [rc])
        self.cs.synchronize()
        self.xs.synchronize()
        em = [int(x) for x in self.emit.tolist()]
        hom = [int(x) for x in self.hom.tolist()]
        fin = torch.tensor([em[0] - em[1] - sum(hom), 0 if failure is None else 1] + hom, dtype=torch.int64,
                           device="cpu" if comm.gloo else self.dev)
        comm.all_reduce(fin, "sum") if world > 1 else None
        red = [int(x) for x in fin.tolist()]
        diff, nfail, hom_all = red[0], red[1], red[2:6]
        self.last = {"windows": nwin, "received_descriptions": n_recv_total, "sent_bytes": 16 * n_sent, "key_sum_diff": diff,
                     "mode": "minimizer", "homopolymers": hom_all}
        if nfail:
            if failure is not None:
                bad = [s for s in failure[1] if s != OK]
                _check(bad[0])
            raise TSXException(-4, "minimizer step: another rank failed")
        if diff != 0:
            raise RuntimeError("minimizer step: k-mer occurrences described - walked - homopolymers, over all ranks = %d" % diff)
        # the homopolymer k-mers, on their owners, with the totals of all ranks
        sel = [b for b in range(4) if hom_all[b] and int(self.hom_owner[b]) == rank]
        if sel:
            keys = self.hom_dev[sel].contiguous()
            cnts = torch.tensor([hom_all[b] for b in sel], **i64)
            _check(L.tsx_hip_add_kmers_device(m.handle, vp(keys.data_ptr()), vp(cnts.data_ptr()), len(sel), None))
        m.sync()
        return em[1]


def owner_of(kmers, k, world):
    """Rank that owns each one-limb k-mer under the minimizer exchange (numpy uint32)."""
    import numpy as np
    from . import _check, lib
    a = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
    out = np.zeros(a.shape[0], dtype=np.uint32)
    _check(lib().tsx_hip_mini_owner_host(k, world, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), a.shape[0],
                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    return out
