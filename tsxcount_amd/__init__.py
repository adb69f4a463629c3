"""tsxcount_amd -- MI355X (gfx950) k-mer counting hash map behind tsxCount's --mode=HIP.

The product is ``lib/libtsxcount_hip.so`` (hand-written HIP, C ABI in
``include/tsxcount_hip.h``).  This module is the thin host-side mirror of the
reference's ``TSXHashMap`` surface over that ABI (ctypes), used by the tests
and the bench.  There is no CPU fallback: without the built library the
import fails, and without a GPU ``TSXHashMapHIP(...)`` raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtsxcount_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tsxcount_hip.h")

OK, EINVAL, ENODEVICE, ENOMEM, EHIP, EFULL, EOVERFLOW, ERANGE, ELOCK = 0, -1, -2, -3, -4, -5, -6, -7, -8


class TSXException(RuntimeError):
    """Mirror of TSXException (TSXHashMap.h:28-47); carries the C-ABI code."""

    def __init__(self, code, what):
        super().__init__(what)
        self.code = code


class Layout(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int32), ("l", ctypes.c_int32), ("key_limbs", ctypes.c_int32),
                ("entry_limbs", ctypes.c_int32), ("func_bits", ctypes.c_int32),
                ("reprobe_bits", ctypes.c_int32), ("count_bits", ctypes.c_int32),
                ("overflow_l", ctypes.c_int32), ("shard_bits", ctypes.c_int32), ("shard_index", ctypes.c_int32),
                ("max_reprobes", ctypes.c_uint32),
                ("slots", ctypes.c_uint64), ("table_bytes", ctypes.c_uint64)]


class Stats(ctypes.Structure):
    _fields_ = [("kmers_added", ctypes.c_uint64), ("insert_failures", ctypes.c_uint64),
                ("overflow_carries", ctypes.c_uint64), ("overflow_failures", ctypes.c_uint64),
                ("distinct", ctypes.c_uint64), ("overflow_used", ctypes.c_uint64),
                ("lock_timeouts", ctypes.c_uint64), ("fallback_inserts", ctypes.c_uint64),
                ("count_sum", ctypes.c_uint64)]

    def as_dict(self):
        return {f: int(getattr(self, f)) for f, _ in self._fields_}


def build():
    """Compile the library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "all"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    """Load libtsxcount_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("tsxcount_amd: %s is missing; run tsxcount_amd.build() "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, sz, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.tsx_hip_strerror.restype = ctypes.c_char_p
    L.tsx_hip_strerror.argtypes = [ci]
    L.tsx_hip_last_error.restype = ctypes.c_char_p
    L.tsx_hip_key_limbs.argtypes = [ci]
    L.tsx_hip_encode.argtypes = [ctypes.c_char_p, ci, u64p]
    L.tsx_hip_decode.argtypes = [u64p, ci, ctypes.c_char_p]
    L.tsx_hip_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, u64, ci]
    L.tsx_hip_create_shard.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, u64, ci, ci, ci]
    L.tsx_hip_shard_send_capacity.argtypes = [vp, sz, ctypes.POINTER(sz)]
    L.tsx_hip_shard_scan_device.argtypes = [vp, vp, sz, vp, sz, vp, vp, vp, sz, vp, vp]
    L.tsx_hip_shard_scan_window_device.argtypes = [vp, vp, sz, sz, sz, vp, sz, vp, sz, vp, vp, vp, sz, vp, vp, vp]
    L.tsx_hip_shard_build_device.argtypes = [vp, vp, sz, vp, vp]
    L.tsx_hip_shard_build_pieces_device.argtypes = [vp, vp, u64p, u64p, sz, vp, vp]
    L.tsx_hip_add_hashed_device.argtypes = [vp, vp, vp, sz, vp]
    L.tsx_hip_bgzf_index_host.argtypes = [vp, sz, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.tsx_hip_inflate_bgzf_host.argtypes = [ctypes.c_int, vp, sz, vp, sz, ctypes.POINTER(sz)]
    L.tsx_hip_count_fastq_bgzf_host.argtypes = [vp, vp, sz]
    L.tsx_hip_shard_l1_supported.argtypes = [vp]
    L.tsx_hip_shard_l1_window_device.argtypes = [vp, vp, sz, ctypes.c_uint32, ctypes.c_uint32, sz, vp, vp]
    L.tsx_hip_shard_build_l1_device.argtypes = [vp, vp]
    L.tsx_hip_shard_desc_capacity.argtypes = [vp, sz, ctypes.c_int, ctypes.POINTER(sz)]
    L.tsx_hip_shard_desc_window_device.argtypes = [vp, vp, sz, sz, sz, ctypes.c_int, vp, sz, vp, vp, vp]
    L.tsx_hip_shard_walk_device.argtypes = [vp, vp, sz, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, sz, vp, vp]
    L.tsx_hip_shard_filter_device.argtypes = [vp, vp, sz, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, sz, vp, vp]
    L.tsx_hip_mini_supported.argtypes = [vp]
    L.tsx_hip_mini_capacity.argtypes = [vp, sz, ctypes.c_int, ctypes.POINTER(sz)]
    L.tsx_hip_mini_window_device.argtypes = [vp, vp, sz, sz, sz, ctypes.c_int, vp, sz, vp, vp, vp]
    L.tsx_hip_mini_part_capacity.argtypes = [vp, sz, ctypes.c_uint32, ctypes.POINTER(sz)]
    L.tsx_hip_mini_describe_device.argtypes = [vp, vp, sz, sz, sz, vp, vp]
    L.tsx_hip_mini_split_device.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, vp, sz, vp, vp]
    L.tsx_hip_mini_owner_host.argtypes = [ci, ci, u64p, sz, ctypes.POINTER(ctypes.c_uint32)]
    L.tsx_hip_destroy.argtypes = [vp]
    L.tsx_hip_destroy.restype = None
    L.tsx_hip_get_layout.argtypes = [vp, ctypes.POINTER(Layout)]
    L.tsx_hip_clear.argtypes = [vp]
    L.tsx_hip_sync.argtypes = [vp]
    L.tsx_hip_count_fastq_host.argtypes = [vp, ctypes.c_char_p, sz]
    L.tsx_hip_count_fastq_device.argtypes = [vp, vp, sz, vp]
    L.tsx_hip_add_kmers_host.argtypes = [vp, u64p, u64p, sz]
    L.tsx_hip_add_kmers_device.argtypes = [vp, vp, vp, sz, vp]
    L.tsx_hip_get_counts_host.argtypes = [vp, u64p, sz, u64p]
    L.tsx_hip_get_counts_device.argtypes = [vp, vp, sz, vp, vp]
    L.tsx_hip_lookup_host.argtypes = [vp, u64p, sz, u64p, u64p]
    L.tsx_hip_kmer_starts_host.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8), sz]
    L.tsx_hip_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.tsx_hip_dump_host.argtypes = [vp, u64p, u64p, sz, ctypes.POINTER(sz)]
    L.tsx_hip_dump_device.argtypes = [vp, vp, vp, sz, vp, vp]
    L.tsx_hip_partition_device.argtypes = [vp, ci, vp, vp, sz, vp, vp]
    L.tsx_hip_dump_range_device.argtypes = [vp, u64, u64, vp, vp, sz, vp, vp]
    L.tsx_hip_owner_host.argtypes = [vp, u64p, ci]
    L.tsx_hip_hash_apply.argtypes = [vp, u64p, u64p]
    L.tsx_hip_hash_invert.argtypes = [vp, u64p, u64p]
    L.tsx_hip_hash_rows.argtypes = [vp, u64p]
    L.tsx_hip_set_timing.argtypes = [vp, ci]
    L.tsx_hip_get_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_double), u64p]
    L.tsx_hip_get_stage_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_double), u64p]
    L.tsx_hip_set_path.argtypes = [vp, ci]
    L.tsx_hip_set_record_lines.argtypes = [vp, ci]
    L.tsx_hip_synth_fastq_device.argtypes = [u64, u64, u64, ci, vp, sz, u64p, u64p, u64p, ci, vp]
    L.tsx_hip_synth_zipf_device.argtypes = [u64, u64, ctypes.c_uint32, ctypes.c_uint32, u64p, vp, sz, u64p, ci, vp]
    L.tsx_hip_group_create.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(ci), ci, ci, ci, ci, u64, ci]
    L.tsx_hip_group_destroy.argtypes = [vp]
    L.tsx_hip_group_destroy.restype = None
    L.tsx_hip_group_size.argtypes = [vp]
    L.tsx_hip_group_map.argtypes = [vp, ci]
    L.tsx_hip_group_map.restype = vp
    L.tsx_hip_group_comm_name.argtypes = [vp]
    L.tsx_hip_group_comm_name.restype = ctypes.c_char_p
    L.tsx_hip_group_last_error.restype = ctypes.c_char_p
    L.tsx_hip_group_set_record_lines.argtypes = [vp, ci]
    L.tsx_hip_group_clear.argtypes = [vp]
    L.tsx_hip_group_count_fastq_host.argtypes = [vp, ctypes.c_char_p, sz]
    L.tsx_hip_group_merge.argtypes = [vp]
    L.tsx_hip_group_get_counts_host.argtypes = [vp, u64p, sz, u64p]
    L.tsx_hip_group_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.tsx_hip_group_exchanged_entries.argtypes = [vp]
    L.tsx_hip_group_set_exchange.argtypes = [vp, ci]
    L.tsx_hip_group_exchange.argtypes = [vp]
    L.tsx_hip_group_exchanged_entries.restype = u64
    L.tsx_hip_cut_records_host.argtypes = [ctypes.c_char_p, sz, ci, ci, ctypes.POINTER(sz)]
    _lib = L
    return L


def _check(code):
    if code != OK:
        L = lib()
        msg = L.tsx_hip_strerror(code).decode()
        if code in (EHIP, ENODEVICE, ENOMEM):
            extra = L.tsx_hip_last_error().decode()
            if extra:
                msg += " (" + extra + ")"
        raise TSXException(code, msg)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def key_limbs(k):
    return (2 * k + 63) // 64


def encode(seq, k=None):
    """TSXSeqUtils::fromSequence (SequenceUtils.h:86-160) -> uint64 limbs."""
    s = seq.encode() if isinstance(seq, str) else bytes(seq)
    k = len(s) if k is None else k
    out = np.zeros(key_limbs(k), dtype=np.uint64)
    _check(lib().tsx_hip_encode(s, k, _p(out)))
    return out


def decode(limbs, k):
    """TSXSeqUtils::toSequence (SequenceUtils.h:47-84)."""
    a = np.ascontiguousarray(limbs, dtype=np.uint64)
    buf = ctypes.create_string_buffer(k + 1)
    _check(lib().tsx_hip_decode(_p(a), k, buf))
    return buf.value.decode()


def encode_many(seqs, k):
    out = np.zeros((len(seqs), key_limbs(k)), dtype=np.uint64)
    for i, s in enumerate(seqs):
        out[i] = encode(s, k)
    return out


class TSXHashMapHIP:
    """Host mirror of TSXHashMap / TSXHashMapCAS for --mode=HIP.

    Constructor arguments follow TSXHashMap(iL, iStorageBits, iK)
    (TSXHashMap.h:79); method names follow the reference class.
    """

    def __init__(self, iL, iStorageBits, iK, iThreads=0, hash_seed=1, overflow_l=0, device=0, shard_bits=0,
                 shard_index=0):
        self._h = ctypes.c_void_p()
        self._lib = lib()
        _check(self._lib.tsx_hip_create_shard(ctypes.byref(self._h), iK, iL, iStorageBits, overflow_l,
                                              hash_seed, device, shard_bits, shard_index))
        self.layout = Layout()
        _check(self._lib.tsx_hip_get_layout(self._h, ctypes.byref(self.layout)))
        self.k, self.l, self.wk, self.device = iK, iL, self.layout.key_limbs, device

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.tsx_hip_destroy(h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter teardown: ctypes may already be gone
            pass

    @property
    def handle(self):
        if self._h is None:
            raise TSXException(EINVAL, "map is closed")
        return self._h

    # --- reference surface -------------------------------------------------
    def getK(self):
        return self.k

    def getMaxElements(self):
        return int(self.layout.slots)

    def addKmer(self, kmer):
        """TSXHashMap::addKmer (TSXHashMap.h:182); kmer = limbs or sequence."""
        self.addKmers(self._as_kmers([kmer]))
        return True

    def addKmers(self, kmers, counts=None):
        a = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.wk)
        c = None
        if counts is not None:
            c = np.ascontiguousarray(counts, dtype=np.uint64)
            assert c.shape[0] == a.shape[0]
        _check(self._lib.tsx_hip_add_kmers_host(self._h, _p(a), _p(c) if c is not None else None, a.shape[0]))

    def getKmerCount(self, kmer=None):
        """getKmerCount(kmer) (TSXHashMap.h:548) or getKmerCount() (:645)."""
        if kmer is None:
            return self.stats()["distinct"]
        return int(self.getKmerCounts(self._as_kmers([kmer]))[0])

    def getKmerCounts(self, kmers):
        a = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.wk)
        out = np.zeros(a.shape[0], dtype=np.uint64)
        _check(self._lib.tsx_hip_get_counts_host(self._h, _p(a), a.shape[0], _p(out)))
        return out

    def getKmerCountDebug(self, kmers):
        """getKmerCountDebug (TSXHashMap.h:477): (counts, slots); slot = 2^64-1 for absent k-mers."""
        a = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.wk)
        out = np.zeros(a.shape[0], dtype=np.uint64)
        pos = np.zeros(a.shape[0], dtype=np.uint64)
        _check(self._lib.tsx_hip_lookup_host(self._h, _p(a), a.shape[0], _p(out), _p(pos)))
        return out, pos

    def getKmerStarts(self):
        """getKmerStarts (TSXHashMap.h:650) as a numpy bool array over the 2^l slots."""
        nb = (int(self.layout.slots) + 7) // 8
        bits = np.zeros(nb, dtype=np.uint8)
        _check(self._lib.tsx_hip_kmer_starts_host(self._h, bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), nb))
        return np.unpackbits(bits, bitorder="little")[:int(self.layout.slots)].astype(bool)

    def getAllKmers(self):
        """TSXHashMap::getAllKmers (TSXHashMap.h:660) with counts; order unspecified."""
        n = self.stats()["distinct"]
        kmers = np.zeros((max(n, 1), self.wk), dtype=np.uint64)
        counts = np.zeros(max(n, 1), dtype=np.uint64)
        got = ctypes.c_size_t(0)
        _check(self._lib.tsx_hip_dump_host(self._h, _p(kmers), _p(counts), max(n, 1), ctypes.byref(got)))
        return kmers[:got.value], counts[:got.value]

    def dumpRangeDevice(self, slot_lo, slot_hi, kmers_ptr, counts_ptr, cap, n_ptr, stream=None):
        """getAllKmers for the slots [slot_lo, slot_hi) into device buffers (tsx_hip_dump_range_device)."""
        vp = ctypes.c_void_p
        _check(self._lib.tsx_hip_dump_range_device(self._h, slot_lo, slot_hi, vp(kmers_ptr), vp(counts_ptr), cap,
                                                   vp(n_ptr), vp(stream) if stream else None))

    def getKmerCountsDevice(self, kmers_ptr, n, out_ptr, stream=None):
        vp = ctypes.c_void_p
        _check(self._lib.tsx_hip_get_counts_device(self._h, vp(kmers_ptr), n, vp(out_ptr),
                                                   vp(stream) if stream else None))

    def print_stats(self):
        s = self.stats()
        import sys
        print("Used fields: %d" % s["distinct"], file=sys.stderr)
        print("Available fields: %d" % self.getMaxElements(), file=sys.stderr)
        print("k=%d l=%d entry limbs=%d storage bits=%d" % (self.k, self.l, self.layout.entry_limbs,
                                                          self.layout.count_bits), file=sys.stderr)
        return s

    # --- counting path -----------------------------------------------------
    def countFastq(self, data):
        """countKMers (main.cpp:104-218) for one FASTQ text held in host memory."""
        b = bytes(data)
        _check(self._lib.tsx_hip_count_fastq_host(self._h, b, len(b)))

    def countFastqBgzf(self, gz):
        """The same for the image of a blocked gzip (BGZF) file: members inflated on the device (tsx_inflate.h)."""
        b = bytes(gz)
        _check(self._lib.tsx_hip_count_fastq_bgzf_host(self._h, b, len(b)))

    def countFastqDevice(self, dev_ptr, nbytes, stream=None):
        _check(self._lib.tsx_hip_count_fastq_device(self._h, ctypes.c_void_p(dev_ptr), nbytes,
                                                    ctypes.c_void_p(stream) if stream else None))

    def clear(self):
        _check(self._lib.tsx_hip_clear(self._h))

    def sync(self):
        _check(self._lib.tsx_hip_sync(self._h))

    def stats(self):
        s = Stats()
        _check(self._lib.tsx_hip_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def set_timing(self, enable=True):
        _check(self._lib.tsx_hip_set_timing(self._h, 1 if enable else 0))

    def get_timing(self):
        """(line-pass ms, count_fastq_kernel ms, partition+build ms, pieces) since the last call."""
        a, b, c, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_uint64(0)
        _check(self._lib.tsx_hip_get_timing(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c),
                                            ctypes.byref(n)))
        return a.value, b.value, c.value, int(n.value)

    def get_stage_timing(self):
        """({stage: ms}, pieces) since the last call; stages: line, scan, level1, level2, build, gap, post
        (gap = scan end to partition start: the owner split and the key exchange of a sharded run;
        post = overflow queues + deferred list, inserted after the build kernel)."""
        ms, n = (ctypes.c_double * 7)(), ctypes.c_uint64(0)
        _check(self._lib.tsx_hip_get_stage_timing(self._h, ms, ctypes.byref(n)))
        return dict(zip(("line", "scan", "level1", "level2", "build", "gap", "post"), [float(x) for x in ms])), int(n.value)

    def set_record_lines(self, lines):
        """4 = FASTQ records (default), 2 = FASTA as FASTXreader<FASTAEntry> reads it (tsx_hip_set_record_lines)."""
        _check(self._lib.tsx_hip_set_record_lines(self._h, lines))

    def set_path(self, path):
        """0 auto, 1 atomic, 2 partitioned (tsx_hip_set_path)."""
        _check(self._lib.tsx_hip_set_path(self._h, {"auto": 0, "atomic": 1, "partitioned": 2}.get(path, path)))

    # --- mapping -------------------------------------------------------------
    def hash_rows(self):
        out = np.zeros((2 * self.k, self.wk), dtype=np.uint64)
        _check(self._lib.tsx_hip_hash_rows(self._h, _p(out)))
        return out

    def hash_apply(self, kmer):
        a = np.ascontiguousarray(kmer, dtype=np.uint64)
        out = np.zeros(self.wk, dtype=np.uint64)
        _check(self._lib.tsx_hip_hash_apply(self._h, _p(a), _p(out)))
        return out

    def hash_invert(self, key):
        a = np.ascontiguousarray(key, dtype=np.uint64)
        out = np.zeros(self.wk, dtype=np.uint64)
        _check(self._lib.tsx_hip_hash_invert(self._h, _p(a), _p(out)))
        return out

    def owner(self, kmer, nranks):
        a = np.ascontiguousarray(kmer, dtype=np.uint64)
        return int(self._lib.tsx_hip_owner_host(self._h, _p(a), nranks))

    def _as_kmers(self, items):
        out = np.zeros((len(items), self.wk), dtype=np.uint64)
        for i, it in enumerate(items):
            out[i] = encode(it, self.k) if isinstance(it, (str, bytes)) else np.asarray(it, dtype=np.uint64)
        return out


def cut_records(text, parts, lines_per_record=4):
    """Where the multi-GPU host cuts a text: parts + 1 offsets, every one a record boundary of the reference's reader."""
    b = bytes(text)
    out = (ctypes.c_size_t * (parts + 1))()
    _check(lib().tsx_hip_cut_records_host(b, len(b), parts, lines_per_record, out))
    return [int(x) for x in out]


class TSXHashMapHIPGroup:
    """One table per GPU of this node behind one object (tsx_hip_group_*, csrc/tsx_multi.cpp): countFastq cuts the text
    into record shards, every GPU counts its own, the tables are merged (comm "rccl": RCCL, one GPU per rank; "copy":
    device copies, ranks may share a GPU), lookups go to the owner of each k-mer."""

    def __init__(self, gpus, iL, iStorageBits, iK, hash_seed=1, devices=None, comm="rccl", exchange="merge"):
        self._lib = lib()
        self._h = ctypes.c_void_p()
        dv = (ctypes.c_int * gpus)(*devices) if devices is not None else None
        rc = self._lib.tsx_hip_group_create(ctypes.byref(self._h), gpus, dv, iK, iL, iStorageBits, 0, hash_seed,
                                            {"rccl": 0, "copy": 1}[comm])
        self._check(rc)
        self.k, self.wk, self.gpus = iK, key_limbs(iK), gpus
        if exchange != "merge":      # "mini": the minimizer exchange (20 <= k <= 32), nothing is merged afterwards
            self._check(self._lib.tsx_hip_group_set_exchange(self._h, {"merge": 0, "mini": 1}[exchange]))
        self.exchange = exchange

    def _check(self, rc):
        if rc != OK:
            raise TSXException(rc, self._lib.tsx_hip_strerror(rc).decode() + " (" + self._lib.tsx_hip_group_last_error().decode() + ")")

    def close(self):
        if self._h:
            self._lib.tsx_hip_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def comm_name(self):
        return self._lib.tsx_hip_group_comm_name(self._h).decode()

    def set_record_lines(self, lines):
        self._check(self._lib.tsx_hip_group_set_record_lines(self._h, lines))

    def clear(self):
        self._check(self._lib.tsx_hip_group_clear(self._h))

    def countFastq(self, data):
        b = bytes(data)
        self._check(self._lib.tsx_hip_group_count_fastq_host(self._h, b, len(b)))

    def getKmerCounts(self, kmers):
        k = np.ascontiguousarray(np.asarray(kmers, dtype=np.uint64).reshape(-1, self.wk))
        out = np.zeros(k.shape[0], dtype=np.uint64)
        self._check(self._lib.tsx_hip_group_get_counts_host(self._h, _p(k), k.shape[0], _p(out)))
        return out

    def stats(self):
        s = Stats()
        self._check(self._lib.tsx_hip_group_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def rank_stats(self, rank):
        s = Stats()
        _check(self._lib.tsx_hip_get_stats(self._lib.tsx_hip_group_map(self._h, rank), ctypes.byref(s)))
        return s.as_dict()

    def exchanged_entries(self):
        return int(self._lib.tsx_hip_group_exchanged_entries(self._h))


def bgzf_index(gz):
    """(members, text bytes) of a BGZF image, or None when the buffer is not BGZF."""
    b = bytes(gz)
    nm, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = lib().tsx_hip_bgzf_index_host(b, len(b), ctypes.byref(nm), ctypes.byref(nb))
    return (nm.value, nb.value) if rc == OK else None


def bgzf_inflate(gz, device=0):
    """Inflate a BGZF image on the device and return the text (tests, tools)."""
    b = bytes(gz)
    ix = bgzf_index(b)
    if ix is None:
        raise TSXException(EINVAL, "not a BGZF image")
    out = ctypes.create_string_buffer(max(ix[1], 1))
    got = ctypes.c_size_t(0)
    _check(lib().tsx_hip_inflate_bgzf_host(device, b, len(b), out, ix[1], ctypes.byref(got)))
    return out.raw[:got.value]


def bgzf_compress(data, level=6, block=65280):
    """BGZF writer (the layout `bgzip` produces: SAM specification section 4.1) -- for tests and tools; zlib does
    the deflate, one gzip member with a BC extra field per `block` bytes, the empty end-of-file member last."""
    import struct
    import zlib
    out = bytearray()
    data = bytes(data)
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = c.compress(chunk) + c.flush()
        assert len(raw) + 26 <= 65536
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(raw) + 25)
                + raw + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def synth_sizes(seed, first_read, n_reads, k, want_polya=False):
    """Byte and k-mer totals of tsx_hip_synth_fastq_device without generating."""
    nb, nk, npa = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    _check(lib().tsx_hip_synth_fastq_device(seed, first_read, n_reads, k, None, 0, ctypes.byref(nb),
                                            ctypes.byref(nk), ctypes.byref(npa) if want_polya else None, 0, None))
    return int(nb.value), int(nk.value), int(npa.value)


def synth_fastq_device(seed, first_read, n_reads, k, dev_ptr, cap, device=0, stream=None):
    nb, nk = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _check(lib().tsx_hip_synth_fastq_device(seed, first_read, n_reads, k, ctypes.c_void_p(dev_ptr), cap,
                                            ctypes.byref(nb), ctypes.byref(nk), None, device,
                                            ctypes.c_void_p(stream) if stream else None))
    return int(nb.value), int(nk.value)


def synth_zipf_device(seed, n_reads, read_len, thr, dev_ptr=None, cap=0, device=0, stream=None):
    """Zipf-skewed reads straight into device memory (tsx_hip_synth_zipf_device; thr from synth.zipf_thresholds).
    Returns the byte count; dev_ptr None = sizing call."""
    thr = np.ascontiguousarray(thr, dtype=np.uint64)
    nb = ctypes.c_uint64(0)
    _check(lib().tsx_hip_synth_zipf_device(seed, n_reads, read_len, len(thr), _p(thr), ctypes.c_void_p(dev_ptr) if dev_ptr else None,
                                           cap, ctypes.byref(nb), device, ctypes.c_void_p(stream) if stream else None))
    return int(nb.value)
