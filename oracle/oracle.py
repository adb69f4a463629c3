"""ctypes binding for oracle/libtsx_oracle.so (the CPU parity checker).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from tsxcount_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtsx_oracle.so")
_lib = None


def build():
    """Compile the C restatement (and, where /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_key_limbs.argtypes = [ctypes.c_void_p]
        L.orc_hash_rows.argtypes = [ctypes.c_void_p, u64p]
        L.orc_hash_apply.argtypes = [ctypes.c_void_p, u64p, u64p]
        L.orc_hash_invert.argtypes = [ctypes.c_void_p, u64p, u64p]
        L.orc_encode.argtypes = [ctypes.c_char_p, ctypes.c_int, u64p]
        L.orc_add_kmer.argtypes = [ctypes.c_void_p, u64p]
        L.orc_get_count.restype = ctypes.c_uint64
        L.orc_get_count.argtypes = [ctypes.c_void_p, u64p]
        for f in ("orc_distinct", "orc_adds", "orc_used_slots"):
            getattr(L, f).restype = ctypes.c_uint64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.orc_failed.argtypes = [ctypes.c_void_p]
        L.orc_dump.restype = ctypes.c_uint64
        L.orc_dump.argtypes = [ctypes.c_void_p, u64p, u64p, ctypes.c_uint64]
        L.orc_count_fastq.restype = ctypes.c_int64
        L.orc_count_fastq.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64]
        L.orc_count_fastx.restype = ctypes.c_int64
        L.orc_count_fastx.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


class Oracle:
    """Serial restatement of TSXHashMapPerf / TSXHashMapCAS (see tsx_oracle.c)."""

    def __init__(self, k, l, storagebits=4, seed=1):
        self.k, self.l, self.s, self.seed = k, l, storagebits, seed
        self._h = lib().orc_create(k, l, storagebits, seed)
        if not self._h:
            raise ValueError("Invalid lengths for hashmap size and value of k")
        self.wk = lib().orc_key_limbs(self._h)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    __del__ = close

    def encode(self, seq):
        out = np.zeros(self.wk, dtype=np.uint64)
        s = seq.encode() if isinstance(seq, str) else seq
        assert len(s) == self.k
        lib().orc_encode(s, self.k, _p(out))
        return out

    def hash_rows(self):
        out = np.zeros((2 * self.k, self.wk), dtype=np.uint64)
        lib().orc_hash_rows(self._h, _p(out))
        return out

    def hash_apply(self, kmer):
        k = np.ascontiguousarray(kmer, dtype=np.uint64)
        out = np.zeros(self.wk, dtype=np.uint64)
        lib().orc_hash_apply(self._h, _p(k), _p(out))
        return out

    def hash_invert(self, key):
        k = np.ascontiguousarray(key, dtype=np.uint64)
        out = np.zeros(self.wk, dtype=np.uint64)
        lib().orc_hash_invert(self._h, _p(k), _p(out))
        return out

    def add_kmer(self, kmer):
        k = np.ascontiguousarray(kmer, dtype=np.uint64)
        return bool(lib().orc_add_kmer(self._h, _p(k)))

    def get_count(self, kmer):
        k = np.ascontiguousarray(kmer, dtype=np.uint64)
        return int(lib().orc_get_count(self._h, _p(k)))

    def count_fastq(self, data, lines_per_record=4):
        n = lib().orc_count_fastx(self._h, data, len(data), lines_per_record)
        if n < 0:
            raise RuntimeError("oracle table exhausted")
        return int(n)

    def distinct(self):
        return int(lib().orc_distinct(self._h))

    def adds(self):
        return int(lib().orc_adds(self._h))

    def used_slots(self):
        return int(lib().orc_used_slots(self._h))

    def dump(self):
        n = self.distinct()
        kmers = np.zeros((n, self.wk), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint64)
        w = lib().orc_dump(self._h, _p(kmers), _p(counts), n)
        assert w == n
        return kmers, counts


REF_BIN = os.path.join(_HERE, "_ref", "tsxCount_ref")
