"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/tsxcount_hip.h declares.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

import tsxcount_amd as T


def declared_symbols():
    text = open(T.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsx_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(T.LIB_PATH), "run __graft_entry__.build()"
    assert os.path.dirname(T.LIB_PATH).endswith(os.path.join("tsxcount_amd", "lib"))


def test_every_declared_symbol_is_exported():
    L = T.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "missing export: " + s


def test_header_cites_reference_interfaces():
    text = open(T.HEADER_PATH).read()
    for cite in ("TSXHashMap.h:182", "TSXHashMap.h:548", "TSXHashMap.h:645", "TSXHashMap.h:660",
                 "main.cpp:104-218", "SequenceUtils.h:86-160", "TSXHashMapCAS.h:239-245"):
        assert cite in text


def test_host_only_entry_points():
    assert T.key_limbs(31) == 1 and T.key_limbs(32) == 1 and T.key_limbs(33) == 2 and T.key_limbs(127) == 4
    assert T.lib().tsx_hip_key_limbs(128) == T.EINVAL
    # A=0 C=1 G=2 T=3, base i at bits 2i,2i+1 (SequenceUtils.h:98-125)
    assert int(T.encode("ACGT")[0]) == 0b11100100
    assert T.decode(T.encode("GATTACA"), 7) == "GATTACA"
    long = "ACGTTGCA" * 15 + "ACGTTGC"
    assert len(long) == 127 and T.decode(T.encode(long), 127) == long
    # lower case encodes like upper case; any other byte gets the fixed stand-in code
    assert (T.encode("acgt") == T.encode("ACGT")).all()
    assert (T.encode("ANNA") == T.encode("AAAA")).all()
    assert T.lib().tsx_hip_strerror(T.EFULL).decode().startswith("Could not insert kmer")


def test_geometry_errors_need_no_gpu():
    h = ctypes.c_void_p()
    L = T.lib()
    # 2k <= l: TSXException("Invalid lengths ...") in the reference (TSXHashMap.h:91-94)
    assert L.tsx_hip_create(ctypes.byref(h), 10, 20, 4, 0, 1, 0) == T.EINVAL
    assert L.tsx_hip_create(ctypes.byref(h), 0, 20, 4, 0, 1, 0) == T.EINVAL
    assert L.tsx_hip_create(ctypes.byref(h), 128, 20, 4, 0, 1, 0) == T.EINVAL
    assert L.tsx_hip_create(ctypes.byref(h), 31, 20, 33, 0, 1, 0) == T.EINVAL
    with pytest.raises(T.TSXException):
        T.TSXHashMapHIP(20, 4, 10)


def test_no_cpu_fallback_without_a_gpu():
    L = T.lib()
    if L.tsx_hip_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(T.TSXException) as e:
        T.TSXHashMapHIP(20, 4, 14)
    assert e.value.code == T.ENODEVICE


def test_product_does_not_reach_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "tsxcount_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower().replace("no cpu", ""), os.path.join(dirpath, f)


def test_synth_sizes_match_numpy_generator():
    from tsxcount_amd import synth
    for seed, first, n, k in ((5, 0, 17, 31), (9, 1000, 23, 14), (1, 99999, 5, 63)):
        text = synth.fastq(seed, first, n)
        nb, nk, npa = T.synth_sizes(seed, first, n, k, want_polya=True)
        assert nb == len(text)
        seqs = [l for l in text.split(b"\n") if l][1::4]
        assert nk == sum(max(0, len(s) - k + 1) for s in seqs)
        assert npa == sum(sum(1 for i in range(len(s) - k + 1) if s[i:i + k] == b"A" * k) for s in seqs)
