import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_first():
    """On a GPU box bring torch up before the first test.  Cause (round 3, checked with readelf): this image holds TWO copies
    of the HIP runtime with the same SONAMEs -- /opt/rocm-7.2.0/lib/libamdhip64.so.7 + libhsa-runtime64.so.1, which
    libtsxcount_hip.so is linked against, and torch's own bundled pair (ROCm 7.0, torch/lib, RPATH $ORIGIN).  The dynamic
    loader keeps whichever is loaded FIRST for the whole process.  When ~90 tests of the library alone ran before the
    first torch tensor, the system runtime was the loaded one and torch 2.10+rocm7.0 did not come up on it ("No HIP GPUs
    are available", gpurun_out/r2_pytest3.log); with torch first its bundled runtime serves both, which the library --
    plain HIP runtime API -- is happy with.  Nothing in the library touches HIP_VISIBLE_DEVICES or the device state.
    bench.py and the workers import torch first for the same reason; a C++ host without torch has one runtime only."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda:0")
            torch.cuda.synchronize()
    except Exception:  # noqa: BLE001 -- CPU-only box, or torch without a device: the gpu tests will say so themselves
        pass
    yield


@pytest.fixture(scope="session")
def golden_fastq():
    """data/small_t7.1000.fastq from the reference (a data fixture, 250 reads)."""
    with open(os.path.join(GOLDEN, "small_t7.1000.fastq"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def golden_counts():
    """data/small_t7.1000.fastq.14.count from the reference: {kmer: count}, k=14."""
    ref = {}
    with gzip.open(os.path.join(GOLDEN, "small_t7.1000.fastq.14.count.gz"), "rt") as f:
        for line in f:
            a, b = line.rstrip("\n").split("\t")
            ref[a] = int(b)
    return ref


def python_counts(text, k, lines_per_record=4):
    """Independent dictionary count of a FASTQ (4 lines per record) or FASTA (2, as FASTXreader<FASTAEntry>
    reads it) text with the reference's record rules (FastXReader.h:62-116,359-372, testExecution.h:15-36)."""
    from collections import Counter
    lines = [l for l in text.split(b"\n") if len(l) > 0]
    c = Counter()
    for seq in lines[1::lines_per_record]:
        for i in range(len(seq) - k + 1):
            c[seq[i:i + k]] += 1
    return c
