#!/usr/bin/env python3
"""Regenerates tests/golden/ref_runs.json (needs /root/reference and the binary
built by oracle/Makefile).  For every case: write the FASTQ, write a
<fastq>.<k>.count file FROM THE ORACLE's dump, run the real reference
`tsxCount --mode=CAS --threads=1 --check` on it and record what it printed.
"total errors0" + equal distinct counts mean the reference's own getKmerCount
returned the oracle's count for every k-mer.

threads=1 and 2k+s a multiple of 8: the reference's byte-wise CAS stores are
only reliable single-threaded with byte-aligned slots.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import tsxcount_amd as T  # noqa: E402
from oracle.oracle import REF_BIN, Oracle  # noqa: E402
from tsxcount_amd import synth  # noqa: E402

CASES = [
    {"input": "golden", "k": 14, "l": 20, "s": 4},
    {"input": "synth", "seed": 21, "n_reads": 120, "k": 15, "l": 18, "s": 2},
    {"input": "synth", "seed": 22, "n_reads": 150, "k": 31, "l": 19, "s": 2},
    # two-limb keys; the reference's CLI aborts ("terminate called after throwing an instance of
    # 'char const*'") for k = 40, 47, 63 in --mode=CAS and in the --check loader: k = 33 is as far as the
    # CLI pins.  PERF_CASES below go further through the reference's serial table.
    {"input": "synth", "seed": 23, "n_reads": 60, "k": 33, "l": 17, "s": 6},
]

# k >= 40 through oracle/_ref/ref_perf_driver (oracle/ref_perf_driver.cpp): the reference's own FASTQ reader,
# fromSequence, TSXHashMapPerf::addKmer and getKmerCount(kmer).  It answers every k-mer whose counter has not
# overflowed; for an overflowed one the reference throws from its overflow walk (UBigInt arithmetic "not yet
# implemented") -- those k-mers are listed as `thrown` and stay unpinned.
PERF_CASES = [
    {"input": "synth", "seed": 31, "n_reads": 15, "k": 40, "l": 18, "s": 12},     # nothing overflows, polyA included
    {"input": "synth", "seed": 32, "n_reads": 15, "k": 47, "l": 18, "s": 10},     # polyA overflows
    {"input": "synth", "seed": 33, "n_reads": 15, "k": 55, "l": 18, "s": 6},
    {"input": "repeated", "seed": 7, "n_reads": 40, "lo": 150, "hi": 400, "k": 63, "l": 17, "s": 2},   # counts 1..3
    {"input": "synth", "seed": 34, "n_reads": 10, "k": 63, "l": 17, "s": 2},      # the A-tails overflow
    {"input": "synth", "seed": 35, "n_reads": 12, "k": 33, "l": 17, "s": 6},      # and one k the CLI pins as well
]


def perf_text(c):
    if c["input"] == "repeated":
        return synth.repeated_reads_fastq(c["seed"], c["n_reads"], c["lo"], c["hi"])
    return synth.fastq(c["seed"], 0, c["n_reads"])


def perf_digest(pairs):
    """sha256 over the sorted `kmer<TAB>count` lines."""
    h = hashlib.sha256()
    for kmer, cnt in sorted(pairs):
        h.update(b"%s\t%d\n" % (kmer, cnt))
    return h.hexdigest()


def perf_runs():
    driver = os.path.join(os.path.dirname(REF_BIN), "ref_perf_driver")
    runs = []
    for c in PERF_CASES:
        text = perf_text(c)
        with tempfile.TemporaryDirectory() as td:
            fq = os.path.join(td, "in.fastq")
            open(fq, "wb").write(text)
            p = subprocess.run([driver, fq, str(c["k"]), str(c["l"]), str(c["s"])], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, timeout=1500)
        assert p.returncode == 0, p.returncode
        pairs, thrown = [], []
        for line in p.stdout.split(b"\n"):
            if b"\t" not in line:
                continue
            kmer, cnt = line.split(b"\t")
            if cnt == b"!":
                thrown.append(kmer.decode())
            else:
                pairs.append((kmer, int(cnt)))
        # the oracle on the same text, the thrown k-mers left out
        o = Oracle(c["k"], c["l"], c["s"], seed=1)
        o.count_fastq(text)
        kmers, counts = o.dump()
        mine = [(T.decode(kmers[i], c["k"]).encode(), int(counts[i])) for i in range(len(kmers))]
        mine = [x for x in mine if x[0].decode() not in set(thrown)]
        r = dict(c)
        r.update({"reference_answered": len(pairs), "reference_thrown": thrown, "reference_sha256": perf_digest(pairs),
                  "oracle_sha256": perf_digest(mine), "max_count_answered": max(x[1] for x in pairs),
                  "command": "ref_perf_driver in.fastq %d %d %d" % (c["k"], c["l"], c["s"])})
        print({k2: v for k2, v in r.items() if k2 != "reference_thrown"}, len(thrown), "thrown", flush=True)
        assert r["reference_sha256"] == r["oracle_sha256"], "the restatement disagrees with the reference"
        runs.append(r)
    return runs


def main():
    runs = []
    for c in CASES:
        if c["input"] == "golden":
            text = open(os.path.join(ROOT, "tests", "golden", "small_t7.1000.fastq"), "rb").read()
        else:
            text = synth.fastq(c["seed"], 0, c["n_reads"])
        o = Oracle(c["k"], c["l"], c["s"], seed=1)
        o.count_fastq(text)
        kmers, counts = o.dump()
        order = np.lexsort(kmers.T[::-1])
        h = hashlib.sha256()
        h.update(kmers[order].tobytes())
        h.update(counts[order].tobytes())
        with tempfile.TemporaryDirectory() as td:
            fq = os.path.join(td, "in.fastq")
            open(fq, "wb").write(text)
            with open(fq + ".%d.count" % c["k"], "w") as f:
                for i in range(len(kmers)):
                    f.write("%s\t%d\n" % (T.decode(kmers[i], c["k"]), int(counts[i])))
            cmd = [REF_BIN, "--input=" + fq, "--k=%d" % c["k"], "--l=%d" % c["l"], "--s=%d" % c["s"],
                   "--mode=CAS", "--threads=1", "--check"]
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=1500)
            out = p.stdout.decode(errors="replace")
        errs = int(re.search(r"total errors(\d+)", out).group(1))
        distinct = int(re.search(r"Added a total of (\d+) different kmers", out).group(1))
        r = dict(c)
        r.update({"reference_total_errors": errs, "reference_distinct": distinct, "reference_rc": p.returncode,
                  "oracle_distinct": o.distinct(), "oracle_counts_sha256": h.hexdigest(),
                  "command": " ".join(["tsxCount"] + cmd[2:])})
        print(r, flush=True)
        runs.append(r)
    json.dump({"generator": "tests/golden/make_ref_runs.py", "runs": runs, "perf_runs": perf_runs()},
              open(os.path.join(ROOT, "tests", "golden", "ref_runs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
