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
    # two-limb keys; the reference itself aborts ("terminate called after throwing
    # an instance of 'char const*'") for k = 40, 47, 63, so k = 33 is as far as it pins
    {"input": "synth", "seed": 23, "n_reads": 60, "k": 33, "l": 17, "s": 6},
]


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
    json.dump({"generator": "tests/golden/make_ref_runs.py", "runs": runs},
              open(os.path.join(ROOT, "tests", "golden", "ref_runs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
