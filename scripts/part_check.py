"""Scratch: partitioned path vs oracle and vs atomic path (not a pytest file)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import tsxcount_amd as T
from tsxcount_amd import synth
from oracle.oracle import Oracle

def check(text, k, l, s=0):
    o = Oracle(k, min(max(l, 16), 2*k-1), 4, seed=1); n = o.count_fastq(text)
    ok, oc = o.dump()
    for path in ("atomic", "partitioned"):
        m = T.TSXHashMapHIP(l, s, k); m.set_path(path)
        m.countFastq(text); st = m.stats()
        got = m.getKmerCounts(ok)
        bad = int((got != oc).sum())
        print(k, l, s, path, 'added', st['kmers_added'], n, 'distinct', st['distinct'], o.distinct(), 'bad', bad, st['insert_failures'])
        assert st['kmers_added'] == n and st['distinct'] == o.distinct() and bad == 0
        # count a second time: incremental build over dirty segments
        m.countFastq(text); got = m.getKmerCounts(ok)
        assert (got == 2 * oc).all() and m.stats()['distinct'] == o.distinct()
        gk, gc = m.getAllKmers(); assert int(gc.sum()) == 2 * n
        m.close()

check(synth.fastq(5, 0, 300), 31, 20)
check(synth.fastq(6, 0, 40), 31, 16)
check(synth.fastq(7, 0, 100), 14, 18, 4)
check(synth.fastq(8, 0, 400), 32, 19)
check(synth.fastq(9, 0, 2000), 31, 24)
check(synth.fastq(9, 0, 20), 21, 15, 2)
check(open('tests/golden/small_t7.1000.fastq','rb').read(), 14, 26, 4)
check(synth.zipf_fastq(7, 3000, 150, 400, 31), 31, 18)
check(synth.fastq(10, 0, 6000), 31, 31)   # 512 x 256 lists
check(synth.fastq(11, 0, 3000), 27, 32)   # 512 x 512 lists
print("PART OK")
