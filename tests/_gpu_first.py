"""Scratch first-contact script (not a pytest file): golden fixture through the HIP path."""
import sys, time, gzip
sys.path.insert(0, '.')
import numpy as np
import tsxcount_amd as T
from tsxcount_amd import synth
from oracle.oracle import Oracle

data = open('tests/golden/small_t7.1000.fastq', 'rb').read()
ref = {}
for line in gzip.open('tests/golden/small_t7.1000.fastq.14.count.gz', 'rt'):
    a, b = line.split('\t'); ref[a] = int(b)
for (l, s) in [(26, 4), (20, 0), (18, 2)]:
    m = T.TSXHashMapHIP(l, s, 14)
    print('layout', {f: getattr(m.layout, f) for f, _ in m.layout._fields_})
    t = time.time(); m.countFastq(data); dt = time.time() - t
    st = m.stats(); print(st, 'dt=%.3f' % dt)
    kmers = T.encode_many(list(ref.keys()), 14)
    got = m.getKmerCounts(kmers)
    exp = np.array(list(ref.values()), dtype=np.uint64)
    bad = int((got != exp).sum())
    print('l', l, 's', s, 'bad', bad, 'distinct', st['distinct'], 'added', st['kmers_added'])
    assert bad == 0 and st['distinct'] == len(ref) and st['kmers_added'] == 202204
    ak, ac = m.getAllKmers()
    assert int(ac.sum()) == 202204 and len(ak) == len(ref)
    d = {T.decode(ak[i], 14): int(ac[i]) for i in range(0, len(ak), 997)}
    assert all(ref[kk] == v for kk, v in d.items())
    m.close()
# synthetic k=31 vs oracle
txt = synth.fastq(5, 0, 300)
o = Oracle(31, 20, 4, seed=1); n = o.count_fastq(txt)
m = T.TSXHashMapHIP(20, 0, 31); m.countFastq(txt); st = m.stats(); print(st, n, o.distinct())
assert st['kmers_added'] == n and st['distinct'] == o.distinct()
ok, oc = o.dump(); got = m.getKmerCounts(ok); assert (got == oc).all()
# k=63, k=127 multi-limb
for k, l in [(63, 20), (127, 20), (33, 18), (32, 12)]:
    txt = synth.fastq(7, 0, 200)
    o = Oracle(k, 22, 4, seed=1); n = o.count_fastq(txt)
    m = T.TSXHashMapHIP(l if l >= 19 else 20, 0, k); m.countFastq(txt); st = m.stats()
    print(k, st, n, o.distinct(), {f: getattr(m.layout, f) for f, _ in m.layout._fields_})
    assert st['kmers_added'] == n and st['distinct'] == o.distinct()
    ok, oc = o.dump(); got = m.getKmerCounts(ok); assert (got == oc).all(), k
    m.close()
print('ALL OK')
