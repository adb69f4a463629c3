"""How far does base-composition skew unbalance the 16 Ki-slot segments?  AT-rich random reads
(A,T 40 % each), tables filled to ~0.5 / 0.7 / 0.85: insert failures on either path would mean a
segment overflowed while the table as a whole still had room."""
import sys
sys.path.insert(0, '.')
import numpy as np
import tsxcount_amd as T
rng = np.random.default_rng(1)
k = 31
n_reads, L = 60000, 300
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
codes = rng.choice(4, size=(n_reads, L), p=[0.4, 0.1, 0.1, 0.4])
parts = []
for r in range(n_reads):
    s = lut[codes[r]].tobytes()
    parts.append(b"@r%d\n" % r + s + b"\n+\n" + b"I" * L + b"\n")
text = b"".join(parts)
nk = n_reads * (L - k + 1)
for l in (25, 24):
    for path in ("atomic", "partitioned"):
        m = T.TSXHashMapHIP(l, 0, k); m.set_path(path)
        try:
            m.countFastq(text); err = None
        except T.TSXException as e:
            err = str(e)
        st = m.stats()
        print('l=%d load=%.2f %s' % (l, st['distinct'] / (1 << l), path), 'failures', st['insert_failures'], 'fallback', st['fallback_inserts'], err)
        m.close()
