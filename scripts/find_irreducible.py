"""Irreducible polynomials over GF(2) of every even degree 2..254 (trinomials where one exists, else
pentanomials), found with Rabin's test.  Prints the table make_mapping() embeds (GF_POLY_EXP)."""
import sys


def mulmod(a, b, p, n):
    r = 0
    while b:
        if b & 1:
            r ^= a
        b >>= 1
        a <<= 1
        if a >> n:
            a ^= p
    return r


def powx2k(k, p, n):
    """x^(2^k) mod p"""
    r = 2
    for _ in range(k):
        r = mulmod(r, r, p, n)
    return r


def gcd(a, b):
    while b:
        while a.bit_length() >= b.bit_length() and a:
            a ^= b << (a.bit_length() - b.bit_length())
        a, b = b, a
    return a


def prime_factors(n):
    f, d = [], 2
    while d * d <= n:
        if n % d == 0:
            f.append(d)
            while n % d == 0:
                n //= d
        d += 1
    if n > 1:
        f.append(n)
    return f


def irreducible(p, n):
    if powx2k(n, p, n) != 2:
        return False
    for q in prime_factors(n):
        if gcd(p, powx2k(n // q, p, n) ^ 2) != 1:
            return False
    return True


def find(n):
    for a in range(1, n):
        p = (1 << n) | (1 << a) | 1
        if irreducible(p, n):
            return (a, 0, 0)
    for a in range(3, n):
        for b in range(2, a):
            for c in range(1, b):
                p = (1 << n) | (1 << a) | (1 << b) | (1 << c) | 1
                if irreducible(p, n):
                    return (a, b, c)
    raise SystemExit("none for %d" % n)


rows = []
for n in range(2, 256, 2):
    rows.append(find(n))
print("static const uint8_t GF_POLY_EXP[127][3] = {  // degree 2, 4, ..., 254: x^n + x^a [+ x^b + x^c] + 1")
for i in range(0, len(rows), 8):
    print("    " + " ".join("{%d, %d, %d}," % r for r in rows[i:i + 8]))
print("};")
