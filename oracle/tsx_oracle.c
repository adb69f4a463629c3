/*
 * tsx_oracle.c -- CPU restatement of tsxCount's k-mer counting path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under tsxcount_amd/ links, imports or calls it.
 *
 * What it restates (reference file:line, repo mjoppich/tsxCount):
 *   - FASTQ scan ............ src/fastxutils/FastXReader.h:307-385 (skip empty
 *                             lines, 4 lines per record, line 2 = sequence)
 *   - k-mer enumeration ..... src/mains/testExecution.h:15-36 (createKMers,
 *                             all len-k+1 windows, reads shorter than k skipped)
 *   - 2-bit encode .......... src/utils/SequenceUtils.h:86-160 (fromSequence,
 *                             base i -> bits 2i,2i+1; A=0 C=1 G=2 T=3)
 *   - bijective hash ........ src/tsxcount/BijectiveKMapping.h:202-225,284-303
 *   - table geometry ........ src/tsxcount/TSXHashMap.h:79-154,1135-1189
 *   - addKmer ............... src/tsxcount/TSXHashMapPerf.h:56-205 (the serial
 *                             body that TSXHashMapCAS.h:268-508 repeats with
 *                             byte-wise CAS stores)
 *   - increment / overflow .. src/tsxcount/TSXHashMapPerf.h:218-289,300-424,
 *                             426-462,547-697,699-881
 *   - getKmerCount .......... src/tsxcount/TSXHashMap.h:548-638,951-1039
 *   - getKmerCount() ........ src/tsxcount/TSXHashMap.h:645-648 (kmerStarts)
 *   - getAllKmers ........... src/tsxcount/TSXHashMap.h:660-722
 *
 * Two deliberate, documented differences from the reference:
 *   1. The reference seeds its random hash matrix with time(NULL)
 *      (BijectiveKMapping.h:84,287).  Here the same family (unit upper
 *      triangular over GF(2)) is drawn from a caller-supplied seed so runs
 *      are reproducible.  Counts do not depend on the matrix.
 *   2. The reference maps non-ACGT bytes to rand()%2 bits
 *      (SequenceUtils.h:126-136).  Here every byte b maps to
 *      ((b>>1)^(b>>2))&3, which equals the reference code for A,C,G,T (and
 *      a,c,g,t) and is a fixed stand-in for the random draw otherwise.
 *
 * Parity pin: tests/test_oracle.py checks this file against the reference's
 * own fixture (data/small_t7.1000.fastq + .14.count, copied as data under
 * tests/golden/) and against runs of the real reference binary built by
 * oracle/Makefile into oracle/_ref/ (tests/golden/ref_runs.json).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define ORC_LIMBS 5 /* 2k+s <= 254+64 bits */

typedef struct { uint64_t w[ORC_LIMBS]; } big_t;

typedef struct {
    int k, l, s;          /* k-mer length, log2(#slots), storage (value) bits */
    int n;                /* 2k: key bits */
    int kv;               /* 2k+s: bits per slot (TSXHashMap.h:84) */
    int wk;               /* limbs per key */
    uint64_t slots;       /* 2^l */
    uint64_t max_reprobes;/* (1<<l)-1 (TSXHashMap.h:86) */
    uint8_t *table;       /* bit-packed counter array (TSXHashMap.h:103) */
    uint64_t table_bytes;
    uint8_t *starts;      /* m_iKmerStarts bitset (TSXHashMap.h:107) */
    big_t *rows;          /* hash matrix rows, row i <-> output bit n-1-i */
    big_t *irows;         /* inverse matrix rows */
    uint64_t adds;        /* successful addKmer calls */
    uint64_t used;        /* occupied slots incl. overflow slots */
    int failed;           /* set when an insert ran out of reprobes */
} orc_t;

/* ---------- small fixed-width big integer helpers (UBigInt stand-in) ------ */

static big_t big_zero(void) { big_t r; memset(&r, 0, sizeof r); return r; }

static int big_is_zero(const big_t *a) {
    uint64_t o = 0; for (int i = 0; i < ORC_LIMBS; ++i) o |= a->w[i]; return o == 0;
}
static int big_eq(const big_t *a, const big_t *b) {
    return memcmp(a, b, sizeof(big_t)) == 0;
}
static big_t big_shl(big_t a, int sh) {
    big_t r = big_zero(); int ws = sh >> 6, bs = sh & 63;
    for (int i = ORC_LIMBS - 1; i >= ws; --i) {
        uint64_t v = a.w[i - ws] << bs;
        if (bs && i - ws - 1 >= 0) v |= a.w[i - ws - 1] >> (64 - bs);
        r.w[i] = v;
    }
    return r;
}
static big_t big_shr(big_t a, int sh) {
    big_t r = big_zero(); int ws = sh >> 6, bs = sh & 63;
    for (int i = 0; i + ws < ORC_LIMBS; ++i) {
        uint64_t v = a.w[i + ws] >> bs;
        if (bs && i + ws + 1 < ORC_LIMBS) v |= a.w[i + ws + 1] << (64 - bs);
        r.w[i] = v;
    }
    return r;
}
static big_t big_or(big_t a, big_t b) { for (int i = 0; i < ORC_LIMBS; ++i) a.w[i] |= b.w[i]; return a; }

/* keep the low `bits` bits (UBigInt::resize / mod2) */
static big_t big_trunc(big_t a, int bits) {
    for (int i = 0; i < ORC_LIMBS; ++i) {
        int lo = i * 64;
        if (bits <= lo) a.w[i] = 0;
        else if (bits < lo + 64) a.w[i] &= (~0ULL) >> (64 - (bits - lo));
    }
    return a;
}
static big_t big_from_u64(uint64_t v) { big_t r = big_zero(); r.w[0] = v; return r; }
static big_t big_add_u64(big_t a, uint64_t v) {
    for (int i = 0; i < ORC_LIMBS && v; ++i) {
        uint64_t s = a.w[i] + v; v = (s < a.w[i]) ? 1 : 0; a.w[i] = s;
    }
    return a;
}
static int big_all_ones(const big_t *a, int bits) { /* (~x).isZero() on `bits` bits */
    big_t m = big_trunc(*a, bits);
    for (int i = 0; i < ORC_LIMBS; ++i) {
        int lo = i * 64; uint64_t want;
        if (bits <= lo) want = 0;
        else if (bits < lo + 64) want = (~0ULL) >> (64 - (bits - lo));
        else want = ~0ULL;
        if (m.w[i] != want) return 0;
    }
    return 1;
}
static int big_parity_and(const big_t *a, const big_t *b) {
    uint64_t x = 0; for (int i = 0; i < ORC_LIMBS; ++i) x ^= (a->w[i] & b->w[i]);
    return __builtin_parityll(x);
}
static int big_bit(const big_t *a, int i) { return (a->w[i >> 6] >> (i & 63)) & 1; }
static void big_setbit(big_t *a, int i, int v) {
    if (v) a->w[i >> 6] |= (1ULL << (i & 63)); else a->w[i >> 6] &= ~(1ULL << (i & 63));
}

/* ---------- deterministic RNG (replaces srand(time(NULL)) / rand()) ------- */

static uint64_t splitmix64(uint64_t *st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* ---------- bijective GF(2) mapping -------------------------------------- */
/* BijectiveKMapping.h:284-303: M[i][i]=1, M[i][j]=rand bit for j>i, 0 below.
 * matrixToRows (227-256): row i gets bit (n-1-j) = M[i][j].
 * applyto (202-225): out bit (n-1-i) = parity(row_i & x).
 * Draw order: i ascending, j ascending, one splitmix64 word per 64 draws.   */
static void orc_make_hash(orc_t *o, uint64_t seed) {
    int n = o->n;
    o->rows = (big_t *)calloc(n, sizeof(big_t));
    o->irows = (big_t *)calloc(n, sizeof(big_t));
    uint64_t st = seed, word = 0; int have = 0;
    for (int i = 0; i < n; ++i) {
        big_setbit(&o->rows[i], n - 1 - i, 1);
        for (int j = i + 1; j < n; ++j) {
            if (!have) { word = splitmix64(&st); have = 64; }
            int bit = (int)(word & 1); word >>= 1; --have;
            big_setbit(&o->rows[i], n - 1 - j, bit);
        }
    }
    /* Inverse of a unit upper triangular matrix over GF(2) by back
     * substitution (the reference gets the same matrix through LU + lubksb,
     * BijectiveKMapping.h:643-766).  In bit terms: out bit p = x_p ^ f(x_q, q<p)
     * so x is recovered from the lowest bit up.  irows[i] is the row that
     * yields original bit (n-1-i) from the hashed value.                    */
    /* Let A be the matrix acting on bit-vectors indexed by bit position p:
     * y_p = x_p ^ sum_{q<p} a[p][q] x_q.  Then x_p = y_p ^ sum_{q<p} a[p][q] x_q,
     * and each x_q (q<p) is already a known linear form in y.               */
    big_t *xform = (big_t *)calloc(n, sizeof(big_t)); /* x_p as a mask over y bits */
    for (int p = 0; p < n; ++p) {
        const big_t *row = &o->rows[n - 1 - p]; /* row producing output bit p */
        big_t acc = big_zero(); big_setbit(&acc, p, 1);
        for (int q = 0; q < p; ++q)
            if (big_bit(row, q))
                for (int t = 0; t < ORC_LIMBS; ++t) acc.w[t] ^= xform[q].w[t];
        xform[p] = acc;
    }
    for (int p = 0; p < n; ++p) o->irows[n - 1 - p] = xform[p];
    free(xform);
}

static big_t orc_apply_rows(const orc_t *o, const big_t *rows, const big_t *x) {
    big_t r = big_zero(); int n = o->n;
    for (int i = 0; i < n; ++i)
        if (big_parity_and(&rows[i], x)) big_setbit(&r, n - 1 - i, 1);
    return r;
}

/* ---------- bit-packed slot access (TSXHashMap.h:1135-1189) --------------- */
/* The reference computes pos*kv in 32 bits (TSXHashMap.h:1143), which limits
 * it to tables under 2^32 bits; 64-bit arithmetic here, same layout.        */
static big_t orc_get(const orc_t *o, uint64_t pos) {
    uint64_t bit = pos * (uint64_t)o->kv; uint64_t byte = bit >> 3; int off = (int)(bit & 7);
    big_t r = big_zero();
    int nbytes = (off + o->kv + 7) >> 3;
    uint8_t tmp[ORC_LIMBS * 8 + 8]; memset(tmp, 0, sizeof tmp);
    memcpy(tmp, o->table + byte, nbytes);
    /* assemble little-endian, then shift right by off */
    uint64_t limbs[ORC_LIMBS + 1]; memcpy(limbs, tmp, sizeof limbs);
    for (int i = 0; i < ORC_LIMBS; ++i) {
        uint64_t v = limbs[i] >> off;
        if (off) v |= limbs[i + 1] << (64 - off);
        r.w[i] = v;
    }
    return big_trunc(r, o->kv);
}
static void orc_put(orc_t *o, uint64_t pos, big_t v) {
    uint64_t bit0 = pos * (uint64_t)o->kv;
    for (int i = 0; i < o->kv; ++i) {
        uint64_t b = bit0 + i; uint8_t m = (uint8_t)(1u << (b & 7));
        if (big_bit(&v, i)) o->table[b >> 3] |= m; else o->table[b >> 3] &= (uint8_t)~m;
    }
}
static int orc_is_start(const orc_t *o, uint64_t pos) { return (o->starts[pos >> 3] >> (pos & 7)) & 1; }
static void orc_set_start(orc_t *o, uint64_t pos) { o->starts[pos >> 3] |= (uint8_t)(1u << (pos & 7)); }

/* reprobe(i) = i(i+1)/2 truncated to l bits (TSXHashMap.h:1046-1054);
 * getPosition = (basekey + reprobe(i)) mod 2^l (TSXHashMap.h:759-778).       */
static uint64_t orc_position(const orc_t *o, const big_t *basekey, uint64_t reprobes) {
    uint32_t i = (uint32_t)reprobes;
    uint32_t j = i * (i + 1) / 2;                 /* 32-bit, as in the reference */
    uint64_t tri = (o->l >= 64) ? j : ((uint64_t)j & (o->slots - 1));
    return (basekey->w[0] + tri) & (o->slots - 1);
}
/* makeKey (TSXHashMap.h:1056-1072): func bits of basekey | reprobe in low l bits */
static big_t orc_make_key(const orc_t *o, const big_t *basekey, uint64_t reprobe) {
    big_t f = big_shl(big_shr(*basekey, o->l), o->l);
    big_t r = big_trunc(big_from_u64(reprobe), o->l);
    return big_trunc(big_or(f, r), o->n);
}
/* makeOverflowReprobe (TSXHashMapPerf.h:426-445) */
static big_t orc_overflow_reprobe(const orc_t *o, uint64_t reprobe, uint64_t perf) {
    int L = o->l, pb = L / 2;
    big_t r = big_trunc(big_from_u64(reprobe), L);
    r = big_trunc(big_shl(r, L - pb), L);
    big_t p = big_trunc(big_from_u64(perf), L);
    return big_or(r, p);
}
/* positionMatchesKeyAndReprobe (TSXHashMap.h:1076-1114) */
static int orc_matches_key(const orc_t *o, const big_t *elem, const big_t *basekey, uint64_t reprobe) {
    big_t ef = big_shr(*elem, o->l + o->s);
    big_t kf = big_shr(*basekey, o->l);
    if (!big_eq(&ef, &kf)) return 0;
    big_t er = big_trunc(big_shr(*elem, o->s), o->l);
    big_t rr = big_trunc(big_from_u64(reprobe), o->n); /* resize(2k) then compare with key&mask_l */
    return big_eq(&er, &rr);
}
/* positionMatchesOverflowReprobe (TSXHashMapPerf.h:447-462) */
static int orc_matches_overflow(const orc_t *o, const big_t *elem, uint64_t reprobe, uint64_t perf) {
    big_t er = big_trunc(big_shr(*elem, o->s), o->l);
    big_t rp = orc_overflow_reprobe(o, reprobe, perf);
    return big_eq(&er, &rp);
}

static int orc_handle_overflow(orc_t *o, const big_t *basekey, uint64_t reprobe);

/* incrementElement_key_value (TSXHashMapPerf.h:218-289): 1 = ok, 2 = value wrapped */
static int orc_inc_value(orc_t *o, uint64_t pos) {
    big_t e = orc_get(o, pos);
    big_t val = big_trunc(e, o->s);
    big_t key = big_shl(big_shr(e, o->s), o->s);
    if (big_all_ones(&val, o->s)) { orc_put(o, pos, key); return 2; }
    val = big_add_u64(val, 1);
    orc_put(o, pos, big_or(key, val));
    return 1;
}
/* incrementElement_func (TSXHashMapPerf.h:300-424): 1 = ok, 2 = func wrapped */
static int orc_inc_func(orc_t *o, uint64_t pos) {
    big_t e = orc_get(o, pos);
    int fb = o->n - o->l;
    big_t func = big_shr(e, o->l + o->s);
    big_t low = big_trunc(e, o->l + o->s);
    if (big_all_ones(&func, fb)) { orc_put(o, pos, low); return 2; }
    func = big_add_u64(func, 1);
    orc_put(o, pos, big_or(low, big_shl(func, o->l + o->s)));
    return 1;
}
/* incrementElement_new (TSXHashMapPerf.h:547-697) */
static int orc_inc_new(orc_t *o, uint64_t pos, const big_t *basekey, uint64_t reprobes, int key_is_value) {
    int st = orc_inc_value(o, pos);
    if (st == 1) return 1;
    if (!key_is_value) return 2;
    st = orc_inc_func(o, pos);
    if (st == 1) return 1;
    if (!orc_handle_overflow(o, basekey, reprobes)) return 0;
    return 1;
}
/* handleOverflow (TSXHashMapPerf.h:699-881) */
static int orc_handle_overflow(orc_t *o, const big_t *basekey, uint64_t reprobe) {
    uint64_t perf = 0;
    while (perf < o->max_reprobes) {
        perf += 1;
        uint64_t pos = orc_position(o, basekey, reprobe + perf);
        big_t e = orc_get(o, pos);
        big_t keypart = big_shr(e, o->s);
        if (big_is_zero(&keypart)) {
            big_t kvnew = big_shl(orc_overflow_reprobe(o, reprobe, perf), o->s);
            kvnew.w[0] |= 1;
            orc_put(o, pos, big_trunc(kvnew, o->kv));
            o->used += 1;
            return 1;
        }
        if (orc_is_start(o, pos) || !orc_matches_overflow(o, &e, reprobe, perf)) continue;
        int inc = orc_inc_new(o, pos, basekey, reprobe + perf, 1);
        return inc != 0;
    }
    o->failed = 1;
    return 0;
}

/* ---------- public API ---------------------------------------------------- */

orc_t *orc_create(int k, int l, int s, uint64_t seed) {
    /* TSXHashMap.h:91-94: 2k must exceed l */
    if (k < 1 || k > 127 || l < 1 || l > 40 || s < 1 || s > 32 || 2 * k <= l) return NULL;
    if (2 * k + s > ORC_LIMBS * 64 - 8) return NULL;
    orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
    o->k = k; o->l = l; o->s = s; o->n = 2 * k; o->kv = 2 * k + s; o->wk = (2 * k + 63) / 64;
    o->slots = 1ULL << l; o->max_reprobes = o->slots - 1;
    o->table_bytes = (o->slots * (uint64_t)o->kv + 7) / 8 + 64;
    o->table = (uint8_t *)calloc(o->table_bytes, 1);
    o->starts = (uint8_t *)calloc((o->slots + 7) / 8, 1);
    if (!o->table || !o->starts) { free(o->table); free(o->starts); free(o); return NULL; }
    orc_make_hash(o, seed);
    return o;
}
void orc_destroy(orc_t *o) {
    if (!o) return;
    free(o->table); free(o->starts); free(o->rows); free(o->irows); free(o);
}
int orc_key_limbs(const orc_t *o) { return o->wk; }

static big_t load_kmer(const orc_t *o, const uint64_t *limbs) {
    big_t x = big_zero(); for (int i = 0; i < o->wk; ++i) x.w[i] = limbs[i];
    return big_trunc(x, o->n);
}

void orc_hash_rows(const orc_t *o, uint64_t *out) { /* n rows x wk limbs */
    for (int i = 0; i < o->n; ++i) for (int t = 0; t < o->wk; ++t) out[(size_t)i * o->wk + t] = o->rows[i].w[t];
}
void orc_hash_apply(const orc_t *o, const uint64_t *in, uint64_t *out) {
    big_t x = load_kmer(o, in); big_t h = orc_apply_rows(o, o->rows, &x);
    for (int t = 0; t < o->wk; ++t) out[t] = h.w[t];
}
void orc_hash_invert(const orc_t *o, const uint64_t *in, uint64_t *out) {
    big_t x = load_kmer(o, in); big_t h = orc_apply_rows(o, o->irows, &x);
    for (int t = 0; t < o->wk; ++t) out[t] = h.w[t];
}

/* code of one sequence byte: reference A=0 C=1 G=2 T=3 (SequenceUtils.h:98-125) */
static inline unsigned orc_code(unsigned char b) { return ((b >> 1) ^ (b >> 2)) & 3u; }

/* fromSequence (SequenceUtils.h:86-160): base i -> bits 2i (low) and 2i+1 */
void orc_encode(const char *seq, int k, uint64_t *out) {
    int wk = (2 * k + 63) / 64; memset(out, 0, (size_t)wk * 8);
    for (int i = 0; i < k; ++i) {
        uint64_t c = orc_code((unsigned char)seq[i]);
        out[(2 * i) >> 6] |= c << ((2 * i) & 63);
    }
}

/* addKmer (TSXHashMapPerf.h:56-205).  Returns 1 inserted, 0 table exhausted. */
int orc_add_kmer(orc_t *o, const uint64_t *kmer_limbs) {
    big_t kmer = load_kmer(o, kmer_limbs);
    big_t basekey = orc_apply_rows(o, o->rows, &kmer);
    uint64_t reprobes = 1;
    while (reprobes < o->max_reprobes) {
        uint64_t pos = orc_position(o, &basekey, reprobes);
        big_t e = orc_get(o, pos);
        if (big_is_zero(&e)) {
            big_t kvnew = big_shl(orc_make_key(o, &basekey, reprobes), o->s);
            kvnew.w[0] |= 1;
            orc_put(o, pos, big_trunc(kvnew, o->kv));
            orc_set_start(o, pos);
            o->used += 1; o->adds += 1;
            return 1;
        }
        if (orc_is_start(o, pos) && orc_matches_key(o, &e, &basekey, reprobes)) {
            int st = orc_inc_new(o, pos, &basekey, reprobes, 0);
            if (st == 2 && !orc_handle_overflow(o, &basekey, reprobes)) return 0;
            o->adds += 1;
            return 1;
        }
        ++reprobes;
    }
    o->failed = 1;
    return 0;
}

/* findOverflowCounts (TSXHashMap.h:951-1039); returns the overflow counter value */
static big_t orc_find_overflow(const orc_t *o, const big_t *basekey, uint64_t reprobe, int *bits_out) {
    uint64_t perf = 0; big_t ret = big_zero(); int req = 0;
    while (perf < o->max_reprobes) {
        perf += 1;
        uint64_t pos = orc_position(o, basekey, reprobe + perf);
        big_t e = orc_get(o, pos);
        if (big_is_zero(&e)) break;
        if (!orc_matches_overflow(o, &e, reprobe, perf)) continue;
        /* getFuncValFromKeyVal (TSXHashMap.h:1269-1279): (func << s) | value */
        int fb = o->n - o->l;
        big_t pv = big_or(big_shl(big_shr(e, o->l + o->s), o->s), big_trunc(e, o->s));
        pv = big_trunc(pv, fb + o->s);
        if (req + fb + o->s <= ORC_LIMBS * 64) ret = big_or(big_shl(pv, req), ret);
        req += fb + o->s;
        reprobe += perf; perf = 0;
    }
    *bits_out = req;
    return ret;
}

/* getKmerCount(kmer) (TSXHashMap.h:548-638); counts above 2^64-1 saturate */
uint64_t orc_get_count(const orc_t *o, const uint64_t *kmer_limbs) {
    big_t kmer = load_kmer(o, kmer_limbs);
    big_t basekey = orc_apply_rows(o, o->rows, &kmer);
    uint64_t reprobes = 1;
    while (reprobes < o->max_reprobes) {
        uint64_t pos = orc_position(o, &basekey, reprobes);
        big_t e = orc_get(o, pos);
        if (big_is_zero(&e)) return 0;
        if (orc_matches_key(o, &e, &basekey, reprobes)) {
            big_t res = big_trunc(e, o->s);
            int ob = 0; big_t ov = orc_find_overflow(o, &basekey, reprobes, &ob);
            if (ob > 0) res = big_or(big_shl(ov, o->s), res);
            for (int i = 1; i < ORC_LIMBS; ++i) if (res.w[i]) return ~0ULL;
            return res.w[0];
        }
        ++reprobes;
    }
    return 0;
}

/* getKmerCount() (TSXHashMap.h:645-648): number of k-mer start slots */
uint64_t orc_distinct(const orc_t *o) {
    uint64_t c = 0, nb = (o->slots + 7) / 8;
    for (uint64_t i = 0; i < nb; ++i) c += (uint64_t)__builtin_popcount(o->starts[i]);
    return c;
}
uint64_t orc_adds(const orc_t *o) { return o->adds; }
uint64_t orc_used_slots(const orc_t *o) { return o->used; }
int orc_failed(const orc_t *o) { return o->failed; }

/* getAllKmers (TSXHashMap.h:660-722) plus the count of each: walks the start
 * bitset, rebuilds the hashed key from (func | position - reprobe(i)) and
 * inverts the mapping.  Returns the number of k-mers written (<= cap).      */
uint64_t orc_dump(const orc_t *o, uint64_t *kmers_out, uint64_t *counts_out, uint64_t cap) {
    uint64_t w = 0;
    for (uint64_t pos = 0; pos < o->slots && w < cap; ++pos) {
        if (!orc_is_start(o, pos)) continue;
        big_t e = orc_get(o, pos);
        big_t key = big_trunc(big_shr(e, o->s), o->n);
        uint64_t reprobe = big_trunc(key, o->l).w[0];
        uint32_t i = (uint32_t)reprobe; uint32_t j = i * (i + 1) / 2;
        uint64_t low = (pos - ((uint64_t)j & (o->slots - 1))) & (o->slots - 1);
        big_t hk = big_or(big_shl(big_shr(key, o->l), o->l), big_from_u64(low));
        big_t kmer = orc_apply_rows(o, o->irows, &hk);
        for (int t = 0; t < o->wk; ++t) kmers_out[w * o->wk + t] = kmer.w[t];
        counts_out[w] = orc_get_count(o, &kmers_out[w * o->wk]);
        ++w;
    }
    return w;
}

/* FASTQ scan + createKMers + addKmer (FastXReader.h:359-372, 71-77;
 * testExecution.h:15-36; main.cpp:161-202).  Lines are split on '\n', empty
 * lines are dropped, every group of 4 remaining lines is one record whose
 * second line is the sequence.  Returns the number of k-mers added, or -1
 * when the table ran out of room.                                            */
int64_t orc_count_fastx(orc_t *o, const char *buf, uint64_t n, int lines_per_record);
int64_t orc_count_fastq(orc_t *o, const char *buf, uint64_t n) { return orc_count_fastx(o, buf, n, 4); }
/* lines_per_record: 4 = FASTQEntry (FastXReader.h:62-95), 2 = FASTAEntry (FastXReader.h:97-116: lines[0] is the
 * read id, lines[1] the sequence -- one line, FASTXreader does not join wrapped sequences).                     */
int64_t orc_count_fastx(orc_t *o, const char *buf, uint64_t n, int lines_per_record) {
    uint64_t line_no = 0, i = 0; int64_t added = 0;
    uint64_t kmer[ORC_LIMBS];
    while (i < n) {
        uint64_t e = i; while (e < n && buf[e] != '\n') ++e;
        uint64_t len = e - i;
        if (len > 0) {
            if ((line_no % (uint64_t)lines_per_record) == 1 && len >= (uint64_t)o->k) {
                for (uint64_t p = 0; p + o->k <= len; ++p) {
                    orc_encode(buf + i + p, o->k, kmer);
                    if (!orc_add_kmer(o, kmer)) return -1;
                    ++added;
                }
            }
            ++line_no;
        }
        i = e + 1;
    }
    return added;
}
