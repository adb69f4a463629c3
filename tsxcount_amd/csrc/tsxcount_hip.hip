// tsxcount_hip.hip -- C ABI (include/tsxcount_hip.h) over the HIP kernels.
// Host side only: layout derivation, the bijective mapping and its lookup
// tables, launches, staging copies.  No CPU counting path exists here.
#include "../../include/tsxcount_hip.h"
#include "tsx_kernels.h"
#include "tsx_partition.h"
#include "tsx_minimizer.h"
#include "tsx_inflate.h"

#include <mutex>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

using namespace tsx;

static thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);            \
            return (_e == hipErrorOutOfMemory) ? TSX_HIP_ENOMEM : TSX_HIP_EHIP;          \
        }                                                                                \
    } while (0)

static int scan_wg_per_cu() {  // walk_log_kernel workgroups per CU
    static int v = 0;
    if (!v) { v = 5; if (const char *e = getenv("TSX_HIP_SCAN_WGS")) v = std::min(16, std::max(1, atoi(e))); }
    return v;
}
#define SCAN_WG_PER_CU scan_wg_per_cu()
static const uint32_t OVQ_CAP = 2048;  // keys per overflow queue (one queue per level-2 workgroup: 32 MiB at 2048 workgroups)
static const size_t STAGE_PIECE_DEFAULT = (size_t)64 << 20;  // bytes of FASTQ per host piece

struct PartPlan;
struct tsx_hip_map;
static void drop_sh_plan(tsx_hip_map *m);
struct tsx_hip_map {
    TableParams p{};
    tsx_hip_layout lay{};
    int device = 0;
    uint64_t seed = 0;
    hipStream_t stream = nullptr;
    // bijective mapping, host copy: rows[i] yields output bit n-1-i
    std::vector<uint64_t> rows, irows;   // n x wk
    std::vector<uint64_t> lut, ilut;     // [groups][1<<g][wk]
    uint64_t *d_lut = nullptr, *d_ilut = nullptr, *d_roll = nullptr;
    uint64_t *d_ovq = nullptr;           // overflow queues of the level-2 partition (OVQ_CAP records per workgroup)
    uint64_t *d_def_rec = nullptr, *d_def_cnt = nullptr;   // deferred list (DeferList, tsx_device.h)
    unsigned long long *d_def_n = nullptr;
    size_t def_cap = 0;
    bool fresh = false;                  // tsx_hip_clear() was called and the table itself has not been zeroed yet:
                                         // the next partitioned build writes every segment, anything else zeroes first
    uint32_t *d_ovq_cnt = nullptr;
    size_t ovq_queues = 0;
    uint64_t roll[64] = {0};             // one-limb keys: sliding-window hash update table
    std::vector<uint64_t> roll_wide;     // multi-limb keys: the same, key_limbs words per entry
    // FASTQ scratch
    uint32_t *d_tile = nullptr; uint64_t tile_cap = 0;
    uint32_t *d_carry = nullptr;
    unsigned long long *d_seg = nullptr;  // 64 owner counters / cursors
    // host staging
    uint8_t *h_stage[2] = {nullptr, nullptr};
    uint8_t *d_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};   // kernels that read d_stage[i] have finished
    hipEvent_t stage_in[2] = {nullptr, nullptr};     // the H2D copy into d_stage[i] has finished
    hipStream_t copy_stream = nullptr;               // H2D copies of the host entry point
    size_t stage_bytes = 0;
    size_t piece = STAGE_PIECE_DEFAULT;  // TSX_HIP_PIECE_BYTES overrides (tests exercise piece seams)
    bool piece_fixed = false;
    int cus = 256;
    // partitioned insert path (tsx_partition.h): grow-only scratch
    int path = 0;                    // 0 auto, 1 atomic, 2 partitioned (tsx_hip_set_path / TSX_HIP_PATH)
    uint64_t *d_buf[2] = {nullptr, nullptr};
    size_t buf_bytes[2] = {0, 0};
    unsigned long long *d_cnt = nullptr;   // [log regions | level-1 lists | segment lists]
    size_t cnt_entries = 0;
    // optional per-pass timing (HIP events on the launch stream)
    uint64_t *d_small = nullptr;     // scratch of tsx_hip_get_counts_host / tsx_hip_lookup_host for a few k-mers
    bool attr_done = false;          // dynamic-LDS limits of the partition / build kernels set on this map's device
    int timing = 0;
    int dbg = 0;                     // TSX_HIP_DEBUG: bit0 = skip the global insert (ablation builds only)
    std::vector<hipEvent_t> ev;      // seven per piece: before pass 1, before pass 3, after pass 3, start of the
                                     // partition phase (later than the scan's end only in a sharded run: the
                                     // exchange lies between), after level 1, level 2, build
    std::vector<unsigned long long> h_regions;   // host copy of the region table of a sharded build (starts, then sizes)
    PartPlan *sh_pl = nullptr;                   // sharded run, level 1 per exchange window: the plan made at window 0,
    uint32_t sh_rw = 0, sh_windows = 0;          // regions per window, windows of the step,
    uint32_t mz_regions = 0; uint64_t mz_dcap = 0; size_t mz_len = 0;   // minimizer exchange: the described text waiting in buffer 1 (regions x capacity; its bytes)
    bool sh_ev3 = false;                         // stage timing: the walks of a description exchange have recorded event 3
    unsigned long long *d_desc_cnt = nullptr;    // strips described per wave of strip_desc_kernel (key log form)
    size_t desc_cnt_entries = 0;
    uint64_t *sh_buf1 = nullptr;                 // and its own sub-list buffer and counters (the scans of the later
    size_t sh_buf1_bytes = 0;                    // windows plan with -- and clear -- the map's while level 1 of the
    unsigned long long *sh_cnt = nullptr;        // earlier ones has already left its sizes there)
    size_t sh_cnt_entries = 0;
    std::deque<long> ev_open;        // tuples of shard scans whose partition phase has not run yet (oldest first)
    size_t ev_used = 0;
    // Ordering between the map's own stream and a caller's stream (the `stream` argument of the *_device entry
    // points): tsx_hip_clear works on the map's stream and records clear_ev behind it; every entry point that
    // launches on a caller's stream waits for that event first.  The other way round, the last caller's stream
    // is remembered (`foreign`) and tsx_hip_clear / tsx_hip_sync order themselves behind what was queued there.
    uint8_t *d_slabdesc = nullptr;      // count_slabs: the descriptions of every text window, then one counter per window
    size_t slabdesc_bytes = 0;
    hipEvent_t clear_ev = nullptr, join_ev = nullptr;
    bool clear_ev_set = false;
    hipStream_t foreign = nullptr;
};

static const size_t STAGE_PAD = 256;
static const int EV_N = 8;   // timing events per piece: before pass 1, before the scan, after it, start of the partition
                             // phase, after level 1, after level 2, after the build kernel, after the inserts behind it
static bool can_partition(const tsx_hip_map *m);
static int clear_impl(tsx_hip_map *m, bool full);
static int ensure_zeroed(tsx_hip_map *m, hipStream_t st);
static inline void join_foreign(tsx_hip_map *m, bool host_wait);

extern "C" int tsx_hip_key_limbs(int k) { return (k < 1 || k > 127) ? TSX_HIP_EINVAL : (2 * k + 63) / 64; }

extern "C" const char *tsx_hip_strerror(int code) {
    switch (code) {
        case TSX_HIP_OK: return "ok";
        case TSX_HIP_EINVAL: return "Invalid lengths for hashmap size and value of k";
        case TSX_HIP_ENODEVICE: return "no HIP device";
        case TSX_HIP_ENOMEM: return "device memory exhausted";
        case TSX_HIP_EHIP: return "HIP runtime error";
        case TSX_HIP_EFULL: return "Could not insert kmer: table full";
        case TSX_HIP_EOVERFLOW: return "count overflow array full";
        case TSX_HIP_ERANGE: return "output buffer too small";
        case TSX_HIP_ELOCK: return "a multi-limb slot stayed locked past the spin bound";
    }
    return "unknown";
}
extern "C" const char *tsx_hip_last_error(void) { return g_last_error.c_str(); }

extern "C" int tsx_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static inline unsigned code_of(unsigned char b) { return ((b >> 1) ^ (b >> 2)) & 3u; }

extern "C" int tsx_hip_encode(const char *seq, int k, uint64_t *out) {
    if (!seq || !out || k < 1 || k > 127) return TSX_HIP_EINVAL;
    const int wk = (2 * k + 63) / 64;
    memset(out, 0, (size_t)wk * 8);
    for (int i = 0; i < k; ++i) out[(2 * i) >> 6] |= (uint64_t)code_of((unsigned char)seq[i]) << ((2 * i) & 63);
    return TSX_HIP_OK;
}
extern "C" int tsx_hip_decode(const uint64_t *limbs, int k, char *out) {
    if (!limbs || !out || k < 1 || k > 127) return TSX_HIP_EINVAL;
    for (int i = 0; i < k; ++i) out[i] = "ACGT"[(limbs[(2 * i) >> 6] >> ((2 * i) & 63)) & 3];
    out[k] = 0;
    return TSX_HIP_OK;
}

// ---- bijective GF(2) mapping ------------------------------------------------
// The reference draws a random UNIT UPPER TRIANGULAR matrix over GF(2)
// (BijectiveKMapping::getRandomMatrix, BijectiveKMapping.h:284-303; row i carries bit
// n-1-j = M[i][j] (matrixToRows, :227-256) and yields output bit n-1-i (applyto,
// :202-225)).  With that family output bit p depends only on input bits <= p, so the
// slot index (the low l bits) is a function of the first l/2 bases alone: all k-mers
// that share a 15-base prefix share one home slot AND one probe sequence.  The
// reference survives that with up to 2^l reprobes; an 8-bit reprobe field does not
// (AT-rich reads at load 0.48 already exhausted 255 probes, scripts/skew_check.py).
// So the matrix here is not triangular (make_mapping() below: multiplication by a random
// field element for one-limb keys, L * U for longer ones): still bijective and
// GF(2)-linear (same IBijectiveFunction contract, same LUT evaluation), but every
// key bit reaches the slot index.  Counts do not depend on the matrix.  Seeded
// splitmix64 replaces srand(time(NULL)).
static uint64_t splitmix_next(uint64_t &st) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline int rbit(const uint64_t *r, int i) { return (int)((r[i >> 6] >> (i & 63)) & 1); }
static inline void rset(uint64_t *r, int i) { r[i >> 6] |= 1ULL << (i & 63); }

// Irreducible polynomials over GF(2) of degree 2, 4, ..., 64 (low terms; the leading term is
// implied): trinomials where one exists, else pentanomials.  Found with Rabin's test; a wrong
// entry would make the one-limb mapping singular, which make_mapping() reports.
static const uint64_t GF_POLY_LOW[33] = {0, 0x3ULL, 0x3ULL, 0x3ULL, 0x87ULL, 0x9ULL, 0x9ULL, 0x21ULL, 0x47ULL, 0x9ULL, 0x9ULL,
    0x3ULL, 0x87ULL, 0x47ULL, 0x3ULL, 0x3ULL, 0x813ULL, 0x81ULL, 0x201ULL, 0x87ULL, 0x20BULL, 0x81ULL, 0x21ULL, 0x3ULL,
    0x823ULL, 0x207ULL, 0x9ULL, 0x201ULL, 0x8403ULL, 0x80001ULL, 0x3ULL, 0x20000001ULL, 0x807ULL};

static inline uint64_t gf_mulz(uint64_t a, uint64_t plow, int n) {  // a * z in GF(2^n)
    const uint64_t top = (a >> (n - 1)) & 1ULL;
    a <<= 1;
    if (n < 64) a &= (1ULL << n) - 1ULL;
    return top ? (a ^ plow) : a;
}
static inline uint64_t gf_mul(uint64_t a, uint64_t b, uint64_t plow, int n) {
    uint64_t r = 0;
    for (int i = 0; i < n; ++i) {
        if ((b >> i) & 1ULL) r ^= a;
        a = gf_mulz(a, plow, n);
    }
    return r;
}

// Degrees 66..254 (multi-limb keys): x^n + x^a [+ x^b + x^c] + 1, found by scripts/find_irreducible.py
// (Rabin's test); the rows for degrees <= 64 are not used (GF_POLY_LOW above keeps the one-limb mapping
// what it was).  A wrong entry would make the mapping singular, which make_mapping() reports.
static const uint8_t GF_POLY_EXP[127][3] = {  // degree 2, 4, ..., 254: x^n + x^a [+ x^b + x^c] + 1
    {1, 0, 0}, {1, 0, 0}, {1, 0, 0}, {4, 3, 1}, {3, 0, 0}, {3, 0, 0}, {5, 0, 0}, {5, 3, 1},
    {3, 0, 0}, {3, 0, 0}, {1, 0, 0}, {4, 3, 1}, {4, 3, 1}, {1, 0, 0}, {1, 0, 0}, {7, 3, 2},
    {7, 0, 0}, {9, 0, 0}, {6, 5, 1}, {5, 4, 3}, {7, 0, 0}, {5, 0, 0}, {1, 0, 0}, {5, 3, 2},
    {4, 3, 2}, {3, 0, 0}, {9, 0, 0}, {7, 4, 2}, {19, 0, 0}, {1, 0, 0}, {29, 0, 0}, {4, 3, 1},
    {3, 0, 0}, {9, 0, 0}, {5, 3, 1}, {10, 9, 3}, {35, 0, 0}, {21, 0, 0}, {6, 5, 3}, {9, 4, 2},
    {8, 3, 1}, {5, 0, 0}, {21, 0, 0}, {7, 6, 2}, {27, 0, 0}, {21, 0, 0}, {21, 0, 0}, {10, 9, 6},
    {11, 0, 0}, {15, 0, 0}, {29, 0, 0}, {4, 3, 1}, {15, 0, 0}, {17, 0, 0}, {33, 0, 0}, {5, 4, 3},
    {5, 3, 2}, {4, 2, 1}, {33, 0, 0}, {4, 3, 1}, {6, 2, 1}, {19, 0, 0}, {21, 0, 0}, {7, 2, 1},
    {3, 0, 0}, {17, 0, 0}, {57, 0, 0}, {5, 3, 2}, {8, 7, 1}, {15, 0, 0}, {21, 0, 0}, {7, 4, 2},
    {71, 0, 0}, {27, 0, 0}, {53, 0, 0}, {6, 3, 2}, {15, 0, 0}, {9, 0, 0}, {8, 6, 5}, {5, 3, 2},
    {27, 0, 0}, {10, 8, 7}, {37, 0, 0}, {15, 3, 2}, {11, 0, 0}, {1, 0, 0}, {13, 0, 0}, {11, 3, 2},
    {31, 0, 0}, {3, 0, 0}, {81, 0, 0}, {9, 8, 7}, {11, 0, 0}, {6, 5, 2}, {8, 7, 6}, {7, 2, 1},
    {87, 0, 0}, {3, 0, 0}, {9, 0, 0}, {5, 3, 2}, {55, 0, 0}, {27, 0, 0}, {10, 9, 5}, {9, 3, 1},
    {7, 0, 0}, {105, 0, 0}, {73, 0, 0}, {7, 3, 1}, {11, 0, 0}, {7, 0, 0}, {5, 4, 2}, {9, 8, 3},
    {10, 7, 3}, {113, 0, 0}, {8, 7, 6}, {9, 4, 2}, {31, 0, 0}, {5, 0, 0}, {73, 0, 0}, {8, 5, 3},
    {95, 0, 0}, {111, 0, 0}, {11, 2, 1}, {15, 14, 10}, {103, 0, 0}, {15, 0, 0}, {7, 2, 1},
};

struct Big { uint64_t w[4]; };   // an element of GF(2^n), n <= 254: bit i of the polynomial in w[i / 64]
static inline Big big_zero() { Big b; b.w[0] = b.w[1] = b.w[2] = b.w[3] = 0; return b; }
static inline int big_bit(const Big &a, int i) { return (int)((a.w[i >> 6] >> (i & 63)) & 1); }
static inline void big_xor(Big &a, const Big &b) { for (int t = 0; t < 4; ++t) a.w[t] ^= b.w[t]; }
static inline Big big_shr2(const Big &a) {
    Big r;
    for (int t = 0; t < 4; ++t) r.w[t] = (a.w[t] >> 2) | (t < 3 ? a.w[t + 1] << 62 : 0);
    return r;
}
static inline Big big_mulz(const Big &a, const Big &plow, int n) {  // a * z in GF(2^n)
    const int top = big_bit(a, n - 1);
    Big r;
    for (int t = 3; t >= 0; --t) r.w[t] = (a.w[t] << 1) | (t > 0 ? a.w[t - 1] >> 63 : 0);
    for (int i = n; i < 256; ++i) r.w[i >> 6] &= ~(1ULL << (i & 63));
    if (top) big_xor(r, plow);
    return r;
}
static inline Big big_mul(Big a, const Big &b, const Big &plow, int n) {
    Big r = big_zero();
    for (int i = 0; i < n; ++i) {
        if (big_bit(b, i)) big_xor(r, a);
        a = big_mulz(a, plow, n);
    }
    return r;
}

// The bijective k-mer mapping (IBijectiveFunction / BijectiveKMapping in the reference: a random
// invertible GF(2) matrix): M = multiplication by a random element c of GF(2^2k), a universal
// family whose matrix is dense like a random one -- and because the k-mer window slides by one
// base (x' = (x >> 2) | b << (2k-2), i.e. x' = (x - low2) / z^2 + b z^(2k-2) as polynomials),
//     c*x' = (c*x + c*low2) * z^-2 + c*b*z^(2k-2),
// the scan kernels get the hash of the next window from the current one with one lookup in the
// 64-entry table roll[(h & 3) | out << 2 | in << 4] built here (key_limbs words per entry).
static int make_mapping(tsx_hip_map *m) {
    const int n = m->p.n, wk = m->p.wk;
    m->rows.assign((size_t)n * wk, 0);
    m->irows.assign((size_t)n * wk, 0);
    uint64_t st = m->seed;
    if (wk == 1) {
        const uint64_t plow = GF_POLY_LOW[n / 2], mask = (n < 64) ? ((1ULL << n) - 1ULL) : ~0ULL;
        uint64_t c = 0;
        while (c == 0 || c == 1) c = splitmix_next(st) & mask;       // 0 is not invertible, 1 is the identity
        // column j of M is c * z^j; rows[i] yields output bit n-1-i
        uint64_t col = c;
        for (int j = 0; j < n; ++j) {
            for (int r = 0; r < n; ++r)
                if ((col >> r) & 1ULL) rset(&m->rows[(size_t)(n - 1 - r)], j);
            col = gf_mulz(col, plow, n);
        }
        const uint64_t zinv = (plow >> 1) | (1ULL << (n - 1));       // z * zinv = 1 (P has constant term 1)
        const uint64_t zinv2 = gf_mul(zinv, zinv, plow, n);
        uint64_t A[4], C[4], E[4];
        for (int v = 0; v < 4; ++v) {
            A[v] = gf_mul(c, (uint64_t)v, plow, n);
            C[v] = A[v];
            for (int t = 0; t < n - 2; ++t) C[v] = gf_mulz(C[v], plow, n);
            E[v] = ((v & 1) ? zinv2 : 0ULL) ^ ((v & 2) ? zinv : 0ULL);
        }
        for (int idx = 0; idx < 64; ++idx) {
            const int hb = idx & 3, out = (idx >> 2) & 3, in = (idx >> 4) & 3;
            m->roll[idx] = (A[out] >> 2) ^ E[(hb ^ (int)(A[out] & 3ULL)) & 3] ^ C[in];
        }
    } else {
        const uint8_t *e = GF_POLY_EXP[n / 2 - 1];
        Big plow = big_zero();
        plow.w[0] = 1;
        for (int t = 0; t < 3; ++t) if (e[t]) plow.w[e[t] >> 6] |= 1ULL << (e[t] & 63);
        Big c = big_zero();
        bool trivial = true;
        while (trivial) {   // 0 is not invertible, 1 is the identity
            for (int t = 0; t < 4; ++t) c.w[t] = splitmix_next(st);
            for (int i = n; i < 256; ++i) c.w[i >> 6] &= ~(1ULL << (i & 63));
            trivial = (c.w[0] <= 1 && !c.w[1] && !c.w[2] && !c.w[3]);
        }
        // column j of M is c * z^j; rows[i] yields output bit n-1-i
        Big col = c;
        for (int j = 0; j < n; ++j) {
            for (int r = 0; r < n; ++r)
                if (big_bit(col, r)) rset(&m->rows[(size_t)(n - 1 - r) * wk], j);
            col = big_mulz(col, plow, n);
        }
        Big zinv = big_zero();                                         // z * zinv = 1 (P has constant term 1)
        for (int t = 0; t < 4; ++t) zinv.w[t] = (plow.w[t] >> 1) | (t < 3 ? plow.w[t + 1] << 63 : 0);
        zinv.w[(n - 1) >> 6] |= 1ULL << ((n - 1) & 63);
        const Big zinv2 = big_mul(zinv, zinv, plow, n);
        Big A[4], C[4], E[4];
        for (int v = 0; v < 4; ++v) {
            Big bv = big_zero(); bv.w[0] = (uint64_t)v;
            A[v] = big_mul(c, bv, plow, n);
            C[v] = A[v];
            for (int t = 0; t < n - 2; ++t) C[v] = big_mulz(C[v], plow, n);
            E[v] = big_zero();
            if (v & 1) big_xor(E[v], zinv2);
            if (v & 2) big_xor(E[v], zinv);
        }
        m->roll_wide.assign((size_t)64 * wk, 0);
        for (int idx = 0; idx < 64; ++idx) {
            const int hb = idx & 3, out = (idx >> 2) & 3, in = (idx >> 4) & 3;
            Big r = big_shr2(A[out]);
            big_xor(r, E[(hb ^ (int)(A[out].w[0] & 3ULL)) & 3]);
            big_xor(r, C[in]);
            for (int t = 0; t < wk; ++t) m->roll_wide[(size_t)idx * wk + t] = r.w[t];
        }
    }
    // Inverse by Gauss-Jordan on [A | I], A[r][c] = coefficient of input bit c in output bit r
    // (output bit r is produced by rows[n-1-r]).
    std::vector<uint64_t> A((size_t)n * wk), I((size_t)n * wk, 0);
    for (int r = 0; r < n; ++r) {
        memcpy(&A[(size_t)r * wk], &m->rows[(size_t)(n - 1 - r) * wk], (size_t)wk * 8);
        rset(&I[(size_t)r * wk], r);
    }
    for (int c = 0; c < n; ++c) {
        int piv = -1;
        for (int r = c; r < n; ++r) if (rbit(&A[(size_t)r * wk], c)) { piv = r; break; }
        if (piv < 0) return TSX_HIP_EINVAL;  // singular: cannot happen for c != 0 with P irreducible
        if (piv != c) for (int t = 0; t < wk; ++t) { std::swap(A[(size_t)piv * wk + t], A[(size_t)c * wk + t]); std::swap(I[(size_t)piv * wk + t], I[(size_t)c * wk + t]); }
        for (int r = 0; r < n; ++r)
            if (r != c && rbit(&A[(size_t)r * wk], c))
                for (int t = 0; t < wk; ++t) { A[(size_t)r * wk + t] ^= A[(size_t)c * wk + t]; I[(size_t)r * wk + t] ^= I[(size_t)c * wk + t]; }
    }
    // input bit c = XOR of the output bits in I[c]; irows[i] yields original bit n-1-i
    for (int c = 0; c < n; ++c) memcpy(&m->irows[(size_t)(n - 1 - c) * wk], &I[(size_t)c * wk], (size_t)wk * 8);
    return TSX_HIP_OK;
}

static void apply_rows(const tsx_hip_map *m, const std::vector<uint64_t> &rows, const uint64_t *x, uint64_t *out) {
    const int n = m->p.n, wk = m->p.wk;
    memset(out, 0, (size_t)wk * 8);
    for (int i = 0; i < n; ++i) {
        uint64_t acc = 0;
        for (int t = 0; t < wk; ++t) acc ^= rows[(size_t)i * wk + t] & x[t];
        if (__builtin_parityll(acc)) rset(out, n - 1 - i);
    }
}

// LUT[group][v] = A * (v << (g*group)): the mapping is linear, so A*x is the
// XOR of one table entry per g-bit group of x.
static void make_lut(const tsx_hip_map *m, const std::vector<uint64_t> &rows, std::vector<uint64_t> &lut) {
    const int wk = m->p.wk, g = m->p.g, groups = m->p.groups;
    lut.assign((size_t)groups * (1u << g) * wk, 0);
    std::vector<uint64_t> x(wk), y(wk);
    for (int grp = 0; grp < groups; ++grp)
        for (unsigned v = 0; v < (1u << g); ++v) {
            std::fill(x.begin(), x.end(), 0);
            const int bit = grp * g;
            x[bit >> 6] = (uint64_t)v << (bit & 63);
            x[wk - 1] &= m->p.top_mask;
            apply_rows(m, rows, x.data(), y.data());
            memcpy(&lut[((size_t)grp * (1u << g) + v) * wk], y.data(), (size_t)wk * 8);
        }
}

// ---- layout -------------------------------------------------------------------
static int derive_layout(tsx_hip_map *m, int k, int l, int s, int overflow_l, int shard_bits, int shard_index) {
    if (k < 1 || k > 127 || l < 4 || l > 36 || s < 0 || s > 32) return TSX_HIP_EINVAL;
    if (shard_bits < 0 || shard_bits > 3 || shard_index < 0 || shard_index >= (1 << shard_bits)) return TSX_HIP_EINVAL;
    if (2 * k <= l + shard_bits) return TSX_HIP_EINVAL;  // TSXHashMap.h:91-94, on the whole (sharded) table
    TableParams &p = m->p;
    p.k = k; p.l = l; p.n = 2 * k; p.wk = (2 * k + 63) / 64;
    p.lg = l + shard_bits; p.shard = (uint32_t)shard_index;
    p.R = std::min(l, 8);
    p.F = 2 * k - p.lg;
    const int KB = p.R + p.F;
    int W, C;
    if (s == 0) {
        W = (KB + 5 + 63) / 64;
        C = std::min(32, 64 * W - KB - (W > 1 ? 1 : 0));
    } else {
        C = s;
        W = (KB + C + 63) / 64;
        if (W > 1) W = (KB + C + 1 + 63) / 64;
    }
    const int lock = (W > 1) ? 1 : 0;
    p.K0 = 64 - C - lock;
    if (W > 4 || p.K0 < p.R || p.K0 > 63) return TSX_HIP_EINVAL;
    // all spilled func bits must fit limbs 1..W-1
    if (KB - p.K0 > 64 * (W - 1)) return TSX_HIP_EINVAL;
    p.W = W; p.C = C; p.cshift = 64 - C;
    p.k0mask = (p.K0 >= 64) ? ~0ULL : ((1ULL << p.K0) - 1ULL);
    p.lock_bit = lock ? (1ULL << p.K0) : 0ULL;
    p.slot_mask = (1ULL << l) - 1ULL;
    // a segment fits a CU's LDS: 128 KiB = 2^14 one-limb slots or 2^12 four-limb slots (96 KiB of three-limb slots).  Two-limb
    // slots: 2^12 = 64 KiB, so that TWO build workgroups of 512 threads share a CU and one sweeps while the other inserts
    // (k = 63: build 13.3 -> 11.8 ms; 2^13-slot segments when that would need more than 2^18 of them).  One-limb slots at
    // 64 KiB: build 4.41 -> 3.91 ms, but a radix level of 512 lists costs 0.5 ms more: 11.97 against 12.05 ms per step, not taken.
    const int smax = (W == 1) ? 14 : (W == 2) ? 13 : 12;
    p.S = std::min(l, (W == 2 && l - 12 <= 18) ? 12 : smax);
    if (const char *e = getenv("TSX_HIP_SEG_BITS")) p.S = std::min(l, std::min(smax, std::max(8, atoi(e))));
    p.seg_mask = (1ULL << p.S) - 1ULL;
    const uint64_t maxr = (1ULL << p.R) - 1ULL;
    p.max_reprobes = (uint32_t)std::min<uint64_t>(maxr, p.slot_mask);
    p.top_mask = (p.n & 63) ? ((1ULL << (p.n & 63)) - 1ULL) : ~0ULL;
    p.line_mask = 3;                 // FASTQ records (tsx_hip_set_record_lines)
    // LUT granularity: bytes when the table stays <= 32 KiB of LDS, nibbles otherwise
    const size_t lut8 = (size_t)((p.n + 7) / 8) * 256 * p.wk * 8;
    p.g = (lut8 <= (32u << 10)) ? 8 : 4;
    p.groups = (p.n + p.g - 1) / p.g;
    // secondary array: one entry per slot whose counter overflowed.  With the automatic (wide) counters that
    // is a handful of hot k-mers: 2^(l-8) entries; with explicit narrow --s counters many keys carry: 2^(l-4).
    // (It is cleared with the table whenever something carried: 1 GiB at l = 30 cost 0.23 ms per clear.)
    int ol = overflow_l ? overflow_l : std::max(10, (s == 0 && C >= 16) ? l - 8 : l - 4);
    if (ol < 4 || ol > 34) return TSX_HIP_EINVAL;
    p.sec_mask = (1ULL << ol) - 1ULL;
    tsx_hip_layout &L = m->lay;
    L.shard_bits = shard_bits; L.shard_index = shard_index;
    L.k = k; L.l = l; L.key_limbs = p.wk; L.entry_limbs = W; L.func_bits = p.F; L.reprobe_bits = p.R;
    L.count_bits = C; L.overflow_l = ol; L.max_reprobes = p.max_reprobes; L.slots = 1ULL << l;
    L.table_bytes = L.slots * (uint64_t)W * 8ULL;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_create(tsx_hip_map **out, int k, int l, int storagebits, int overflow_l,
                              uint64_t hash_seed, int device) {
    return tsx_hip_create_shard(out, k, l, storagebits, overflow_l, hash_seed, device, 0, 0);
}

extern "C" int tsx_hip_create_shard(tsx_hip_map **out, int k, int l, int storagebits, int overflow_l,
                                    uint64_t hash_seed, int device, int shard_bits, int shard_index) {
    if (!out) return TSX_HIP_EINVAL;
    *out = nullptr;
    tsx_hip_map *m = new tsx_hip_map();
    int rc = derive_layout(m, k, l, storagebits, overflow_l, shard_bits, shard_index);
    if (rc != TSX_HIP_OK) { delete m; return rc; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        delete m;
        g_last_error = "no HIP device (the HIP path has no CPU fallback)";
        return TSX_HIP_ENODEVICE;
    }
    m->device = device; m->seed = hash_seed;
    if (const char *e = getenv("TSX_HIP_DEBUG")) m->dbg = atoi(e);
    if (const char *e = getenv("TSX_HIP_PATH")) m->path = atoi(e);
    if (const char *e = getenv("TSX_HIP_PIECE_BYTES")) {
        const long long v = atoll(e);
        if (v >= 256) { m->piece = ((size_t)v + 15) & ~(size_t)15; m->piece_fixed = true; }
    }
    auto fail = [&](int code) { tsx_hip_destroy(m); return code; };
#define HIP_TRY_C(expr)                                                         \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) {                                                 \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);   \
            return fail(_e == hipErrorOutOfMemory ? TSX_HIP_ENOMEM : TSX_HIP_EHIP); \
        }                                                                       \
    } while (0)
    // Host pieces large enough for the partitioned path to pay off (text >= table / 32),
    // between 64 MiB and 1 GiB of pinned staging per buffer.
    if (!m->piece_fixed) {
        const size_t want = (size_t)(m->lay.table_bytes / 16);
        m->piece = std::min<size_t>((size_t)1 << 30, std::max<size_t>(STAGE_PIECE_DEFAULT, want)) & ~(size_t)4095;
    }
    HIP_TRY_C(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY_C(hipGetDeviceProperties(&prop, device));
    m->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY_C(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    HIP_TRY_C(hipEventCreateWithFlags(&m->clear_ev, hipEventDisableTiming));
    HIP_TRY_C(hipEventCreateWithFlags(&m->join_ev, hipEventDisableTiming));
    TableParams &p = m->p;
    HIP_TRY_C(hipMalloc((void **)&p.table, m->lay.table_bytes));
    HIP_TRY_C(hipMalloc((void **)&p.sec_keys, (p.sec_mask + 1) * 8));
    HIP_TRY_C(hipMalloc((void **)&p.sec_cnt, (p.sec_mask + 1) * 8));
    HIP_TRY_C(hipMalloc((void **)&p.stats, ST_N * sizeof(unsigned long long)));
    HIP_TRY_C(hipMalloc((void **)&p.seg_dirty, (size_t)(m->lay.slots >> p.S)));
    HIP_TRY_C(hipMalloc((void **)&m->d_carry, 64));
    HIP_TRY_C(hipMalloc((void **)&m->d_seg, 64 * sizeof(unsigned long long)));
    rc = make_mapping(m);
    if (rc != TSX_HIP_OK) return fail(rc);
    make_lut(m, m->rows, m->lut);
    make_lut(m, m->irows, m->ilut);
    HIP_TRY_C(hipMalloc((void **)&m->d_lut, m->lut.size() * 8));
    HIP_TRY_C(hipMalloc((void **)&m->d_ilut, m->ilut.size() * 8));
    HIP_TRY_C(hipMemcpy(m->d_lut, m->lut.data(), m->lut.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY_C(hipMemcpy(m->d_ilut, m->ilut.data(), m->ilut.size() * 8, hipMemcpyHostToDevice));
    p.lut = m->d_lut; p.ilut = m->d_ilut; p.roll = nullptr;
    {
        const void *src = (p.wk == 1) ? (const void *)m->roll : (const void *)m->roll_wide.data();
        const size_t bytes = (size_t)64 * p.wk * 8;
        // one-limb keys: the same mapping as a LUT by 4-bit groups (16 x 16 entries = 2 KiB) behind the roll
        // table -- walk_part_kernel has no LDS to spare for the 8-bit-group LUT (16 KiB)
        std::vector<uint64_t> lut4;
        if (p.wk == 1) {
            lut4.assign(256, 0);
            for (int grp = 0; grp < 16; ++grp)
                for (uint64_t v = 0; v < 16; ++v) {
                    uint64_t x = (v << (4 * grp)) & p.top_mask, y = 0;
                    apply_rows(m, m->rows, &x, &y);
                    lut4[grp * 16 + v] = y;
                }
        }
        HIP_TRY_C(hipMalloc((void **)&m->d_roll, bytes + lut4.size() * 8));
        HIP_TRY_C(hipMemcpy(m->d_roll, src, bytes, hipMemcpyHostToDevice));
        if (!lut4.empty())
            HIP_TRY_C(hipMemcpy(m->d_roll + 64, lut4.data(), lut4.size() * 8, hipMemcpyHostToDevice));
        p.roll = m->d_roll;
    }
    rc = clear_impl(m, true);
    if (rc != TSX_HIP_OK) return fail(rc);
    *out = m;
    return TSX_HIP_OK;
#undef HIP_TRY_C
}

extern "C" void tsx_hip_destroy(tsx_hip_map *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    (void)hipFree(m->p.table); (void)hipFree(m->p.sec_keys); (void)hipFree(m->p.sec_cnt);
    drop_sh_plan(m);
    (void)hipFree(m->sh_buf1); (void)hipFree(m->sh_cnt); (void)hipFree(m->d_desc_cnt); (void)hipFree(m->d_slabdesc);
    (void)hipFree(m->p.stats); (void)hipFree(m->d_lut); (void)hipFree(m->d_ilut); (void)hipFree(m->d_roll);
    (void)hipFree(m->d_ovq); (void)hipFree(m->d_ovq_cnt); (void)hipFree(m->d_small);
    (void)hipFree(m->d_def_rec); (void)hipFree(m->d_def_cnt); (void)hipFree(m->d_def_n);
    (void)hipFree(m->d_tile); (void)hipFree(m->d_carry); (void)hipFree(m->d_seg);
    (void)hipFree(m->p.seg_dirty); (void)hipFree(m->d_buf[0]); (void)hipFree(m->d_buf[1]); (void)hipFree(m->d_cnt);
    for (int i = 0; i < 2; ++i) {
        if (m->h_stage[i]) (void)hipHostFree(m->h_stage[i]);
        if (m->d_stage[i]) (void)hipFree(m->d_stage[i]);
        if (m->stage_done[i]) (void)hipEventDestroy(m->stage_done[i]);
        if (m->stage_in[i]) (void)hipEventDestroy(m->stage_in[i]);
    }
    if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
    for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
    if (m->clear_ev) (void)hipEventDestroy(m->clear_ev);
    if (m->join_ev) (void)hipEventDestroy(m->join_ev);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int tsx_hip_get_layout(const tsx_hip_map *m, tsx_hip_layout *out) {
    if (!m || !out) return TSX_HIP_EINVAL;
    *out = m->lay;
    return TSX_HIP_OK;
}

// Zero the secondary array only when something ever carried into it (it is 1/16 of the table's slots).
__global__ __launch_bounds__(NT) void sec_clear_kernel(TableParams p, int force) {
    if (!force && p.stats[ST_CARRY] == 0) return;
    const uint64_t n2 = (p.sec_mask + 1) / 2;   // uint4 = two slots
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * NT) {
        reinterpret_cast<uint4 *>(p.sec_keys)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4 *>(p.sec_cnt)[i] = make_uint4(0, 0, 0, 0);
    }
}

// full: zero everything now (creation).  Otherwise the table itself is only MARKED clear when the next
// partitioned build can take care of it (m->fresh): that build writes every segment exactly once -- built,
// or zeroed when it has no keys -- and every other entry point zeroes the table first (ensure_zeroed).
static int clear_impl(tsx_hip_map *m, bool full) {
    HIP_TRY(hipSetDevice(m->device));
    join_foreign(m, false);
    if (full || !can_partition(m)) {
        HIP_TRY(hipMemsetAsync(m->p.table, 0, m->lay.table_bytes, m->stream));
        m->fresh = false;
    } else {
        m->fresh = true;
    }
    hipLaunchKernelGGL(sec_clear_kernel, dim3(m->cus * 4), dim3(NT), 0, m->stream, m->p, full ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(m->p.stats, 0, ST_N * sizeof(unsigned long long), m->stream));
    HIP_TRY(hipMemsetAsync(m->p.seg_dirty, 0, (size_t)(m->lay.slots >> m->p.S), m->stream));
    HIP_TRY(hipEventRecord(m->clear_ev, m->stream));
    m->clear_ev_set = true;
    return TSX_HIP_OK;
}

// The table was cleared lazily and the caller is not a partitioned build: zero it now.
static int ensure_zeroed(tsx_hip_map *m, hipStream_t st) {
    if (!m->fresh) return TSX_HIP_OK;
    HIP_TRY(hipMemsetAsync(m->p.table, 0, m->lay.table_bytes, st));
    m->fresh = false;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_clear(tsx_hip_map *m) {
    if (!m) return TSX_HIP_EINVAL;
    return clear_impl(m, false);
}

static int read_stats(tsx_hip_map *m, unsigned long long *st) {
    HIP_TRY(hipMemcpyAsync(st, m->p.stats, ST_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_sync(tsx_hip_map *m) {
    if (!m) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    join_foreign(m, true);
    unsigned long long st[ST_N];
    int rc = read_stats(m, st);
    if (rc != TSX_HIP_OK) return rc;
    if (st[ST_FAIL]) return TSX_HIP_EFULL;
    if (st[ST_SECFAIL]) return TSX_HIP_EOVERFLOW;
    if (st[ST_LOCKTO]) return TSX_HIP_ELOCK;   // an expired spin re-probes: the key may sit in two slots
    return TSX_HIP_OK;
}

static inline hipStream_t pick_stream(tsx_hip_map *m, void *stream) {
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    if (st != m->stream) {
        if (m->clear_ev_set) (void)hipStreamWaitEvent(st, m->clear_ev, 0);   // behind the last tsx_hip_clear
        m->foreign = st;
    }
    return st;
}
// The map's own stream is about to touch what a caller's stream may still be working on: order it behind.
static inline void join_foreign(tsx_hip_map *m, bool host_wait) {
    if (!m->foreign) return;
    if (host_wait) {
        (void)hipStreamSynchronize(m->foreign);
        m->foreign = nullptr;
    } else if (m->join_ev && hipEventRecord(m->join_ev, m->foreign) == hipSuccess) {
        (void)hipStreamWaitEvent(m->stream, m->join_ev, 0);
    }
}
static inline int grid_for(const tsx_hip_map *m, uint64_t work_items, int per_cu) {
    uint64_t blocks = (work_items + NT - 1) / NT;
    uint64_t cap = (uint64_t)m->cus * per_cu;
    return (int)std::max<uint64_t>(1, std::min(blocks, cap));
}

#define DISPATCH_WK(m, CALL)                         \
    switch ((m)->p.wk) {                             \
        case 1: { constexpr int WKV = 1; CALL; } break; \
        case 2: { constexpr int WKV = 2; CALL; } break; \
        case 3: { constexpr int WKV = 3; CALL; } break; \
        default: { constexpr int WKV = 4; CALL; } break; \
    }

// ---- partitioned path: plan, scratch, launches ------------------------------------
struct PartPlan {
    int g;                       // regions of the key log (= scan waves, or cuts of a received array)
    int rw;                      // 64-bit words per record
    uint64_t log_cap;            // records per log region
    int b1, b2;                  // radix bits of level 1 / level 2 (b2 == 0: one level)
    uint32_t nb1, nb2, nseg, cpr2;
    uint64_t cap_sub;            // records per level-2 sub-list
    uint32_t hist_nb;            // bins of the scan-side histogram (nb1, or #owners for a sharded scan)
    unsigned long long *c_log, *c_rstart, *c_bstart, *c_bcnt, *c_seg, *d_offs;
    uint32_t *d_hist;
    size_t cnt_need;
    // walk fused with level 1 (walk_part_kernel): G1 workgroups, each with a sub-list of cap1 records per level-1
    // bucket in buffer 1 (list (b, g) at (b * G1 + g) * cap1), sizes in c_l1[b * G1 + g]
    bool fused;
    uint32_t G1;
    uint64_t cap1;
    unsigned long long *c_l1;
    uint64_t *buf1;              // buffer 1 of this plan (the map's, or the window-wise sharded level 1's own)
};

static void drop_sh_plan(tsx_hip_map *m) { delete m->sh_pl; m->sh_pl = nullptr; }

static inline int rec_words(int wk) { return wk == 3 ? 4 : wk; }

// Two radix levels of at most 512 lists reach 2^18 segments (2^32 one-limb slots).  A larger table of one-limb keys and
// slots is built SLAB BY SLAB: a slab is the 2^18 segments that share the top slab_bits() bits of the home slot -- exactly
// what a shard of a multi-GPU table is to its GPU -- and count_slabs() walks the strip descriptions once per slab, keeping
// the slab's keys (the owner-filtered walk of the sharded path), then runs level 2 and the build for that slab.
static int max_seg_bits() {   // TSX_HIP_SLAB_SEGBITS: tests build small tables slab by slab
    static int v = 0;
    if (!v) { v = 18; if (const char *e = getenv("TSX_HIP_SLAB_SEGBITS")) v = std::min(18, std::max(9, atoi(e))); }
    return v;
}
static int slab_bits(const tsx_hip_map *m) {
    const TableParams &p = m->p;
    const int nsegbits = p.l - p.S;
    if (nsegbits <= max_seg_bits() || p.wk != 1 || p.W != 1 || p.lg != p.l) return 0;
    return nsegbits - max_seg_bits();
}
static bool can_partition(const tsx_hip_map *m) {
    const TableParams &p = m->p;
    const int nsegbits = p.l - p.S;
    if (nsegbits >= 1 && nsegbits <= (p.lg == p.l && p.wk == 1 && p.W == 1 ? max_seg_bits() : 18)) return true;  // two levels of <= 512 lists
    const int sb = slab_bits(m);
    return sb >= 1 && sb <= 4;
}
// workgroups of level 2 per level-1 bucket (plan_partition's cpr2) for the map's geometry
static uint32_t level2_cpr(const tsx_hip_map *m) {
    const int nsegbits = m->p.l - m->p.S;
    const int b1 = std::min(9, (nsegbits <= 8) ? nsegbits : (nsegbits + 1) / 2);
    if (nsegbits - b1 <= 0) return 1;
    uint32_t c = std::min<uint32_t>(8, std::max<uint32_t>(1, (uint32_t)(m->cus * 8) / (1u << b1)));
    while (c & (c - 1)) c &= c - 1;
    return c;
}

template <typename T>
static int grow(hipStream_t st, T *&ptr, size_t &have, size_t need) {
    if (need <= have) return TSX_HIP_OK;
    HIP_TRY(hipStreamSynchronize(st));
    if (ptr) HIP_TRY(hipFree(ptr));
    ptr = nullptr; have = 0;
    HIP_TRY(hipMalloc((void **)&ptr, need + need / 8 + 4096));
    have = need + need / 8 + 4096;
    return TSX_HIP_OK;
}

// The deferred list of the map (local runs): room for every record of the pass in the worst case (skewed
// input whose keys all spill); only what is appended is ever touched.
static int ensure_deferred(tsx_hip_map *m, uint64_t maxrec, hipStream_t st) {
    const int rw = rec_words(m->p.wk);
    if (!m->d_def_n) HIP_TRY(hipMalloc((void **)&m->d_def_n, 64));
    // (passes of billions of records: room for 2^28 of them -- what lands here is hot keys with their totals and the
    // spill of skewed lists; beyond the capacity records count as insert failures, reported by tsx_hip_sync)
    if (maxrec > ((uint64_t)1 << 30)) maxrec = std::max<uint64_t>((uint64_t)1 << 28, maxrec / 16);
    if (maxrec > m->def_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (m->d_def_rec) HIP_TRY(hipFree(m->d_def_rec));
        if (m->d_def_cnt) HIP_TRY(hipFree(m->d_def_cnt));
        m->d_def_rec = m->d_def_cnt = nullptr; m->def_cap = 0;
        const size_t cap = maxrec + maxrec / 8 + 4096;
        HIP_TRY(hipMalloc((void **)&m->d_def_rec, cap * rw * 8));
        HIP_TRY(hipMalloc((void **)&m->d_def_cnt, cap * 8));
        m->def_cap = cap;
    }
    return TSX_HIP_OK;
}

// maxrec: upper bound of records; g: number of source regions; own_log: the records come from
// this map's scan kernel (needs the log buffer); hist_nb_override: sharded scan.
static int plan_partition(tsx_hip_map *m, uint64_t maxrec, int g, bool own_log, uint32_t hist_nb_override,
                          hipStream_t st, PartPlan &pl, int fused_g = 0) {
    const TableParams &p = m->p;
    const int nsegbits = p.l - p.S;
    pl.g = g;
    pl.rw = rec_words(p.wk);
    pl.nseg = 1u << nsegbits;
    // fan-out per level is capped at 512 (histogram of the scan kernel and ring staging live in LDS)
    pl.b1 = std::min(9, (nsegbits <= 8) ? nsegbits : (nsegbits + 1) / 2);
    if (const char *e = getenv("TSX_HIP_B1")) {   // experiments: the split between the two levels
        const int b = atoi(e);
        if (b >= 1 && b <= 9 && nsegbits - b >= 1 && nsegbits - b <= 9) pl.b1 = b;
    }
    pl.b2 = nsegbits - pl.b1;
    pl.nb1 = 1u << pl.b1; pl.nb2 = 1u << pl.b2;
    pl.hist_nb = hist_nb_override ? hist_nb_override : pl.nb1;
    auto even = [](uint64_t v) { return (v + 1) & ~1ULL; };
    pl.log_cap = even(maxrec / g + maxrec / g / 3 + 2048);
    if (const char *e = getenv("TSX_HIP_LOG_CAP")) pl.log_cap = even(std::max(16, atoi(e)));  // tests: force region overflow
    // level 2 runs cpr2 workgroups per level-1 bucket; each owns one sub-list per segment
    pl.cpr2 = pl.b2 ? (uint32_t)std::min<uint32_t>(8, std::max<uint32_t>(1, (uint32_t)(m->cus * 8) / pl.nb1)) : 1;
    while (pl.cpr2 & (pl.cpr2 - 1)) pl.cpr2 &= pl.cpr2 - 1;   // a power of two: the build gives every sub-list 16/cpr2 waves
    if (pl.b2) if (const char *e = getenv("TSX_HIP_CPR2")) pl.cpr2 = (uint32_t)std::min(8, std::max(1, atoi(e)));
    const uint64_t per_sub = maxrec / pl.nseg / pl.cpr2;
    // multiple of 16 records: every sub-list starts on a 128-B line
    pl.cap_sub = (per_sub + per_sub / 4 + 6 * (uint64_t)std::sqrt((double)per_sub + 1.0) + 64 + 15) & ~15ULL;
    // fused scan + level 1: one-limb keys, two levels (a one-level split feeds the build, which takes <= 8 pieces)
    pl.fused = fused_g > 0 && pl.b2 > 0 && p.wk == 1 && (fused_g + pl.cpr2 - 1) / pl.cpr2 <= (uint32_t)PART_MAX_PIECES;
    pl.G1 = pl.fused ? (uint32_t)fused_g : 0;
    pl.cap1 = 0;
    if (pl.fused) {
        const uint64_t per1 = maxrec / pl.nb1 / pl.G1;
        pl.cap1 = (per1 + per1 / 4 + 6 * (uint64_t)std::sqrt((double)per1 + 1.0) + 64 + 15) & ~15ULL;
        if (const char *e = getenv("TSX_HIP_CAP1")) pl.cap1 = (uint64_t)std::max(16, atoi(e) & ~15);   // tests: force overflow
    }
    // buffer 0: key log, later the segment sub-lists of a two-level split; buffer 1: packed level-1 output, or the
    // level-1 sub-lists of the fused scan
    const uint64_t rec_cap = pl.fused ? 0 : (own_log ? (uint64_t)g * pl.log_cap : maxrec);
    const size_t need0 = std::max<uint64_t>(own_log ? rec_cap : 0, pl.b2 ? (uint64_t)pl.nseg * pl.cpr2 * pl.cap_sub : 0) * 8 * pl.rw;
    const size_t need1 = (pl.fused ? (uint64_t)pl.nb1 * pl.G1 * pl.cap1 : rec_cap) * 8 * pl.rw;
    int rc = grow(st, m->d_buf[0], m->buf_bytes[0], need0);
    if (rc != TSX_HIP_OK) return rc;
    rc = grow(st, m->d_buf[1], m->buf_bytes[1], need1);
    if (rc != TSX_HIP_OK) return rc;
    pl.buf1 = m->d_buf[1];
    // counters: [region fill | region start | bucket start | bucket size | sub-list size], then the
    // histogram matrix (u32) and its exclusive scan (u64), both max(nb1, hist_nb) x g
    const uint32_t hb = std::max(pl.nb1, pl.hist_nb);
    pl.cnt_need = 2 * (size_t)g + 2 * (size_t)hb + (size_t)pl.nseg * pl.cpr2 + (size_t)pl.nb1 * pl.G1;
    const size_t mat = (size_t)hb * g;
    size_t have = m->cnt_entries;
    unsigned long long *ptr = m->d_cnt;
    rc = grow(st, ptr, have, pl.cnt_need * 8 + mat * 12 + 64);
    m->d_cnt = ptr; m->cnt_entries = have;
    if (rc != TSX_HIP_OK) return rc;
    pl.c_log = m->d_cnt;                    // fill of each log region, or size of each cut of a received array
    pl.c_rstart = pl.c_log + g;
    pl.c_bstart = pl.c_rstart + g; pl.c_bcnt = pl.c_bstart + hb; pl.c_seg = pl.c_bcnt + hb;
    pl.c_l1 = pl.c_seg + (size_t)pl.nseg * pl.cpr2;
    pl.d_offs = pl.c_l1 + (size_t)pl.nb1 * pl.G1;
    pl.d_hist = reinterpret_cast<uint32_t *>(pl.d_offs + mat);
    HIP_TRY(hipMemsetAsync(m->d_cnt, 0, pl.cnt_need * 8, st));
    if (!m->attr_done) {   // per map, hence per device: the attribute belongs to the device's copy of the kernel
        const int big = 150 << 10, seg = 128 << 10;
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<1, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<1, RING_NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<2, RING_NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<4, RING_NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<1, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<2, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<4, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<2, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)partition_ring_kernel<4, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)build_segments_stream_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, seg + (32 << 10)));
        HIP_TRY(hipFuncSetAttribute((const void *)build_segments_stream_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, seg + (32 << 10)));
        HIP_TRY(hipFuncSetAttribute((const void *)build_segments_stream_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, seg + (32 << 10)));
        HIP_TRY(hipFuncSetAttribute((const void *)build_segments_wide_stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, seg + (32 << 10)));
        HIP_TRY(hipFuncSetAttribute((const void *)build_segments_wide_stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, seg + (32 << 10)));
        HIP_TRY(hipFuncSetAttribute((const void *)build_segments_wide_stream_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, seg + (32 << 10)));
        HIP_TRY(hipFuncSetAttribute((const void *)build_segments_wide_stream_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, seg + (32 << 10)));
        HIP_TRY(hipFuncSetAttribute((const void *)walk_part_kernel<SP_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)walk_part_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute((const void *)walk_log_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
        HIP_TRY(hipFuncSetAttribute((const void *)walk_log_wide_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
        m->attr_done = true;
    }
    return TSX_HIP_OK;
}

// Overflow queues (OVQ_CAP records each): one per level-2 workgroup, plus one per workgroup of the fused scan.
static int ensure_ovq(tsx_hip_map *m, size_t nq, int rw, hipStream_t st) {
    if (nq * rw <= m->ovq_queues) return TSX_HIP_OK;
    HIP_TRY(hipStreamSynchronize(st));
    if (m->d_ovq) HIP_TRY(hipFree(m->d_ovq));
    if (m->d_ovq_cnt) HIP_TRY(hipFree(m->d_ovq_cnt));
    m->d_ovq = nullptr; m->d_ovq_cnt = nullptr; m->ovq_queues = 0;
    HIP_TRY(hipMalloc((void **)&m->d_ovq, nq * OVQ_CAP * 8 * rw));
    HIP_TRY(hipMalloc((void **)&m->d_ovq_cnt, nq * 4 + 512 * 4 + 64));   // (+ the skew flags of level 2, one per bucket, behind the counters)
    m->ovq_queues = nq * rw;
    return TSX_HIP_OK;
}

// Launch with the record width as a compile-time constant.
#define DISPATCH_RW(rw, CALL)                        \
    switch (rw) {                                    \
        case 1: { constexpr int RWV = 1; CALL; } break; \
        case 2: { constexpr int RWV = 2; CALL; } break; \
        default: { constexpr int RWV = 4; CALL; } break; \
    }

// Radix level 1 (+ level 2), the segment build, then everything that waited for the build (overflow
// queues, deferred list).  Source records: g regions of `src`, either src_cap apart with fills c_log (a
// key log) or at region_start/c_log (cuts of a packed array); pl.d_hist must hold their level-1
// histogram.  The caller has made room in the map's deferred list (ensure_deferred) and reset its
// counter before the first kernel that may append to it.
static int run_partition_build(tsx_hip_map *m, const PartPlan &pl, const uint64_t *src,
                               const unsigned long long *region_start, uint64_t src_cap, hipStream_t st,
                               hipEvent_t *ev = nullptr) {
    TableParams pp = m->p;
    const TableParams &p = m->p;
    pp.defer = DeferList{m->d_def_rec, m->d_def_cnt, m->d_def_n, (uint64_t)m->def_cap};
    const int rw = pl.rw;
    // ring depth in words: PART_FLUSH-1 words may stay behind a flush, plus one batch of arrivals (mean = batch / nb)
    auto ring_bits = [](uint32_t nb) {
        const uint32_t mean = std::max<uint32_t>(1, RING_NT * PART_WPT / nb);
        uint32_t bits = 4;
        while ((1u << bits) < PART_FLUSH + 2 * mean && bits < 6) ++bits;
        if (const char *e = getenv("TSX_HIP_RING_BITS")) bits = (uint32_t)std::min(6, std::max(4, atoi(e)));   // experiments
        return bits;
    };
    auto part_lds = [](uint32_t nb, uint32_t bits) { return (size_t)nb * (((size_t)8 << bits) + 36); };
    if (!pl.fused) {
    hipLaunchKernelGGL(offsets_rows_kernel, dim3(pl.nb1), dim3(1024), 0, st, (const uint32_t *)pl.d_hist, pl.d_offs,
                       (uint32_t)pl.g, pl.c_bcnt);
    hipLaunchKernelGGL(offsets_finish_kernel, dim3(1), dim3(1024), 0, st, pl.nb1, pl.c_bstart, pl.c_bcnt);
    {   // level 1: every region -> packed array ordered by the top b1 bits of the home slot
        const uint32_t bits = ring_bits(pl.nb1);
#define TSX_LEVEL1(RWV, NTV)                                                                                                   \
        hipLaunchKernelGGL((partition_ring_kernel<RWV, NTV>), dim3(pl.g), dim3(NTV), part_lds(pl.nb1, bits), st,                 \
                           pp, src, region_start, (const unsigned long long *)pl.c_log, src_cap, (uint32_t)pl.g, 1u,             \
                           pl.buf1, (const unsigned long long *)pl.d_offs, (const unsigned long long *)pl.c_bstart,              \
                           (unsigned long long *)nullptr, (uint64_t)0, pl.nb1, (uint32_t)(p.l - pl.b1), bits, m->dbg,            \
                           (uint64_t *)nullptr, (uint32_t *)nullptr, 0u, (const unsigned long long *)nullptr, 0u, (uint64_t)0,   \
                           0, (unsigned long long *)nullptr, 0u, 0u, 0, 0u, (const uint32_t *)nullptr)
        // (512 lists: the rings leave room for one workgroup per CU -- 1024 threads then, 16 waves either way)
        if (part_lds(pl.nb1, bits) > ((size_t)80 << 10)) { DISPATCH_RW(rw, TSX_LEVEL1(RWV, 1024)); }
        else { DISPATCH_RW(rw, TSX_LEVEL1(RWV, RING_NT)); }
#undef TSX_LEVEL1
        HIP_TRY(hipGetLastError());
    }
    }   // (fused: walk_part_kernel has left the level-1 sub-lists in buffer 1)
    if (ev) HIP_TRY(hipEventRecord(ev[4], st));
    const uint64_t *lists = pl.buf1;
    const unsigned long long *lists_start = pl.c_bstart, *lists_cnt = pl.c_bcnt;
    uint64_t lists_cap = 0;
    uint32_t pieces = 1;
    uint32_t nq2 = 0;
    // level 2 leaves PRE-FORMATTED records (format_record, tsx_partition.h) where the stream build of one-limb keys and
    // slots reads them and the fields fit: slot image in bits [0, R + F), first probe position above (TSX_HIP_BUILD_PRE=0: raw keys)
    static int pre_ok = -1;
    if (pre_ok < 0) { const char *e = getenv("TSX_HIP_BUILD_PRE"); pre_ok = e ? atoi(e) : 1; }
    const int pre = (pre_ok && pl.b2 && p.wk == 1 && p.W == 1 && !m->dbg && p.R + p.F >= 32 && p.R + p.F + p.S <= 64) ? 1 : 0;
    if (pl.b2) {  // level 2: cpr2 workgroups per level-1 bucket, each with its own sub-list per segment
        const uint32_t bits = ring_bits(pl.nb2);
        nq2 = pl.nb1 * pl.cpr2;   // one overflow queue per workgroup (the fused scan's queues follow them)
        // records per private chunk of the deferred list (mass spills of skewed input, see partition_ring_kernel): a
        // quarter of the list shared out over the workgroups, a power of two in 64 .. 4096; 0: the list is too small
        uint32_t dch = 0;
        for (uint32_t c = 4096; c >= 64 && !dch; c >>= 1)
            if ((uint64_t)c * nq2 * 4 <= (uint64_t)m->def_cap) dch = c;
        {
            const int rco = ensure_ovq(m, (size_t)nq2 + pl.G1, rw, st);
            if (rco != TSX_HIP_OK) return rco;
        }
        // Which form of the level-2 kernel works (partition_ring_kernel: SKEW) is decided on the device: skew_probe_kernel
        // samples every bucket and raises the flag -- the last word of the overflow-queue counters -- when it finds a hot
        // key; both forms are launched, per bucket one of them returns at once.  TSX_HIP_SKEW=0|1 forces the plain / the skew form.
        static int skew_force = -2;
        if (skew_force == -2) { const char *e = getenv("TSX_HIP_SKEW"); skew_force = e ? atoi(e) : -1; }
        uint32_t *d_skew = m->d_ovq_cnt + m->ovq_queues / rw;   // (ensure_ovq keeps one spare counter behind the queues')
        if (skew_force >= 0) {
            HIP_TRY(hipMemsetAsync(d_skew, skew_force ? 1 : 0, (size_t)pl.nb1 * 4, st));   // (any non-zero word means "skewed")
        } else {
            DISPATCH_RW(rw, hipLaunchKernelGGL((skew_probe_kernel<RWV>), dim3(pl.nb1), dim3(256), 0, st, (const uint64_t *)pl.buf1,
                                               (const unsigned long long *)pl.c_bstart, (const unsigned long long *)pl.c_bcnt,
                                               (const unsigned long long *)(pl.fused ? pl.c_l1 : nullptr), pl.G1, pl.cap1, d_skew));
        }
        // (512 lists of one-word records: the rings leave room for one workgroup per CU -- 1024 threads then)
#define TSX_LEVEL2(RWV, NTV, SK, BITS)                                                                                          \
        hipLaunchKernelGGL((partition_ring_kernel<RWV, NTV, SK>), dim3(pl.nb1 * pl.cpr2), dim3(NTV), part_lds(pl.nb2, BITS), st,  \
                           pp, (const uint64_t *)pl.buf1, (const unsigned long long *)pl.c_bstart,                              \
                           (const unsigned long long *)pl.c_bcnt, (uint64_t)0, pl.nb1, pl.cpr2, m->d_buf[0],                    \
                           (const unsigned long long *)nullptr, (const unsigned long long *)nullptr, pl.c_seg, pl.cap_sub,      \
                           pl.nb2, (uint32_t)p.S, BITS, m->dbg, m->d_ovq, m->d_ovq_cnt, OVQ_CAP,                                 \
                           (const unsigned long long *)(pl.fused ? pl.c_l1 : nullptr), pl.G1, pl.cap1, 0,                       \
                           (unsigned long long *)nullptr, 0u, 0u, pre, dch, (const uint32_t *)d_skew)
        if (part_lds(pl.nb2, bits) > ((size_t)80 << 10)) {
            DISPATCH_RW(rw, TSX_LEVEL2(RWV, 1024, false, bits); TSX_LEVEL2(RWV, 1024, true, bits));
        } else {
            DISPATCH_RW(rw, TSX_LEVEL2(RWV, RING_NT, false, bits); TSX_LEVEL2(RWV, RING_NT, true, bits));
        }
#undef TSX_LEVEL2
        HIP_TRY(hipGetLastError());
        lists = m->d_buf[0]; lists_start = nullptr; lists_cnt = pl.c_seg; lists_cap = pl.cap_sub; pieces = pl.cpr2;
    }
    if (ev) HIP_TRY(hipEventRecord(ev[5], st));
    const int fresh = m->fresh ? 1 : 0;
    if (!(m->dbg & 64)) {  // ablation: bit 6 skips the build (partition timing experiments)
        const int gb = (int)std::min<uint32_t>(pl.nseg, (uint32_t)m->cus * 16);
        const size_t seg_bytes = ((size_t)8 << p.S) * p.W;
        static int build_la = -1;   // TSX_HIP_BUILD_LOOKAHEAD=0|1: the tail's look-ahead over the next probe positions
        if (build_la < 0) { const char *e = getenv("TSX_HIP_BUILD_LOOKAHEAD"); build_la = e ? atoi(e) : 1; }
        if (p.wk == 1 && p.W == 1) {
            if (m->dbg)   // the instance with the ablation / diagnostic switches compiled in
                hipLaunchKernelGGL((build_segments_stream_kernel<true, false>), dim3(gb), dim3(1024), seg_bytes + (32 << 10), st, pp,
                                   lists, lists_start, lists_cnt, lists_cap, pieces, pl.nseg, m->dbg, fresh, build_la);
            else {
                // TSX_HIP_BUILD_SNT: threads per workgroup of the stream build (1024; 512 with TSX_HIP_SEG_BITS=13 puts two
                // workgroups on a CU: 64 KiB segment + 16 KiB of rings each)
                int snt = 1024;
                if (const char *e = getenv("TSX_HIP_BUILD_SNT")) snt = std::min(1024, std::max(64 * (int)pieces, atoi(e) & ~63));
                if (pre)
                    hipLaunchKernelGGL((build_segments_stream_kernel<false, true>), dim3(gb), dim3(snt), seg_bytes + (size_t)(snt / 64) * 2048, st, pp,
                                       lists, lists_start, lists_cnt, lists_cap, pieces, pl.nseg, 0, fresh, build_la);
                else
                    hipLaunchKernelGGL((build_segments_stream_kernel<false, false>), dim3(gb), dim3(snt), seg_bytes + (size_t)(snt / 64) * 2048, st, pp,
                                       lists, lists_start, lists_cnt, lists_cap, pieces, pl.nseg, 0, fresh, build_la);
            }
        } else {   // multi-limb keys and / or slots: wave streams too (64 records of LDS per wave behind the segment)
            // a segment of <= 64 KiB (TSX_HIP_SEG_BITS): 512 threads, two workgroups per CU -- one sweeps while the other inserts
            int wnt = (seg_bytes <= ((size_t)64 << 10)) ? 512 : 1024;
            if (const char *e = getenv("TSX_HIP_BUILD_WNT")) wnt = std::min(1024, std::max(64 * (int)pieces, atoi(e) & ~63));
            const size_t ring_bytes = (size_t)(wnt / 64) * 64 * 8 * rw;
            DISPATCH_WK(m, hipLaunchKernelGGL((build_segments_wide_stream_kernel<WKV>), dim3(gb), dim3(wnt),
                                              seg_bytes + ring_bytes, st, pp, lists, lists_start, lists_cnt, lists_cap, pieces,
                                              pl.nseg, fresh));
        }
        HIP_TRY(hipGetLastError());
        m->fresh = false;   // every segment has been written: built, or zeroed
    }
    if (ev) HIP_TRY(hipEventRecord(ev[6], st));
    // Records that found their sub-list filled up by a hot key, and the deferred list: inserted now, by the
    // whole chip, into a table whose segments are all in place.
    if (nq2 && !(m->dbg & 1)) {
        const uint32_t nq = nq2 + pl.G1;
        DISPATCH_WK(m, hipLaunchKernelGGL((overflow_insert_kernel<WKV>), dim3(std::min<uint32_t>(nq, (uint32_t)m->cus * 8)),
                                          dim3(PART_NT), 0, st, pp, (const uint64_t *)m->d_ovq,
                                          (const uint32_t *)m->d_ovq_cnt, OVQ_CAP, nq));
        HIP_TRY(hipGetLastError());
    }
    if (!(m->dbg & 1)) {
        DISPATCH_WK(m, hipLaunchKernelGGL((deferred_insert_kernel<WKV>), dim3(m->cus * 2), dim3(PART_NT), 0, st, pp,
                                          (const uint64_t *)m->d_def_rec, (const uint64_t *)m->d_def_cnt,
                                          (const unsigned long long *)m->d_def_n, (uint64_t)0, (uint64_t)m->def_cap));
        HIP_TRY(hipGetLastError());
    }
    return TSX_HIP_OK;
}

struct HotOut { uint64_t *keys = nullptr, *cnts = nullptr; uint64_t cap = 0; unsigned long long *n = nullptr; };
// Sharded scan: where the keys go.  send: groups by owner GPU (own group left out when `own` is given),
// own: this GPU's keys (they never travel), counts[o]: keys per owner, key_sum += sum of all keys.
struct ShardOut {
    uint64_t *send = nullptr; uint64_t send_cap = 0;
    uint64_t *own = nullptr; uint64_t own_cap = 0;
    unsigned long long *counts = nullptr, *key_sum = nullptr;
};

// One FASTQ piece already on the device: passes 1-3.  own_end = number of start
// positions this piece owns (bytes past it are halo for windows that begin
// before it); head_open = the piece starts in the middle of a line.
// shard_send != nullptr: sharded scan -- the keys are not built into the local table but
// split by owner into shard_send (counts per owner to shard_counts), hot keys to `hot`.
// Sharded run, description exchange: the piece is only DESCRIBED (strip_desc_kernel), the descriptions packed
// into out[0 .. *count).
// owners > 0 (minimizer exchange): one packed list per owner GPU at out + o * cap, count[0 .. owners) their lengths,
// count[owners .. owners + 4) the homopolymer k-mer occurrences taken out of the descriptions, per base.
struct DescOut { uint4 *out = nullptr; uint64_t cap = 0; unsigned long long *count = nullptr, *sum = nullptr; int long_desc = 0; int owners = 0; };
// (owners != 0: describe only -- the wave regions stay in buffer 1 for mini_split)

static int run_fastq_piece(tsx_hip_map *m, const uint8_t *d_text, uint64_t n, uint64_t own_end, int head_open,
                           hipStream_t st, ShardOut sh = ShardOut(), HotOut hot = HotOut(), DescOut dsc = DescOut()) {
    uint64_t *shard_send = sh.send;
    const uint64_t shard_cap = sh.send_cap;
    unsigned long long *shard_counts = sh.counts;
    if (own_end == 0) return TSX_HIP_OK;
    const uint64_t ntiles = (own_end + TILE - 1) / TILE;
    if (ntiles > m->tile_cap) {
        if (m->d_tile) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(m->d_tile)); m->d_tile = nullptr; }
        m->tile_cap = ntiles + ntiles / 4 + 1024;
        // tile counts, then one partial sum per SCAN_CHUNK tiles
        HIP_TRY(hipMalloc((void **)&m->d_tile, (m->tile_cap + m->tile_cap / SCAN_CHUNK + 16) * sizeof(uint32_t)));
    }
    hipEvent_t *ev = nullptr;
    if (m->timing) {
        if (m->ev_used + EV_N > m->ev.size()) {
            for (int i = 0; i < EV_N; ++i) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); m->ev.push_back(e); }
        }
        ev = &m->ev[m->ev_used]; m->ev_used += EV_N;
        HIP_TRY(hipEventRecord(ev[0], st));
    }
    const int g1 = (int)std::min<uint64_t>(ntiles, (uint64_t)m->cus * 8);
    hipLaunchKernelGGL(line_count_kernel, dim3(g1), dim3(NT), 0, st, d_text, n, own_end, head_open, m->d_tile, ntiles);
    {
        const uint64_t nchunks = (ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
        uint32_t *chunk = m->d_tile + m->tile_cap;
        hipLaunchKernelGGL(line_chunk_sum_kernel, dim3((uint32_t)nchunks), dim3(SCAN_CHUNK), 0, st,
                           (const uint32_t *)m->d_tile, ntiles, chunk);
        hipLaunchKernelGGL(line_chunk_scan_kernel, dim3(1), dim3(1024), 0, st, chunk, nchunks, m->d_carry);
        hipLaunchKernelGGL(line_scan_kernel, dim3((uint32_t)nchunks), dim3(SCAN_CHUNK), 0, st, m->d_tile, ntiles,
                           (const uint32_t *)chunk);
    }
    if (ev) HIP_TRY(hipEventRecord(ev[1], st));
    if (dsc.out || dsc.owners) {
        const int gdd = (int)std::min<uint64_t>(ntiles, (uint64_t)m->cus * 8), gdr = gdd * (NT / 64);
        const uint32_t du = dsc.long_desc ? 2u : 1u;   // 16-byte units per description (long: four strips in 32 bytes)
        const uint64_t dcap = ((ntiles + gdd - 1) / gdd) * (dsc.long_desc ? 16 : 64);
        int rcd = grow(st, m->d_buf[1], m->buf_bytes[1], (size_t)gdr * dcap * du * 16);
        if (rcd != TSX_HIP_OK) return rcd;
        {   // region sizes | region offsets | total
            size_t have = m->desc_cnt_entries;
            rcd = grow(st, m->d_desc_cnt, have, ((size_t)2 * gdr + 16) * 8 + (size_t)MZ_MAX_RANKS * m->cus * MZ_WG_PER_CU * 4 + 64);
            m->desc_cnt_entries = have;
            if (rcd != TSX_HIP_OK) return rcd;
        }
        unsigned long long *d_cnt = m->d_desc_cnt, *d_offs = d_cnt + gdr, *d_tot = d_offs + gdr;
        if (dsc.owners) {   // homopolymers leave here already: counted in the four words behind the chunk counters
            unsigned long long *d_hom = d_cnt + 2 * (size_t)gdr + 8 + ((size_t)MZ_MAX_RANKS * m->cus * MZ_WG_PER_CU + 1) / 2;
            HIP_TRY(hipMemsetAsync(d_hom, 0, 32, st));
            hipLaunchKernelGGL(strip_desc_kernel<true>, dim3(gdd), dim3(NT), 0, st, m->p, d_text, n, own_end, head_open,
                               (const uint32_t *)m->d_tile, ntiles, (uint4 *)m->d_buf[1], dcap, d_cnt, dsc.sum, 0, d_hom);
        } else
        hipLaunchKernelGGL(strip_desc_kernel<false>, dim3(gdd), dim3(NT), 0, st, m->p, d_text, n, own_end, head_open,
                           (const uint32_t *)m->d_tile, ntiles, (uint4 *)m->d_buf[1], dcap, d_cnt, dsc.sum, dsc.long_desc);
        if (dsc.owners) {   // owner = f(minimizer): the regions stay where they are, mini_split hands them out by owner
            HIP_TRY(hipGetLastError());
            m->mz_regions = (uint32_t)gdr; m->mz_dcap = dcap;
            if (ev) {
                for (int i = 2; i < EV_N; ++i) HIP_TRY(hipEventRecord(ev[i], st));
                m->ev_open.push_back((long)(ev - m->ev.data()));
            }
            return TSX_HIP_OK;
        }
        hipLaunchKernelGGL(desc_prefix_kernel, dim3(1), dim3(1024), 0, st, (const unsigned long long *)d_cnt, (uint32_t)gdr,
                           d_offs, d_tot, dcap, (uint64_t)dsc.cap, m->p.stats);
        hipLaunchKernelGGL(desc_pack_kernel, dim3(std::min(gdr, m->cus * 8)), dim3(256), 0, st, (const uint4 *)m->d_buf[1], dcap,
                           (const unsigned long long *)d_cnt, (const unsigned long long *)d_offs, (uint32_t)gdr, dsc.out,
                           dsc.cap * du, du);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(dsc.count, d_tot, 8, hipMemcpyDeviceToDevice, st));
        if (ev) {   // the walks, level 2 and the build follow in other calls, which record 3..7 of the first window's tuple again
            for (int i = 2; i < EV_N; ++i) HIP_TRY(hipEventRecord(ev[i], st));
            m->ev_open.push_back((long)(ev - m->ev.data()));
        }
        return TSX_HIP_OK;
    }
    const size_t lut_bytes = m->lut.size() * 8;
    const int g3 = (int)std::min<uint64_t>(ntiles, (uint64_t)m->cus * 3);

    // Which insert path?  The partitioned path rewrites every touched segment
    // once (2 x table bytes at worst), the atomic path pays ~60 ps per k-mer.
    const TableParams &p = m->p;
    // (a table built slab by slab takes its partitioned path in count_slabs, over the whole text at once)
    const bool use_part = shard_send || (can_partition(m) && !slab_bits(m) &&
                                         (m->path == 2 || (m->path == 0 && own_end * 32 >= m->lay.table_bytes)));
    if (!use_part) {
        int rcz = ensure_zeroed(m, st);
        if (rcz != TSX_HIP_OK) return rcz;
        DISPATCH_WK(m, hipLaunchKernelGGL((count_fastq_kernel<WKV>), dim3(g3), dim3(NT), lut_bytes, st, m->p, d_text, n,
                                          own_end, head_open, (const uint32_t *)m->d_tile, ntiles, m->dbg));
        HIP_TRY(hipGetLastError());
        if (ev) for (int i = 2; i < EV_N; ++i) HIP_TRY(hipEventRecord(ev[i], st));
        return TSX_HIP_OK;
    }

    // FASTQ: the quality line is as long as the sequence, at most half of the bytes start a k-mer; FASTA: all may
    const uint64_t maxrec = (p.line_mask == 3 ? own_end / 2 : own_end) + 65536;
    const uint32_t nown = 1u << (p.lg - p.l);
    // the scan kernels of this path keep one log region per WAVE
    // workgroups per CU of the walk that writes the key log (it carries no tile state): 6 for two-limb keys, 3 above
    // (LUT of up to 32 KiB)
    const int scan_wgs = (p.wk == 1) ? SCAN_WG_PER_CU : (p.wk == 2 ? 6 : 3);
    const int gs = (int)std::min<uint64_t>(ntiles, (uint64_t)m->cus * scan_wgs);
    const int greg = gs * (NT / 64);
    PartPlan pl;
    // The walk fused with radix level 1 (local runs, one-limb keys, two radix levels); TSX_HIP_FUSE=0: the key log +
    // a separate level 1 (what sharded scans and one-level tables take).  Read per call: the tests run both in one process.
    const char *fuse_env = getenv("TSX_HIP_FUSE");
    const int fuse = (fuse_env && atoi(fuse_env) == 0) ? 0 : 2;
    const int g_sp = (int)std::min<uint64_t>((own_end + 8191) / 8192, (uint64_t)m->cus * 2);   // walk workgroups
    // strip_desc_kernel: 52 VGPRs, 2.4 KiB of LDS -- eight workgroups per CU (five: 3.9 ms for both kernels, eight: 3.7)
    static const int desc_wgs = getenv("TSX_HIP_DESC_WGS") ? std::min(16, std::max(1, atoi(getenv("TSX_HIP_DESC_WGS")))) : 8;
    const int gd = (int)std::min<uint64_t>(ntiles, (uint64_t)m->cus * desc_wgs), gdreg = gd * (NT / 64);
    const bool want_fuse = fuse && !shard_send && p.wk == 1;

    int rc = plan_partition(m, maxrec, (want_fuse && fuse == 2) ? std::max(greg, gdreg) : greg, true, shard_send ? nown : 0, st,
                            pl, want_fuse ? g_sp : 0);
    if (rc == TSX_HIP_OK && want_fuse && !pl.fused)   // a one-level table: the key log form, planned for its own regions
        rc = plan_partition(m, maxrec, greg, true, 0, st, pl, 0);
    if (rc != TSX_HIP_OK) return rc;
    // what cannot take the fast route: the caller's hot list (sharded scan), else the map's deferred list
    TableParams pp = m->p;
    if (shard_send) {
        pp.defer = DeferList{hot.keys, hot.cnts, hot.n, hot.cap};
    } else {
        rc = ensure_deferred(m, maxrec, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_def_n, 0, 8, st));
        pp.defer = DeferList{m->d_def_rec, m->d_def_cnt, m->d_def_n, (uint64_t)m->def_cap};
    }
    // scan -> key log + histogram by level-1 bucket, or by owner GPU for a sharded scan
    const uint32_t hist_nb = shard_send ? nown : pl.nb1, hist_shift = (uint32_t)(shard_send ? p.l : p.l - pl.b1);
    if (pl.fused) {
        const uint32_t nq2 = pl.nb1 * pl.cpr2;
        rc = ensure_ovq(m, (size_t)nq2 + pl.G1, pl.rw, st);
        if (rc != TSX_HIP_OK) return rc;
        const size_t lds = (size_t)pl.nb1 * (((size_t)8 << SP_CAPBITS) + 8 + 8 + 4 + 4);   // ring, flush descriptor, tail|head, cursor, job
        {
            // two kernels: strip descriptions (16 B per strip with a k-mer start, one region per wave, in buffer 0 --
            // level 2 overwrites it later), then the walk with every lane busy
            // all keys stay on this GPU: a ring flush per quarter strip (TSX_HIP_WALK_FLUSHQ=2|4: experiments)
            uint32_t local_fq = 1u;
            if (const char *e = getenv("TSX_HIP_WALK_FLUSHQ")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) local_fq = (uint32_t)v; }
            // (TSX_HIP_LOCAL_LONG=1: four strips per 32-byte description, as in the exchange of a sharded run)
            const char *ll_env = getenv("TSX_HIP_LOCAL_LONG");
            const int lng = ll_env ? (atoi(ll_env) != 0) : 0;
            const uint64_t desc_cap = ((ntiles + gd - 1) / gd) * (lng ? 16 : 64);
            rc = grow(st, m->d_buf[0], m->buf_bytes[0], (size_t)gdreg * desc_cap * (lng ? 32 : 16));
            if (rc != TSX_HIP_OK) return rc;
            hipLaunchKernelGGL(strip_desc_kernel<false>, dim3(gd), dim3(NT), 0, st, pp, d_text, n, own_end, head_open,
                               (const uint32_t *)m->d_tile, ntiles, (uint4 *)m->d_buf[0], desc_cap, pl.c_log, (unsigned long long *)nullptr, lng);
            HIP_TRY(hipGetLastError());
            if (lds > ((size_t)80 << 10))   // 512 lists: one workgroup per CU, 1024 threads
                hipLaunchKernelGGL(walk_part_kernel<1024>, dim3(pl.G1), dim3(1024), lds, st, pp, (const uint4 *)m->d_buf[0], desc_cap,
                                   (const unsigned long long *)pl.c_log, (uint32_t)gdreg, m->dbg, pl.buf1, pl.cap1, pl.c_l1, pl.nb1,
                                   (uint32_t)(p.l - pl.b1), m->d_ovq + (size_t)nq2 * OVQ_CAP, m->d_ovq_cnt + nq2, OVQ_CAP,
                                   (uint64_t)0, 0u, pl.G1, 0, (unsigned long long *)nullptr, lng, local_fq);
            else
                hipLaunchKernelGGL(walk_part_kernel<SP_NT>, dim3(pl.G1), dim3(SP_NT), lds, st, pp, (const uint4 *)m->d_buf[0], desc_cap,
                                   (const unsigned long long *)pl.c_log, (uint32_t)gdreg, m->dbg, pl.buf1, pl.cap1, pl.c_l1, pl.nb1,
                                   (uint32_t)(p.l - pl.b1), m->d_ovq + (size_t)nq2 * OVQ_CAP, m->d_ovq_cnt + nq2, OVQ_CAP,
                                   (uint64_t)0, 0u, pl.G1, 0, (unsigned long long *)nullptr, lng, local_fq);
        }
    } else if (p.wk == 1) {
        // key log form (sharded scans, one-level tables), the scan in two kernels as well: descriptions into buffer 1
        // (level 1 fills it only afterwards), then the walk with every lane busy into the wave's log region
        const uint64_t desc_cap = ((ntiles + gd - 1) / gd) * 64;
        rc = grow(st, m->d_buf[1], m->buf_bytes[1], (size_t)gdreg * desc_cap * 16);
        if (rc != TSX_HIP_OK) return rc;
        pl.buf1 = m->d_buf[1];
        {
            size_t have = m->desc_cnt_entries;
            rc = grow(st, m->d_desc_cnt, have, (size_t)gdreg * 8);
            m->desc_cnt_entries = have;
            if (rc != TSX_HIP_OK) return rc;
        }
        hipLaunchKernelGGL(strip_desc_kernel<false>, dim3(gd), dim3(NT), 0, st, pp, d_text, n, own_end, head_open,
                           (const uint32_t *)m->d_tile, ntiles, (uint4 *)m->d_buf[1], desc_cap, m->d_desc_cnt,
                           (unsigned long long *)nullptr, 0);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(walk_log_kernel, dim3(gs), dim3(NT), lut_bytes, st, pp, (const uint4 *)m->d_buf[1], desc_cap,
                           (const unsigned long long *)m->d_desc_cnt, (uint32_t)gdreg, m->dbg, m->d_buf[0], pl.log_cap,
                           pl.c_log, pl.d_hist, hist_nb, hist_shift, (uint64_t)0, 0, 0, (unsigned long long *)nullptr);
    } else {
        // multi-limb keys, two kernels as well: descriptions (first k-mer + entering bases + validity) into buffer 1,
        // then the walk with every lane busy
        const int du = (2 * p.wk + 2 + 3) / 4;
        const uint64_t desc_cap = ((ntiles + gd - 1) / gd) * 64;
        rc = grow(st, m->d_buf[1], m->buf_bytes[1], (size_t)gdreg * desc_cap * du * 16);
        if (rc != TSX_HIP_OK) return rc;
        pl.buf1 = m->d_buf[1];
        {
            size_t have = m->desc_cnt_entries;
            rc = grow(st, m->d_desc_cnt, have, (size_t)gdreg * 8);
            m->desc_cnt_entries = have;
            if (rc != TSX_HIP_OK) return rc;
        }
#define TSX_WIDE2(WKV)                                                                                                      \
        hipLaunchKernelGGL((strip_desc_wide_kernel<WKV>), dim3(gd), dim3(NT), 0, st, pp, d_text, n, own_end, head_open,      \
                           (const uint32_t *)m->d_tile, ntiles, (uint4 *)m->d_buf[1], desc_cap, m->d_desc_cnt);              \
        hipLaunchKernelGGL((walk_log_wide_kernel<WKV>), dim3(gs), dim3(NT), lut_bytes, st, pp, (const uint4 *)m->d_buf[1],   \
                           desc_cap, (const unsigned long long *)m->d_desc_cnt, (uint32_t)gdreg, m->dbg, m->d_buf[0],        \
                           pl.log_cap, pl.c_log, pl.d_hist, hist_nb, hist_shift)
        switch (p.wk) {
            case 2: TSX_WIDE2(2); break;
            case 3: TSX_WIDE2(3); break;
            default: TSX_WIDE2(4); break;
        }
#undef TSX_WIDE2
    }
    HIP_TRY(hipGetLastError());
    if (ev) { HIP_TRY(hipEventRecord(ev[2], st)); HIP_TRY(hipEventRecord(ev[3], st)); }
    if (shard_send) {
        // level 0: split every log region by owner into the caller's send buffer (exact offsets)
        if ((uint64_t)greg * pl.log_cap > shard_cap || (sh.own && (uint64_t)greg * pl.log_cap > sh.own_cap))
            return TSX_HIP_ERANGE;
        hipLaunchKernelGGL(offsets_rows_kernel, dim3(nown), dim3(1024), 0, st, (const uint32_t *)pl.d_hist, pl.d_offs,
                           (uint32_t)greg, pl.c_bcnt);
        hipLaunchKernelGGL(offsets_finish_kernel, dim3(1), dim3(1024), 0, st, nown, pl.c_bstart, pl.c_bcnt);
        hipLaunchKernelGGL(split_owner_kernel, dim3(std::min(greg, m->cus * 8)), dim3(PART_NT), 0, st,
                           (const uint64_t *)m->d_buf[0], (const unsigned long long *)pl.c_log, pl.log_cap,
                           (uint32_t)greg, shard_send, (const unsigned long long *)pl.d_offs,
                           (const unsigned long long *)pl.c_bstart, (const unsigned long long *)pl.c_bcnt, nown,
                           (uint32_t)p.l, sh.own, p.shard, sh.key_sum);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(shard_counts, pl.c_bcnt, nown * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
    } else {
        rc = run_partition_build(m, pl, m->d_buf[0], nullptr, pl.log_cap, st, ev);
        if (rc != TSX_HIP_OK) return rc;
    }
    if (ev && shard_send) {   // the partition phase follows in tsx_hip_shard_build_device, which records 3..6 again
        for (int i = 3; i < EV_N; ++i) HIP_TRY(hipEventRecord(ev[i], st));
        m->ev_open.push_back((long)(ev - m->ev.data()));
    }
    if (ev && !shard_send) HIP_TRY(hipEventRecord(ev[7], st));
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_shard_scan_window_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t win_off,
                                                size_t win_len, void *dev_send, size_t send_cap_keys, void *dev_own,
                                                size_t own_cap_keys, void *dev_send_counts, void *dev_hot_keys,
                                                void *dev_hot_counts, size_t hot_cap, void *dev_hot_n,
                                                void *dev_key_sum, void *stream) {
    if (!m || (!dev_text && n_total) || ((uintptr_t)dev_text & 15) || (win_off & 15) || !dev_send || !dev_send_counts ||
        !dev_hot_keys || !dev_hot_counts || !dev_hot_n || win_off > n_total || win_len > n_total - win_off)
        return TSX_HIP_EINVAL;
    if (m->p.wk != 1 || m->p.W != 1) return TSX_HIP_EINVAL;  // one-limb keys only (k <= 32)
    if (win_len >= ((size_t)4 << 30)) return TSX_HIP_ERANGE;  // one window per call
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    if (win_off == 0) HIP_TRY(hipMemsetAsync(m->d_carry, 0, 64, st));   // line index restarts with the text
    HIP_TRY(hipMemsetAsync(dev_hot_n, 0, 8, st));
    HIP_TRY(hipMemsetAsync(dev_send_counts, 0, sizeof(unsigned long long) << (m->p.lg - m->p.l), st));
    if (win_len == 0) return TSX_HIP_OK;
    HotOut hot;
    hot.keys = (uint64_t *)dev_hot_keys; hot.cnts = (uint64_t *)dev_hot_counts; hot.cap = hot_cap;
    hot.n = (unsigned long long *)dev_hot_n;
    ShardOut sh;
    sh.send = (uint64_t *)dev_send; sh.send_cap = send_cap_keys;
    sh.own = (uint64_t *)dev_own; sh.own_cap = own_cap_keys;
    sh.counts = (unsigned long long *)dev_send_counts; sh.key_sum = (unsigned long long *)dev_key_sum;
    // the window owns win_len start positions and sees the k-1 bytes after them; whether it starts inside a
    // line is read from the byte in front of it, on the device
    const size_t halo = (size_t)m->p.k - 1;
    const size_t len = std::min(win_len + halo, n_total - win_off);
    return run_fastq_piece(m, (const uint8_t *)dev_text + win_off, len, win_len, win_off ? -1 : 0, st, sh, hot);
}

extern "C" int tsx_hip_shard_scan_device(tsx_hip_map *m, const void *dev_text, size_t n, void *dev_send,
                                         size_t send_cap_keys, void *dev_send_counts, void *dev_hot_keys,
                                         void *dev_hot_counts, size_t hot_cap, void *dev_hot_n, void *stream) {
    return tsx_hip_shard_scan_window_device(m, dev_text, n, 0, n, dev_send, send_cap_keys, nullptr, 0, dev_send_counts,
                                            dev_hot_keys, dev_hot_counts, hot_cap, dev_hot_n, nullptr, stream);
}

extern "C" int tsx_hip_shard_send_capacity(tsx_hip_map *m, size_t text_bytes, size_t *keys_out) {
    if (!m || !keys_out) return TSX_HIP_EINVAL;
    const uint64_t ntiles = (text_bytes + TILE - 1) / TILE;
    const int g = (int)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, (uint64_t)m->cus * SCAN_WG_PER_CU)) * (NT / 64);
    const uint64_t maxrec = (m->p.line_mask == 3 ? text_bytes / 2 : text_bytes) + 65536;
    const uint64_t log_cap = (maxrec / g + maxrec / g / 3 + 2048 + 1) & ~1ULL;
    *keys_out = (size_t)g * log_cap;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_shard_build_pieces_device(tsx_hip_map *m, const void *dev_keys, const uint64_t *piece_off,
                                                 const uint64_t *piece_cnt, size_t npieces, void *dev_key_sum,
                                                 void *stream) {
    if (!m || !dev_keys || ((uintptr_t)dev_keys & 7) || (npieces && (!piece_off || !piece_cnt))) return TSX_HIP_EINVAL;
    unsigned long long *key_sum = (unsigned long long *)dev_key_sum;
    uint64_t n_keys = 0;
    for (size_t i = 0; i < npieces; ++i) n_keys += piece_cnt[i];
    if (n_keys == 0) return TSX_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    const uint64_t *keys = (const uint64_t *)dev_keys;
    if (!can_partition(m)) {  // tiny tables: plain atomic inserts of the hashed keys
        int rcz = ensure_zeroed(m, st);
        if (rcz != TSX_HIP_OK) return rcz;
        for (size_t i = 0; i < npieces; ++i) {
            if (!piece_cnt[i]) continue;
            hipLaunchKernelGGL(add_hashed_kernel, dim3(grid_for(m, piece_cnt[i], 8)), dim3(PART_NT), 0, st, m->p,
                               keys + piece_off[i], (const uint64_t *)nullptr, (uint64_t)piece_cnt[i], key_sum);
            HIP_TRY(hipGetLastError());
        }
        if (!m->ev_open.empty()) m->ev_open.pop_front();
        return TSX_HIP_OK;
    }
    // regions of the level-1 partition: every piece cut into runs of about n_keys / (3 per CU) keys
    const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>((n_keys + 4095) / 4096, (uint64_t)m->cus * 3));
    const uint64_t region_len = (n_keys + want - 1) / want;
    m->h_regions.clear();
    std::vector<unsigned long long> cnts;
    for (size_t i = 0; i < npieces; ++i)
        for (uint64_t o = 0; o < piece_cnt[i]; o += region_len) {
            m->h_regions.push_back(piece_off[i] + o);
            cnts.push_back(std::min<uint64_t>(region_len, piece_cnt[i] - o));
        }
    const int g = (int)m->h_regions.size();
    m->h_regions.insert(m->h_regions.end(), cnts.begin(), cnts.end());
    PartPlan pl;
    // Level 1 without a histogram pass: every region's workgroup keeps its own fixed-capacity sub-list per bucket
    // (as level 2 does), level 2 reads a bucket as the regions' pieces -- the plan's `fused` form with G1 = regions.
    // (One-level tables, many-limb keys: histogram + exact offsets as before.)
    int rc = plan_partition(m, n_keys + 65536, g, false, 0, st, pl, m->p.wk == 1 ? g : 0);
    if (rc != TSX_HIP_OK) return rc;
    rc = ensure_deferred(m, n_keys + 65536, st);
    if (rc != TSX_HIP_OK) return rc;
    HIP_TRY(hipMemsetAsync(m->d_def_n, 0, 8, st));
    // region table: starts into c_rstart, sizes into c_log (plan_partition laid them out back to back: [c_log | c_rstart])
    HIP_TRY(hipMemcpyAsync(pl.c_rstart, m->h_regions.data(), (size_t)g * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(pl.c_log, m->h_regions.data() + g, (size_t)g * 8, hipMemcpyHostToDevice, st));
    hipEvent_t *ev = nullptr;
    if (m->timing && !m->ev_open.empty() && (size_t)m->ev_open.front() + EV_N <= m->ev_used) {
        ev = &m->ev[(size_t)m->ev_open.front()];
        HIP_TRY(hipEventRecord(ev[3], st));   // the histogram of the received keys counts as level 1
    }
    if (!m->ev_open.empty()) m->ev_open.pop_front();
    if (pl.fused) {
        const uint32_t nq2 = pl.nb1 * pl.cpr2;
        rc = ensure_ovq(m, (size_t)nq2 + pl.G1, pl.rw, st);
        if (rc != TSX_HIP_OK) return rc;
        TableParams pp = m->p;
        pp.defer = DeferList{m->d_def_rec, m->d_def_cnt, m->d_def_n, (uint64_t)m->def_cap};
        const uint32_t bits = 5;   // 32-word rings: 16 words may stay behind a flush, 8 arrive per batch on average
        hipLaunchKernelGGL((partition_ring_kernel<1>), dim3(g), dim3(RING_NT), (size_t)pl.nb1 * (((size_t)8 << bits) + 36), st,
                           pp, keys, (const unsigned long long *)pl.c_rstart, (const unsigned long long *)pl.c_log, (uint64_t)0,
                           (uint32_t)g, 1u, pl.buf1, (const unsigned long long *)nullptr,
                           (const unsigned long long *)nullptr, pl.c_l1, pl.cap1, pl.nb1, (uint32_t)(m->p.l - pl.b1), bits,
                           m->dbg, m->d_ovq + (size_t)nq2 * OVQ_CAP, m->d_ovq_cnt + nq2, OVQ_CAP,
                           (const unsigned long long *)nullptr, 0u, (uint64_t)0, 1, key_sum, 0u, (uint32_t)g, 0, 0u, (const uint32_t *)nullptr);
    } else {
        hipLaunchKernelGGL(hist_kernel, dim3(g), dim3(PART_NT), 0, st, keys, (uint32_t)g, pl.nb1, (uint32_t)(m->p.l - pl.b1),
                           pl.d_hist, (const unsigned long long *)pl.c_rstart, (const unsigned long long *)pl.c_log, key_sum);
    }
    HIP_TRY(hipGetLastError());
    rc = run_partition_build(m, pl, keys, pl.c_rstart, 0, st, ev);
    if (rc == TSX_HIP_OK && ev) HIP_TRY(hipEventRecord(ev[7], st));
    return rc;
}

extern "C" int tsx_hip_shard_build_device(tsx_hip_map *m, const void *dev_keys, size_t n_keys, void *dev_key_sum,
                                          void *stream) {
    if (!m || (!dev_keys && n_keys)) return TSX_HIP_EINVAL;
    if (n_keys == 0) return TSX_HIP_OK;
    const uint64_t off = 0, cnt = n_keys;
    return tsx_hip_shard_build_pieces_device(m, dev_keys, &off, &cnt, 1, dev_key_sum, stream);
}

// ---- sharded run, level 1 window by window ------------------------------------------------------------------
// The keys of exchange window w are partitioned by level 1 as soon as they have arrived (the exchange of the later
// windows is still running); only level 2 and the build wait for the last window.
extern "C" int tsx_hip_shard_l1_supported(tsx_hip_map *m) {
    if (!m || !can_partition(m) || m->p.wk != 1 || m->p.W != 1) return 0;
    const int nsegbits = m->p.l - m->p.S;
    const int b1 = std::min(9, (nsegbits <= 8) ? nsegbits : (nsegbits + 1) / 2);
    return nsegbits - b1 > 0 ? 1 : 0;
}

extern "C" int tsx_hip_shard_l1_window_device(tsx_hip_map *m, const void *dev_keys, size_t n_keys, uint32_t window,
                                              uint32_t nwindows, size_t est_total_keys, void *dev_key_sum, void *stream) {
    if (!m || (!dev_keys && n_keys) || ((uintptr_t)dev_keys & 7) || nwindows == 0 || window >= nwindows) return TSX_HIP_EINVAL;
    if (!tsx_hip_shard_l1_supported(m)) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    if (!m->sh_pl) m->sh_pl = new PartPlan();
    PartPlan &pl = *m->sh_pl;
    if (window == 0) {
        // one workgroup per region, two workgroups per CU: every window's launch fills the chip once
        m->sh_rw = (uint32_t)std::min<uint64_t>((uint64_t)m->cus * 2, (uint64_t)PART_MAX_PIECES * level2_cpr(m) / nwindows);
        m->sh_windows = nwindows;
        const int g1 = (int)(m->sh_rw * nwindows);
        const uint64_t maxrec = std::max<uint64_t>(est_total_keys, n_keys) + 65536;
        std::swap(m->d_buf[1], m->sh_buf1); std::swap(m->buf_bytes[1], m->sh_buf1_bytes);
        std::swap(m->d_cnt, m->sh_cnt); std::swap(m->cnt_entries, m->sh_cnt_entries);
        int rc = plan_partition(m, maxrec, g1, false, 0, st, pl, g1);
        std::swap(m->d_buf[1], m->sh_buf1); std::swap(m->buf_bytes[1], m->sh_buf1_bytes);
        std::swap(m->d_cnt, m->sh_cnt); std::swap(m->cnt_entries, m->sh_cnt_entries);
        if (rc != TSX_HIP_OK) return rc;
        if (!pl.fused) return TSX_HIP_EINVAL;
        rc = ensure_deferred(m, maxrec, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_def_n, 0, 8, st));
        const uint32_t nq2 = pl.nb1 * pl.cpr2;
        rc = ensure_ovq(m, (size_t)nq2 + pl.G1, pl.rw, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_ovq_cnt, 0, ((size_t)nq2 + pl.G1) * 4, st));   // queues of windows that never run
        m->h_regions.assign((size_t)2 * g1, 0);
    } else if (m->sh_windows != nwindows || !pl.fused) {
        return TSX_HIP_EINVAL;
    }
    if (n_keys == 0) return TSX_HIP_OK;
    // the window's keys in sh_rw equal runs
    const uint32_t rw = m->sh_rw, g1 = pl.G1;
    const uint64_t len = (n_keys + rw - 1) / rw;
    unsigned long long *hs = m->h_regions.data() + (size_t)window * rw, *hc = m->h_regions.data() + g1 + (size_t)window * rw;
    for (uint32_t r = 0; r < rw; ++r) {
        const uint64_t o = std::min<uint64_t>((uint64_t)r * len, n_keys);
        hs[r] = o;
        hc[r] = std::min<uint64_t>(len, n_keys - o);
    }
    unsigned long long *ds = pl.c_rstart + (size_t)window * rw, *dc = pl.c_log + (size_t)window * rw;
    HIP_TRY(hipMemcpyAsync(ds, hs, (size_t)rw * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dc, hc, (size_t)rw * 8, hipMemcpyHostToDevice, st));
    TableParams pp = m->p;
    pp.defer = DeferList{m->d_def_rec, m->d_def_cnt, m->d_def_n, (uint64_t)m->def_cap};
    const uint32_t nq2 = pl.nb1 * pl.cpr2, bits = 5;
    hipLaunchKernelGGL((partition_ring_kernel<1>), dim3(rw), dim3(RING_NT), (size_t)pl.nb1 * (((size_t)8 << bits) + 36), st, pp,
                       (const uint64_t *)dev_keys, (const unsigned long long *)ds, (const unsigned long long *)dc, (uint64_t)0, rw,
                       1u, pl.buf1, (const unsigned long long *)nullptr, (const unsigned long long *)nullptr, pl.c_l1,
                       pl.cap1, pl.nb1, (uint32_t)(m->p.l - pl.b1), bits, m->dbg,
                       m->d_ovq + ((size_t)nq2 + (size_t)window * rw) * OVQ_CAP, m->d_ovq_cnt + nq2 + (size_t)window * rw, OVQ_CAP,
                       (const unsigned long long *)nullptr, 0u, (uint64_t)0, 1, (unsigned long long *)dev_key_sum,
                       window * rw, g1, 0, 0u, (const uint32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return TSX_HIP_OK;
}

// ---- sharded run, DESCRIPTION exchange (small world sizes) -------------------------------------------------------
// Keys cost 8 bytes per k-mer occurrence on the wire; a strip description (16 bytes) stands for up to 16 of them.
// Instead of sending every key to its owner, every GPU describes its window (tsx_hip_shard_desc_window_device), the
// descriptions are ALL-GATHERED, and every GPU walks all of them, keeping the keys it owns
// (tsx_hip_shard_walk_device): N x the rolling work, N/8 of the traffic of the key exchange -- a quarter at N = 2.
extern "C" int tsx_hip_shard_desc_capacity(tsx_hip_map *m, size_t text_bytes, int long_desc, size_t *descs_out) {
    if (!m || !descs_out) return TSX_HIP_EINVAL;
    // one description per 16 (long: 64) start positions at most; a long one is 32 bytes, a short one 16
    *descs_out = text_bytes / (long_desc ? 64 : 16) + 4096;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_shard_desc_window_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t win_off,
                                                size_t win_len, int long_desc, void *dev_desc, size_t desc_cap,
                                                void *dev_count, void *dev_kmer_sum, void *stream) {
    if (!m || (!dev_text && n_total) || ((uintptr_t)dev_text & 15) || (win_off & 15) || !dev_desc || ((uintptr_t)dev_desc & 15) ||
        !dev_count || win_off > n_total || win_len > n_total - win_off)
        return TSX_HIP_EINVAL;
    if (!tsx_hip_shard_l1_supported(m)) return TSX_HIP_EINVAL;
    if (win_len >= ((size_t)4 << 30)) return TSX_HIP_ERANGE;
    if (desc_cap < win_len / (long_desc ? 64 : 16) + 1) return TSX_HIP_ERANGE;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    if (win_off == 0) HIP_TRY(hipMemsetAsync(m->d_carry, 0, 64, st));
    HIP_TRY(hipMemsetAsync(dev_count, 0, 8, st));
    if (win_len == 0) return TSX_HIP_OK;
    DescOut dsc;
    dsc.out = (uint4 *)dev_desc; dsc.cap = desc_cap; dsc.count = (unsigned long long *)dev_count;
    dsc.sum = (unsigned long long *)dev_kmer_sum;
    dsc.long_desc = long_desc ? 1 : 0;
    const size_t halo = (size_t)m->p.k - 1;
    const size_t len = std::min(win_len + halo, n_total - win_off);
    return run_fastq_piece(m, (const uint8_t *)dev_text + win_off, len, win_len, win_off ? -1 : 0, st, ShardOut(), HotOut(), dsc);
}

// ---- owner = f(minimizer) (tsx_minimizer.h): every GPU holds a whole table of the k-mers it owns -------------------
// tsx_hip_mini_window_device describes one text window and splits the descriptions by owner GPU; the caller ships list o
// to GPU o (all-to-all), walks what it received with tsx_hip_shard_walk_device (a map with shard_bits = 0 keeps every key),
// builds with tsx_hip_shard_build_l1_device and adds the homopolymer totals it owns (tsx_hip_add_kmers_device).
extern "C" int tsx_hip_mini_supported(tsx_hip_map *m) {
    return (m && tsx_hip_shard_l1_supported(m) && m->p.lg == m->p.l && mz_supported((uint32_t)m->p.k)) ? 1 : 0;
}

// descriptions one owner's list may take when a text of text_bytes is described at once and split in nparts shares: a
// strip of 16 start positions yields at most one description per owner, + one for a run it shares with its neighbour; a
// region's share rounds up; every workgroup may leave a chunk open, and a list is as long as its busiest workgroup made it
static size_t mini_part_cap(const tsx_hip_map *m, size_t text_bytes, uint32_t nparts) {
    return text_bytes / 8 / nparts + 65536 + 4096 + (size_t)2 * MZ_CHUNK * (size_t)m->cus * MZ_WG_PER_CU;
}
extern "C" int tsx_hip_mini_part_capacity(tsx_hip_map *m, size_t text_bytes, uint32_t nparts, size_t *descs_per_owner_out) {
    if (!m || !descs_per_owner_out || nparts == 0) return TSX_HIP_EINVAL;
    *descs_per_owner_out = mini_part_cap(m, text_bytes, nparts);
    return TSX_HIP_OK;
}
extern "C" int tsx_hip_mini_capacity(tsx_hip_map *m, size_t text_bytes, int nranks, size_t *descs_per_owner_out) {
    if (!m || !descs_per_owner_out || nranks < 1 || nranks > MZ_MAX_RANKS) return TSX_HIP_EINVAL;
    *descs_per_owner_out = mini_part_cap(m, text_bytes, 1);
    return TSX_HIP_OK;
}

// share `part` of `nparts` of the described text -> one packed list per owner
static int mini_split(tsx_hip_map *m, uint32_t part, uint32_t nparts, int nranks, void *dev_desc, size_t cap_per_owner,
                      void *dev_counts, hipStream_t st) {
    HIP_TRY(hipMemsetAsync(dev_counts, 0, ((size_t)nranks + 4) * 8, st));
    if (m->mz_regions == 0) return TSX_HIP_OK;   // an empty text
    static const int mz_wgs = getenv("TSX_HIP_MZ_WGS") ? std::min(MZ_WG_PER_CU, std::max(1, atoi(getenv("TSX_HIP_MZ_WGS")))) : MZ_WG_PER_CU;
    static const int mz_merge = getenv("TSX_HIP_MZ_MERGE") ? atoi(getenv("TSX_HIP_MZ_MERGE")) : 1;
    const int gdr = (int)m->mz_regions, gsp = std::min(gdr, m->cus * mz_wgs);
    unsigned long long *d_cnt = m->d_desc_cnt, *count = (unsigned long long *)dev_counts;
    uint32_t *d_used = (uint32_t *)(d_cnt + 2 * (size_t)gdr + 8);   // chunks taken per (owner, workgroup)
    hipLaunchKernelGGL(desc_owner_split_kernel, dim3(gsp), dim3(MZ_NT), 0, st, m->p, (const uint4 *)m->d_buf[1], m->mz_dcap,
                       (const unsigned long long *)d_cnt, (uint32_t)gdr, (uint32_t)nranks, (uint4 *)dev_desc, (uint64_t)cap_per_owner,
                       d_used, count + nranks, mz_merge, part, nparts,
                       (const unsigned long long *)(d_cnt + 2 * (size_t)gdr + 8 + ((size_t)MZ_MAX_RANKS * m->cus * MZ_WG_PER_CU + 1) / 2));
    hipLaunchKernelGGL(desc_owner_finish_kernel, dim3(nranks), dim3(MZ_NT), 0, st, (const uint32_t *)d_used, (uint32_t)gsp,
                       (uint32_t)nranks, (uint4 *)dev_desc, (uint64_t)cap_per_owner, count, m->p.stats);
    HIP_TRY(hipGetLastError());
    return TSX_HIP_OK;
}

static int mini_describe(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t off, size_t len, void *dev_kmer_sum,
                         hipStream_t st) {
    if (off == 0) HIP_TRY(hipMemsetAsync(m->d_carry, 0, 64, st));
    m->mz_regions = 0; m->mz_len = len;
    if (len == 0) return TSX_HIP_OK;
    DescOut dsc;
    dsc.sum = (unsigned long long *)dev_kmer_sum;
    dsc.owners = 1;
    const size_t halo = (size_t)m->p.k - 1;
    const size_t ext = std::min(len + halo, n_total - off);
    return run_fastq_piece(m, (const uint8_t *)dev_text + off, ext, len, off ? -1 : 0, st, ShardOut(), HotOut(), dsc);
}

extern "C" int tsx_hip_mini_describe_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t off, size_t len,
                                            void *dev_kmer_sum, void *stream) {
    if (!m || (!dev_text && n_total) || ((uintptr_t)dev_text & 15) || (off & 15) || off > n_total || len > n_total - off)
        return TSX_HIP_EINVAL;
    if (!tsx_hip_mini_supported(m)) return TSX_HIP_EINVAL;
    if (len >= ((size_t)4 << 30)) return TSX_HIP_ERANGE;
    HIP_TRY(hipSetDevice(m->device));
    return mini_describe(m, dev_text, n_total, off, len, dev_kmer_sum, pick_stream(m, stream));
}

extern "C" int tsx_hip_mini_split_device(tsx_hip_map *m, uint32_t part, uint32_t nparts, int nranks, void *dev_desc,
                                         size_t cap_per_owner, void *dev_counts, void *stream) {
    if (!m || nparts == 0 || part >= nparts || !dev_desc || ((uintptr_t)dev_desc & 15) || !dev_counts || nranks < 1 ||
        nranks > MZ_MAX_RANKS)
        return TSX_HIP_EINVAL;
    if (!tsx_hip_mini_supported(m)) return TSX_HIP_EINVAL;
    if (m->mz_regions && cap_per_owner < mini_part_cap(m, m->mz_len, nparts)) return TSX_HIP_ERANGE;
    HIP_TRY(hipSetDevice(m->device));
    return mini_split(m, part, nparts, nranks, dev_desc, cap_per_owner, dev_counts, pick_stream(m, stream));
}

extern "C" int tsx_hip_mini_window_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t win_off, size_t win_len,
                                          int nranks, void *dev_desc, size_t cap_per_owner, void *dev_counts,
                                          void *dev_kmer_sum, void *stream) {
    if (!m || (!dev_text && n_total) || ((uintptr_t)dev_text & 15) || (win_off & 15) || !dev_desc || ((uintptr_t)dev_desc & 15) ||
        !dev_counts || win_off > n_total || win_len > n_total - win_off || nranks < 1 || nranks > MZ_MAX_RANKS)
        return TSX_HIP_EINVAL;
    if (!tsx_hip_mini_supported(m)) return TSX_HIP_EINVAL;
    if (win_len >= ((size_t)4 << 30)) return TSX_HIP_ERANGE;
    size_t need = 0;
    tsx_hip_mini_capacity(m, win_len, nranks, &need);
    if (cap_per_owner < need) return TSX_HIP_ERANGE;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    int rc = mini_describe(m, dev_text, n_total, win_off, win_len, dev_kmer_sum, st);
    if (rc != TSX_HIP_OK) return rc;
    return mini_split(m, 0, 1, nranks, dev_desc, cap_per_owner, dev_counts, st);
}

extern "C" int tsx_hip_mini_owner_host(int k, int nranks, const uint64_t *kmers, size_t n, uint32_t *owners_out) {
    if (!mz_supported((uint32_t)k) || nranks < 1 || nranks > MZ_MAX_RANKS || (!kmers && n) || (!owners_out && n)) return TSX_HIP_EINVAL;
    for (size_t i = 0; i < n; ++i) owners_out[i] = mz_owner_of_kmer(kmers[i], (uint32_t)k, (uint32_t)nranks);
    return TSX_HIP_OK;
}

// Walks n_desc packed descriptions (any GPU's), keeps the keys this shard owns and partitions them by radix level 1
// into list set `slot` of `nslots` (slot 0 plans for est_total_keys owned keys in all).  dev_emit_sum += k-mer
// occurrences kept.  Then tsx_hip_shard_build_l1_device.
extern "C" int tsx_hip_shard_walk_device(tsx_hip_map *m, const void *dev_desc, size_t n_desc, int long_desc, uint32_t slot,
                                         uint32_t nslots, size_t est_total_keys, void *dev_emit_sum, void *stream) {
    if (!m || (!dev_desc && n_desc) || ((uintptr_t)dev_desc & 15) || nslots == 0 || slot >= nslots) return TSX_HIP_EINVAL;
    if (!tsx_hip_shard_l1_supported(m)) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    if (!m->sh_pl) m->sh_pl = new PartPlan();
    PartPlan &pl = *m->sh_pl;
    // long_desc & 2: what the minimizer exchange sent to a map with shard_bits = 0 -- every key stays, homopolymers were
    // taken out by the sender, about half of a description's 16 positions are valid (a flush every second quarter), and
    // the windows of a step APPEND to one set of level-1 sub-lists (level 2 then reads as many pieces as on one GPU)
    const bool mini = (long_desc & 2) != 0;
    long_desc &= 1;
    if (mini && m->p.lg != m->p.l) return TSX_HIP_EINVAL;
    const uint32_t caller_slots = nslots, caller_slot = slot;
    if (mini) { nslots = 1; slot = 0; }
    if (caller_slot == 0) {
        // lists per slot: two workgroups per CU while the pieces of a bucket stay within what a level-2 workgroup walks
        uint32_t gw = (uint32_t)m->cus * 2;
        while (gw > 32 && (uint64_t)gw * nslots > (uint64_t)PART_MAX_PIECES * level2_cpr(m)) gw /= 2;
        m->sh_rw = gw;
        m->sh_windows = caller_slots;
        const int g1 = (int)(gw * nslots);
        const uint64_t maxrec = est_total_keys + 65536;
        std::swap(m->d_buf[1], m->sh_buf1); std::swap(m->buf_bytes[1], m->sh_buf1_bytes);
        std::swap(m->d_cnt, m->sh_cnt); std::swap(m->cnt_entries, m->sh_cnt_entries);
        int rc = plan_partition(m, maxrec, g1, false, 0, st, pl, g1);
        std::swap(m->d_buf[1], m->sh_buf1); std::swap(m->buf_bytes[1], m->sh_buf1_bytes);
        std::swap(m->d_cnt, m->sh_cnt); std::swap(m->cnt_entries, m->sh_cnt_entries);
        if (rc != TSX_HIP_OK) return rc;
        if (!pl.fused) return TSX_HIP_EINVAL;
        rc = ensure_deferred(m, maxrec, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_def_n, 0, 8, st));
        const uint32_t nq2 = pl.nb1 * pl.cpr2;
        rc = ensure_ovq(m, (size_t)nq2 + pl.G1, pl.rw, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_ovq_cnt, 0, ((size_t)nq2 + pl.G1) * 4, st));
    } else if (m->sh_windows != caller_slots || !pl.fused) {
        return TSX_HIP_EINVAL;
    }
    if (caller_slot == 0 && m->timing && !m->ev_open.empty() && (size_t)m->ev_open.front() + EV_N <= m->ev_used) {
        HIP_TRY(hipEventRecord(m->ev[(size_t)m->ev_open.front() + 3], st));   // "level 1" = the walks, up to the start of level 2
        m->sh_ev3 = true;
    }
    if (n_desc == 0) return TSX_HIP_OK;
    const uint32_t gw = m->sh_rw, nq2 = pl.nb1 * pl.cpr2;
    const uint64_t chunk = (n_desc + gw - 1) / gw;   // descriptions per workgroup
    // this GPU keeps one key in 2^shard_bits: a ring flush every 1, 2 or 4 quarter strips (walk_part_kernel)
    const uint32_t nown = 1u << (m->p.lg - m->p.l);
    uint32_t flush_q = nown >= 4 ? 4u : (nown == 2 ? 2u : 1u);
    const int own_mode = mini ? (2 | (caller_slot > 0 ? 4 : 0)) : 1;
    if (mini) flush_q = 2u;
    if (const char *e = getenv("TSX_HIP_WALK_FLUSHQ")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) flush_q = (uint32_t)v; }
    TableParams pp = m->p;
    pp.defer = DeferList{m->d_def_rec, m->d_def_cnt, m->d_def_n, (uint64_t)m->def_cap};
    const size_t lds = (size_t)pl.nb1 * (((size_t)8 << SP_CAPBITS) + 8 + 8 + 4 + 4);
    if (lds > ((size_t)80 << 10))   // 512 lists: one workgroup per CU, 1024 threads
        hipLaunchKernelGGL(walk_part_kernel<1024>, dim3(gw), dim3(1024), lds, st, pp, (const uint4 *)dev_desc, chunk,
                           (const unsigned long long *)nullptr, gw, m->dbg, pl.buf1, pl.cap1, pl.c_l1, pl.nb1,
                           (uint32_t)(m->p.l - pl.b1), m->d_ovq + ((size_t)nq2 + (size_t)slot * gw) * OVQ_CAP,
                           m->d_ovq_cnt + nq2 + (size_t)slot * gw, OVQ_CAP, (uint64_t)n_desc, slot * gw, pl.G1, own_mode,
                           (unsigned long long *)dev_emit_sum, long_desc ? 1 : 0, flush_q);
    else
        hipLaunchKernelGGL(walk_part_kernel<SP_NT>, dim3(gw), dim3(SP_NT), lds, st, pp, (const uint4 *)dev_desc, chunk,
                           (const unsigned long long *)nullptr, gw, m->dbg, pl.buf1, pl.cap1, pl.c_l1, pl.nb1,
                           (uint32_t)(m->p.l - pl.b1), m->d_ovq + ((size_t)nq2 + (size_t)slot * gw) * OVQ_CAP,
                           m->d_ovq_cnt + nq2 + (size_t)slot * gw, OVQ_CAP, (uint64_t)n_desc, slot * gw, pl.G1, own_mode,
                           (unsigned long long *)dev_emit_sum, long_desc ? 1 : 0, flush_q);
    HIP_TRY(hipGetLastError());
    return TSX_HIP_OK;
}

// The same in two kernels, for larger world sizes: this GPU keeps one key in N, so the fused walk is mostly waiting
// for its rolling chains at 16 waves per CU (1.9 ms per 1e9 positions at N = 8).  walk_log_kernel in its owner-filtered
// form has no rings to hold (20 waves per CU) and logs the kept keys per wave; level 1 of partition_ring_kernel then
// reads the wave logs as pieces (512 workgroups, each streaming ten of them) into list set `slot` of `nslots`.
extern "C" int tsx_hip_shard_filter_device(tsx_hip_map *m, const void *dev_desc, size_t n_desc, int long_desc, uint32_t slot,
                                           uint32_t nslots, size_t est_total_keys, void *dev_emit_sum, void *stream) {
    if (!m || (!dev_desc && n_desc) || ((uintptr_t)dev_desc & 15) || nslots == 0 || slot >= nslots) return TSX_HIP_EINVAL;
    if (!tsx_hip_shard_l1_supported(m)) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    if (!m->sh_pl) m->sh_pl = new PartPlan();
    PartPlan &pl = *m->sh_pl;
    if (slot == 0) {
        uint32_t gw = (uint32_t)m->cus * 2;
        while (gw > 32 && (uint64_t)gw * nslots > (uint64_t)PART_MAX_PIECES * level2_cpr(m)) gw /= 2;
        m->sh_rw = gw;
        m->sh_windows = nslots;
        const int g1 = (int)(gw * nslots);
        const uint64_t maxrec = est_total_keys + 65536;
        std::swap(m->d_buf[1], m->sh_buf1); std::swap(m->buf_bytes[1], m->sh_buf1_bytes);
        std::swap(m->d_cnt, m->sh_cnt); std::swap(m->cnt_entries, m->sh_cnt_entries);
        int rc = plan_partition(m, maxrec, g1, false, 0, st, pl, g1);
        std::swap(m->d_buf[1], m->sh_buf1); std::swap(m->buf_bytes[1], m->sh_buf1_bytes);
        std::swap(m->d_cnt, m->sh_cnt); std::swap(m->cnt_entries, m->sh_cnt_entries);
        if (rc != TSX_HIP_OK) return rc;
        if (!pl.fused) return TSX_HIP_EINVAL;
        rc = ensure_deferred(m, maxrec, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_def_n, 0, 8, st));
        const uint32_t nq2 = pl.nb1 * pl.cpr2;
        rc = ensure_ovq(m, (size_t)nq2 + pl.G1, pl.rw, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipMemsetAsync(m->d_ovq_cnt, 0, ((size_t)nq2 + pl.G1) * 4, st));
    } else if (m->sh_windows != nslots || !pl.fused) {
        return TSX_HIP_EINVAL;
    }
    if (slot == 0 && m->timing && !m->ev_open.empty() && (size_t)m->ev_open.front() + EV_N <= m->ev_used) {
        HIP_TRY(hipEventRecord(m->ev[(size_t)m->ev_open.front() + 3], st));
        m->sh_ev3 = true;
    }
    if (n_desc == 0) return TSX_HIP_OK;
    // the wave logs of this slot: the map's own scratch (buffer 0), planned for what this slot may keep
    const int gs = (int)m->cus * SCAN_WG_PER_CU, greg = gs * (NT / 64);
    PartPlan lp;
    const uint64_t keep = est_total_keys / nslots + est_total_keys / nslots / 2 + 65536;
    int rc = plan_partition(m, keep, greg, true, 0, st, lp, 0);
    if (rc != TSX_HIP_OK) return rc;
    TableParams pp = m->p;
    pp.defer = DeferList{m->d_def_rec, m->d_def_cnt, m->d_def_n, (uint64_t)m->def_cap};
    const uint64_t chunk = (n_desc + greg - 1) / greg;   // descriptions per wave
    hipLaunchKernelGGL(walk_log_kernel, dim3(gs), dim3(NT), m->lut.size() * 8, st, pp, (const uint4 *)dev_desc, chunk,
                       (const unsigned long long *)nullptr, (uint32_t)greg, m->dbg, m->d_buf[0], lp.log_cap, lp.c_log, lp.d_hist,
                       lp.nb1, (uint32_t)(m->p.l - lp.b1), (uint64_t)n_desc, long_desc ? 1 : 0, 1,
                       (unsigned long long *)dev_emit_sum);
    HIP_TRY(hipGetLastError());
    const uint32_t gw = m->sh_rw, nq2 = pl.nb1 * pl.cpr2, bits = 5;
    hipLaunchKernelGGL((partition_ring_kernel<1>), dim3(gw), dim3(RING_NT), (size_t)pl.nb1 * (((size_t)8 << bits) + 36), st, pp,
                       (const uint64_t *)m->d_buf[0], (const unsigned long long *)nullptr, (const unsigned long long *)nullptr,
                       (uint64_t)0, 1u, gw, pl.buf1, (const unsigned long long *)nullptr, (const unsigned long long *)nullptr,
                       pl.c_l1, pl.cap1, pl.nb1, (uint32_t)(m->p.l - pl.b1), bits, m->dbg,
                       m->d_ovq + ((size_t)nq2 + (size_t)slot * gw) * OVQ_CAP, m->d_ovq_cnt + nq2 + (size_t)slot * gw, OVQ_CAP,
                       (const unsigned long long *)lp.c_log, (uint32_t)greg, lp.log_cap, 1, (unsigned long long *)nullptr, slot,
                       nslots, 0, 0u, (const uint32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return TSX_HIP_OK;
}

// level 2 + build over the sub-lists the windows' level-1 launches have filled
extern "C" int tsx_hip_shard_build_l1_device(tsx_hip_map *m, void *stream) {
    if (!m || !m->sh_pl || !m->sh_pl->fused) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    hipEvent_t *ev = nullptr;
    if (m->timing && !m->ev_open.empty() && (size_t)m->ev_open.front() + EV_N <= m->ev_used) {
        ev = &m->ev[(size_t)m->ev_open.front()];
        if (!m->sh_ev3) HIP_TRY(hipEventRecord(ev[3], st));
    }
    m->sh_ev3 = false;
    if (!m->ev_open.empty()) m->ev_open.pop_front();
    int rc = run_partition_build(m, *m->sh_pl, nullptr, nullptr, 0, st, ev);
    if (rc == TSX_HIP_OK && ev) HIP_TRY(hipEventRecord(ev[7], st));
    return rc;
}

extern "C" int tsx_hip_add_hashed_device(tsx_hip_map *m, const void *dev_keys, const void *dev_counts, size_t n,
                                         void *stream) {
    if (!m || (!dev_keys && n)) return TSX_HIP_EINVAL;
    if (m->p.wk != 1 || m->p.W != 1) return TSX_HIP_EINVAL;
    if (n == 0) return TSX_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    int rcz = ensure_zeroed(m, st);
    if (rcz != TSX_HIP_OK) return rcz;
    if (dev_counts) {
        // (key, count) lists are the hot lists of the sharded scan: the same few keys over and over (every scan wave
        // drains its cache).  deferred_insert_kernel sums equal keys of a workgroup's share in LDS first -- one
        // same-address global atomic per workgroup instead of one per entry (0.36 -> 0.04 ms per window's list).
        const int grid = (int)std::min<uint64_t>((uint64_t)m->cus * 2, (n + 511) / 512);
        hipLaunchKernelGGL((deferred_insert_kernel<1>), dim3(grid), dim3(PART_NT), 0, st, m->p, (const uint64_t *)dev_keys,
                           (const uint64_t *)dev_counts, (const unsigned long long *)nullptr, (uint64_t)n, (uint64_t)n);
    } else {
        const int grid = grid_for(m, n, 8);
        hipLaunchKernelGGL(add_hashed_kernel, dim3(grid), dim3(PART_NT), 0, st, m->p, (const uint64_t *)dev_keys,
                           (const uint64_t *)dev_counts, (uint64_t)n, (unsigned long long *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_set_record_lines(tsx_hip_map *m, int lines) {
    if (!m || (lines != 2 && lines != 4)) return TSX_HIP_EINVAL;
    m->p.line_mask = (uint32_t)lines - 1u;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_set_path(tsx_hip_map *m, int path) {
    if (!m || path < 0 || path > 2) return TSX_HIP_EINVAL;
    m->path = path;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_set_timing(tsx_hip_map *m, int enable) {
    if (!m) return TSX_HIP_EINVAL;
    m->timing = enable ? 1 : 0;
    m->ev_used = 0;
    m->ev_open.clear();
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_get_stage_timing(tsx_hip_map *m, double *stage_ms, uint64_t *launches) {
    if (!m) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    // line, scan, level 1, level 2, build kernel, gap, inserts after the build (overflow queues + deferred list)
    static const int from[7] = {0, 1, 3, 4, 5, 2, 6}, to[7] = {1, 2, 4, 5, 6, 3, 7};
    for (size_t i = 0; i + EV_N <= m->ev_used; i += EV_N) {
        HIP_TRY(hipEventSynchronize(m->ev[i + 7]));
        for (int sgm = 0; sgm < 7; ++sgm) {
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, m->ev[i + from[sgm]], m->ev[i + to[sgm]]));
            acc[sgm] += t;
        }
    }
    if (stage_ms) for (int sgm = 0; sgm < 7; ++sgm) stage_ms[sgm] = acc[sgm];
    if (launches) *launches = m->ev_used / EV_N;
    m->ev_used = 0;
    m->ev_open.clear();
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_get_timing(tsx_hip_map *m, double *line_ms, double *count_ms, double *build_ms,
                                  uint64_t *launches) {
    double sgm[7];
    const int rc = tsx_hip_get_stage_timing(m, sgm, launches);
    if (rc != TSX_HIP_OK) return rc;
    if (line_ms) *line_ms = sgm[0];
    if (count_ms) *count_ms = sgm[1];
    if (build_ms) *build_ms = sgm[2] + sgm[3] + sgm[4] + sgm[6];
    return TSX_HIP_OK;
}

// ---- tables above 2^32 slots (l - S > 18): built slab by slab ----------------------------------------------------------
// 1. every window of the text is DESCRIBED once (line pass + strip_desc_kernel, long descriptions: 32 bytes per 64 start
//    positions), the descriptions of all windows stay in HBM;
// 2. for every slab: the owner-filtered walk over every window's descriptions keeps the slab's keys and partitions them by
//    radix level 1 (one set of sub-lists per window), then ONE level 2 + build for the slab -- a slab is to this loop what a
//    shard is to a GPU of a multi-GPU run, and the per-slab view of the table parameters is a shard's view (l = slab bits,
//    shard = slab number) with the table pointers moved to the slab and pos_base = its first slot.
// The text is walked slab_count times (rolling work only, ~2 ms per 1e9 positions); every key makes its two trips once.
static const size_t DEV_WINDOW_DEFAULT = (size_t)4 << 30;
static size_t dev_window_bytes() {
    size_t w = DEV_WINDOW_DEFAULT;
    if (const char *e = getenv("TSX_HIP_DEV_WINDOW")) {  // tests exercise the window seams
        const long long v = atoll(e);
        if (v >= 4096) w = ((size_t)v + 15) & ~(size_t)15;
    }
    return w;
}

static int count_slabs(tsx_hip_map *m, const uint8_t *base, size_t n, hipStream_t st) {
    const int sb = slab_bits(m);
    const uint32_t nslab = 1u << sb;
    const size_t halo = (size_t)m->p.k - 1, WIN = dev_window_bytes();
    const uint32_t nwin = (uint32_t)std::max<size_t>(1, (n + WIN - 1) / WIN);
    // descriptions of all windows, back to back (window w at word offset doff[w] of 32-byte descriptions)
    std::vector<size_t> doff(nwin + 1, 0);
    for (uint32_t w = 0; w < nwin; ++w) doff[w + 1] = doff[w] + std::min(WIN, n - (size_t)w * WIN) / 64 + 4096;
    // (scratch of the map, grown on demand: allocating and freeing gigabytes per call costs more than the kernels)
    int rc = grow(st, m->d_slabdesc, m->slabdesc_bytes, doff[nwin] * 32 + (size_t)(nwin + 1) * 8);
    if (rc != TSX_HIP_OK) return rc;
    uint4 *d_desc = reinterpret_cast<uint4 *>(m->d_slabdesc);
    unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(m->d_slabdesc + doff[nwin] * 32);
    auto done = [&](int code) { m->ev_open.clear(); return code; };
    if (hipMemsetAsync(d_cnt, 0, (size_t)(nwin + 1) * 8, st) != hipSuccess) return done(TSX_HIP_EHIP);
    for (uint32_t w = 0; w < nwin && rc == TSX_HIP_OK; ++w) {
        const size_t off = (size_t)w * WIN, own = std::min(WIN, n - off), len = std::min(own + halo, n - off);
        DescOut dsc;
        dsc.out = d_desc + doff[w] * 2; dsc.cap = doff[w + 1] - doff[w]; dsc.count = d_cnt + w; dsc.sum = d_cnt + nwin; dsc.long_desc = 1;
        rc = run_fastq_piece(m, base + off, len, own, off > 0 ? -1 : 0, st, ShardOut(), HotOut(), dsc);
    }
    if (rc != TSX_HIP_OK) return done(rc);
    std::vector<unsigned long long> cnt(nwin + 1);
    if (hipMemcpyAsync(cnt.data(), d_cnt, (size_t)(nwin + 1) * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return done(TSX_HIP_EHIP);
    m->ev_open.clear();   // (the description calls queued timing tuples for a sharded build that never comes)
    const uint64_t kmers = cnt[nwin];
    if (kmers == 0) return done(TSX_HIP_OK);
    const TableParams whole = m->p;
    const bool fresh = m->fresh;
    const size_t est = (size_t)(kmers / nslab + kmers / nslab / 8) + 65536;
    for (uint32_t s = 0; s < nslab && rc == TSX_HIP_OK; ++s) {
        TableParams &v = m->p;     // the slab's view
        v = whole;
        v.l = whole.l - sb;
        v.slot_mask = (1ULL << v.l) - 1ULL;
        v.shard = s;
        v.pos_base = (uint64_t)s << v.l;
        v.table = whole.table + ((uint64_t)s << v.l);
        v.seg_dirty = whole.seg_dirty + ((uint64_t)s << (v.l - v.S));
        m->fresh = fresh;
        hipEvent_t *ev = nullptr;
        if (m->timing) {
            if (m->ev_used + EV_N > m->ev.size())
                for (int i = 0; i < EV_N; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { rc = TSX_HIP_EHIP; break; } m->ev.push_back(e); }
            if (rc != TSX_HIP_OK) break;
            ev = &m->ev[m->ev_used]; m->ev_used += EV_N;
            for (int i = 0; i < 4; ++i) if (hipEventRecord(ev[i], st) != hipSuccess) rc = TSX_HIP_EHIP;
        }
        for (uint32_t w = 0; w < nwin && rc == TSX_HIP_OK; ++w)
            rc = tsx_hip_shard_walk_device(m, d_desc + doff[w] * 2, (size_t)cnt[w], 1, w, nwin, est, nullptr, st);
        if (rc == TSX_HIP_OK) {
            if (!m->sh_pl || !m->sh_pl->fused) rc = TSX_HIP_EINVAL;
            else rc = run_partition_build(m, *m->sh_pl, nullptr, nullptr, 0, st, ev);
        }
        if (rc == TSX_HIP_OK && ev && hipEventRecord(ev[7], st) != hipSuccess) rc = TSX_HIP_EHIP;
    }
    m->p = whole;
    if (rc == TSX_HIP_OK) m->fresh = false;
    return done(rc);
}

// Device texts are processed in windows so that the partition scratch (about
// 10 bytes per text byte) stays bounded; windows overlap by the k-1 byte halo
// exactly like the host pieces.
extern "C" int tsx_hip_count_fastq_device(tsx_hip_map *m, const void *dev_text, size_t n, void *stream) {
    if (!m || (!dev_text && n) || ((uintptr_t)dev_text & 15)) return TSX_HIP_EINVAL;
    if (m->p.lg != m->p.l) return TSX_HIP_EINVAL;   // a shard: keys of other owners must travel (shard_scan / shard_build)
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    HIP_TRY(hipMemsetAsync(m->d_carry, 0, 64, st));
    const uint8_t *base = (const uint8_t *)dev_text;
    const size_t halo = (size_t)m->p.k - 1;
    const size_t DEV_WINDOW = dev_window_bytes();
    if (slab_bits(m) && (m->path == 2 || (m->path == 0 && n * 32 >= m->lay.table_bytes))) return count_slabs(m, base, n, st);
    for (size_t off = 0; off < n || off == 0; off += DEV_WINDOW) {
        const size_t own = std::min(DEV_WINDOW, n - off);
        const size_t len = std::min(own + halo, n - off);
        // whether the previous window ends inside a line is read on the device (the byte in front of this one)
        int rc = run_fastq_piece(m, base + off, len, own, off > 0 ? -1 : 0, st);
        if (rc != TSX_HIP_OK) return rc;
        if (n == 0) break;
    }
    return TSX_HIP_OK;
}

// ---- blocked gzip (BGZF) input: members found on the host, inflated on the device (tsx_inflate.h) ---------
struct BgzfIndex {
    std::vector<uint64_t> in_off, out_off;
    std::vector<uint32_t> in_len, out_len, crc;
    uint64_t text_bytes = 0;
};
static inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static inline uint32_t le32(const uint8_t *p) { return le16(p) | (le16(p + 2) << 16); }

// gzip members (RFC 1952) that all carry the BGZF 'BC' extra subfield (SAM specification, section 4.1):
// BSIZE = size of the member - 1.  false: not BGZF (or damaged) -- the caller reads it with zlib instead.
static bool bgzf_index(const uint8_t *gz, size_t n, BgzfIndex &ix) {
    size_t o = 0;
    while (o < n) {
        if (n - o < 18 || gz[o] != 0x1f || gz[o + 1] != 0x8b || gz[o + 2] != 8 || !(gz[o + 3] & 4)) return false;
        if (gz[o + 3] & ~4) return false;   // FNAME/FCOMMENT/FHCRC: not written by bgzip, not parsed here
        const uint32_t xlen = le16(gz + o + 10);
        if (n - o < 12 + (size_t)xlen + 8) return false;
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *e = gz + o + 12 + x;
            const uint32_t slen = le16(e + 2);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2 && x + 6 <= xlen) { bsize = le16(e + 4); found = true; }
            x += 4 + slen;
        }
        if (!found) return false;
        const size_t total = (size_t)bsize + 1;
        if (total < 12 + (size_t)xlen + 8 || total > n - o) return false;
        const size_t data_off = o + 12 + xlen, data_len = total - (12 + xlen) - 8;
        const uint32_t isize = le32(gz + o + total - 4);
        if (isize > (1u << 16)) return false;   // a BGZF member holds at most 64 KiB
        ix.in_off.push_back(data_off);
        ix.in_len.push_back((uint32_t)data_len);
        ix.out_off.push_back(ix.text_bytes);
        ix.out_len.push_back(isize);
        ix.crc.push_back(le32(gz + o + total - 8));
        ix.text_bytes += isize;
        o += total;
    }
    return !ix.in_off.empty();
}

extern "C" int tsx_hip_bgzf_index_host(const void *gz, size_t n, size_t *members, size_t *text_bytes) {
    if (!gz && n) return TSX_HIP_EINVAL;
    BgzfIndex ix;
    if (!bgzf_index((const uint8_t *)gz, n, ix)) return TSX_HIP_EINVAL;
    if (members) *members = ix.in_off.size();
    if (text_bytes) *text_bytes = (size_t)ix.text_bytes;
    return TSX_HIP_OK;
}

// Device scratch of the BGZF path: the compressed bytes and the member index of ONE batch of members.
struct BgzfDev {
    uint8_t *d_gz = nullptr, *d_ix = nullptr;
    size_t gz_cap = 0, ix_cap = 0;
    uint32_t *d_tab = nullptr;      // CRC-32 tables
    ~BgzfDev() { (void)hipFree(d_gz); (void)hipFree(d_ix); (void)hipFree(d_tab); }
};

// Members are inflated in BATCHES of at most this many bytes of text (whole members, at least one), so that a large
// .fastq.gz needs two batch-sized text buffers instead of the whole text at once.  TSX_HIP_BGZF_BATCH: tests.
static size_t bgzf_batch_bytes() {
    // (a launch of the inflate kernel takes as long as ONE member takes, 16 ms, whatever the number of members: few, big batches)
    size_t v = (size_t)3 << 30;
    if (const char *e = getenv("TSX_HIP_BGZF_BATCH")) { const long long x = atoll(e); if (x > 0) v = (size_t)x; }
    return std::max<size_t>(v, (size_t)128 << 10);
}
// [m0, m1): the next batch from member m0 on -- members while the text stays within `batch` (or below 4 KiB)
static size_t bgzf_next_batch(const BgzfIndex &ix, size_t m0, size_t batch) {
    size_t m1 = m0, acc = 0;
    while (m1 < ix.in_off.size() && (m1 == m0 || acc < 4096 || acc + ix.out_len[m1] <= batch)) acc += ix.out_len[m1++];
    return m1;
}

// Inflates members [m0, m1) of gz: the text of member m0 starts at d_out[0].  Waits for the kernel and checks
// every member's status (stored / fixed / dynamic blocks decoded, ISIZE and CRC-32 right).
static int inflate_batch(const uint8_t *gz, size_t n, const BgzfIndex &ix, size_t m0, size_t m1, BgzfDev &dv,
                         uint8_t *d_out, hipStream_t st) {
    const size_t nm = m1 - m0;
    if (nm == 0) return TSX_HIP_OK;
    // CRC-32 tables for eight bytes per step: crc_tab[j][v] = CRC of byte v followed by j zero bytes
    static uint32_t crc_tab[8 * 256];
    static std::once_flag crc_once;
    std::call_once(crc_once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int b = 0; b < 8; ++b) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_tab[i] = c;
        }
        for (int j = 1; j < 8; ++j)
            for (uint32_t i = 0; i < 256; ++i)
                crc_tab[j * 256 + i] = (crc_tab[(j - 1) * 256 + i] >> 8) ^ crc_tab[crc_tab[(j - 1) * 256 + i] & 0xFFu];
    });
    if (!dv.d_tab) {
        HIP_TRY(hipMalloc((void **)&dv.d_tab, sizeof(crc_tab)));
        HIP_TRY(hipMemcpyAsync(dv.d_tab, crc_tab, sizeof(crc_tab), hipMemcpyHostToDevice, st));
        HIP_TRY(hipFuncSetAttribute((const void *)inflate_members_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)INF_LDS_BYTES));
    }
    const size_t lo = (size_t)ix.in_off[m0], hi = (size_t)ix.in_off[m1 - 1] + ix.in_len[m1 - 1];
    const size_t gz_bytes = std::min(n, hi + 16) - lo;       // the bit reader looks up to 16 bytes past a member
    int rc = grow(st, dv.d_gz, dv.gz_cap, (hi - lo) + 64);
    if (rc != TSX_HIP_OK) return rc;
    // one allocation for the index: in_off | out_off | in_len | out_len | crc | status
    rc = grow(st, dv.d_ix, dv.ix_cap, nm * (8 + 8 + 4 + 4 + 4 + 4));
    if (rc != TSX_HIP_OK) return rc;
    std::vector<uint64_t> in_off(nm), out_off(nm);
    for (size_t i = 0; i < nm; ++i) { in_off[i] = ix.in_off[m0 + i] - lo; out_off[i] = ix.out_off[m0 + i] - ix.out_off[m0]; }
    uint64_t *d_in_off = (uint64_t *)dv.d_ix, *d_out_off = d_in_off + nm;
    uint32_t *d_in_len = (uint32_t *)(d_out_off + nm), *d_out_len = d_in_len + nm, *d_crc = d_out_len + nm, *d_status = d_crc + nm;
    HIP_TRY(hipMemcpyAsync(dv.d_gz, gz + lo, gz_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_in_off, in_off.data(), nm * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_out_off, out_off.data(), nm * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_in_len, ix.in_len.data() + m0, nm * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_out_len, ix.out_len.data() + m0, nm * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_crc, ix.crc.data() + m0, nm * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_status, 0xFF, nm * 4, st));
    hipLaunchKernelGGL(inflate_members_kernel, dim3((uint32_t)((nm + INF_NT - 1) / INF_NT)), dim3(INF_NT), INF_LDS_BYTES, st,
                       (const uint8_t *)dv.d_gz, (const uint64_t *)d_in_off, (const uint32_t *)d_in_len,
                       (const uint64_t *)d_out_off, (const uint32_t *)d_out_len, (const uint32_t *)d_crc, (uint32_t)nm, d_out,
                       d_status, (const uint32_t *)dv.d_tab);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> status(nm);
    HIP_TRY(hipMemcpyAsync(status.data(), d_status, nm * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));      // (also: in_off / out_off of this frame have been copied)
    for (size_t i = 0; i < nm; ++i)
        if (status[i] != INF_OK) {
            static const char *why[] = {"ok", "deflate data truncated", "reserved block type", "stored block length check",
                                        "bad code lengths", "invalid symbol", "output overrun or distance too far",
                                        "size differs from ISIZE", "CRC-32 mismatch"};
            g_last_error = "BGZF member " + std::to_string(m0 + i) + ": " + (status[i] < 9 ? why[status[i]] : "not decoded");
            return TSX_HIP_EINVAL;
        }
    return TSX_HIP_OK;
}

static inline size_t bgzf_batch_text(const BgzfIndex &ix, size_t m0, size_t m1) {
    return (size_t)((m1 < ix.out_off.size() ? ix.out_off[m1] : ix.text_bytes) - ix.out_off[m0]);
}

extern "C" int tsx_hip_inflate_bgzf_host(int device, const void *gz, size_t n, void *out_host, size_t out_cap,
                                         size_t *out_bytes) {
    if ((!gz && n) || !out_bytes) return TSX_HIP_EINVAL;
    BgzfIndex ix;
    if (!bgzf_index((const uint8_t *)gz, n, ix)) { g_last_error = "not a BGZF file (no BC extra field in every gzip member)"; return TSX_HIP_EINVAL; }
    *out_bytes = (size_t)ix.text_bytes;
    if (ix.text_bytes > out_cap || (ix.text_bytes && !out_host)) return TSX_HIP_ERANGE;
    HIP_TRY(hipSetDevice(device));
    BgzfDev dv;
    uint8_t *d_out = nullptr;
    size_t out_have = 0;
    const size_t batch = bgzf_batch_bytes();
    int rc = TSX_HIP_OK;
    for (size_t m0 = 0; m0 < ix.in_off.size() && rc == TSX_HIP_OK;) {
        const size_t m1 = bgzf_next_batch(ix, m0, batch), nb = bgzf_batch_text(ix, m0, m1);
        rc = grow((hipStream_t) nullptr, d_out, out_have, nb + 256);
        if (rc == TSX_HIP_OK) rc = inflate_batch((const uint8_t *)gz, n, ix, m0, m1, dv, d_out, nullptr);
        if (rc == TSX_HIP_OK && nb && hipMemcpy((uint8_t *)out_host + ix.out_off[m0], d_out, nb, hipMemcpyDeviceToHost) != hipSuccess) {
            g_last_error = "hipMemcpy of the inflated text failed";
            rc = TSX_HIP_EHIP;
        }
        m0 = m1;
    }
    (void)hipFree(d_out);
    return rc;
}

// The text never exists as a whole: batch b is inflated into one of two buffers BEHIND the last BGZF_PRE + 16 bytes
// of batch b-1, and counted as a piece that owns the start positions up to BGZF_PRE bytes before its end (the last
// batch: all of them) -- the k-1 bytes a window needs behind its start are always there, the byte in front of a
// piece (is its first line open?) as well.  On an inflate error the table holds the batches before it.
static const size_t BGZF_PRE = 256;   // >= k - 1, a multiple of 16

extern "C" int tsx_hip_count_fastq_bgzf_host(tsx_hip_map *m, const void *gz, size_t n) {
    if (!m || (!gz && n)) return TSX_HIP_EINVAL;
    if (m->p.lg != m->p.l) return TSX_HIP_EINVAL;   // see tsx_hip_count_fastq_device
    BgzfIndex ix;
    if (!bgzf_index((const uint8_t *)gz, n, ix)) { g_last_error = "not a BGZF file (no BC extra field in every gzip member)"; return TSX_HIP_EINVAL; }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    join_foreign(m, false);
    HIP_TRY(hipMemsetAsync(m->d_carry, 0, 64, st));
    const size_t batch = bgzf_batch_bytes(), nm = ix.in_off.size(), head = BGZF_PRE + 16;
    size_t biggest = 0;
    for (size_t m0 = 0; m0 < nm;) { const size_t m1 = bgzf_next_batch(ix, m0, batch); biggest = std::max(biggest, bgzf_batch_text(ix, m0, m1)); m0 = m1; }
    BgzfDev dv;
    uint8_t *d_txt[2] = {nullptr, nullptr};
    const size_t buf_bytes = head + biggest + 256;
    int rc = TSX_HIP_OK;
    for (int i = 0; i < 2 && rc == TSX_HIP_OK; ++i)
        if ((i == 0 || bgzf_next_batch(ix, 0, batch) < nm) && hipMalloc((void **)&d_txt[i], buf_bytes) != hipSuccess) {
            g_last_error = "hipMalloc of a BGZF text buffer failed";
            rc = TSX_HIP_ENOMEM;
        }
    // Batch i + 1 is copied to the device and inflated on a stream of its own while batch i is counted on the map's: the
    // inflate stream only waits for the count that last read the buffer it is about to fill (two batches back).
    hipStream_t st_inf = nullptr;
    hipEvent_t counted[2] = {nullptr, nullptr};
    if (rc == TSX_HIP_OK && d_txt[1] &&
        (hipStreamCreateWithFlags(&st_inf, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&counted[0], hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&counted[1], hipEventDisableTiming) != hipSuccess)) {
        g_last_error = "hipStreamCreate for the BGZF inflate stream failed";
        rc = TSX_HIP_EHIP;
    }
    size_t prev_len = 0;   // bytes of text in the previous batch's buffer, behind its head
    int b = 0;
    size_t nbatch = 0;
    for (size_t m0 = 0; m0 < nm && rc == TSX_HIP_OK; b ^= 1, ++nbatch) {
        const size_t m1 = bgzf_next_batch(ix, m0, batch), nb = bgzf_batch_text(ix, m0, m1);
        const bool first = (m0 == 0), last = (m1 == nm);
        uint8_t *buf = d_txt[b];
        hipStream_t sti = st_inf ? st_inf : st;   // (one batch in all: everything on the map's stream)
        if (st_inf && nbatch >= 2 && hipStreamWaitEvent(st_inf, counted[b], 0) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        rc = inflate_batch((const uint8_t *)gz, n, ix, m0, m1, dv, buf + head, sti);   // (returns when the text is there)
        if (rc != TSX_HIP_OK) break;
        if (!first) {   // the end of the text so far (it may reach back into the previous buffer's own head)
            if (hipMemcpyAsync(buf, d_txt[b ^ 1] + prev_len, head, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
            // (the other buffer is free for the next batch's text only behind this copy as well)
            if (st_inf && hipEventRecord(counted[b ^ 1], st) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        }
        if (hipMemsetAsync(buf + head + nb, '\n', 256, st) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        // piece: from `from` (16-byte aligned) to the end of this batch's text; it owns all start positions but the
        // last BGZF_PRE (they belong to the next piece, which sees them again behind its own head)
        const size_t from = first ? head : 16, len = head + nb - from;
        const size_t own = last ? len : (len > BGZF_PRE ? len - BGZF_PRE : 0);
        rc = run_fastq_piece(m, buf + from, len, own, first ? 0 : -1, st);
        if (rc == TSX_HIP_OK && st_inf && hipEventRecord(counted[b], st) != hipSuccess) rc = TSX_HIP_EHIP;
        prev_len = nb;
        m0 = m1;
    }
    hipError_t e = hipStreamSynchronize(st);
    if (st_inf) { (void)hipStreamSynchronize(st_inf); (void)hipStreamDestroy(st_inf); }
    for (hipEvent_t ev : counted) if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(d_txt[0]); (void)hipFree(d_txt[1]);
    if (rc == TSX_HIP_OK && e != hipSuccess) { g_last_error = hipGetErrorString(e); rc = TSX_HIP_EHIP; }
    return rc;
}

static int ensure_staging(tsx_hip_map *m, size_t n) {
    const size_t bytes = std::min(m->piece, n) + STAGE_PAD + 128;
    if (m->stage_bytes >= bytes) return TSX_HIP_OK;
    for (int i = 0; i < 2; ++i) {  // grow: drop the smaller buffers first
        if (m->h_stage[i]) { HIP_TRY(hipHostFree(m->h_stage[i])); m->h_stage[i] = nullptr; }
        if (m->d_stage[i]) { HIP_TRY(hipFree(m->d_stage[i])); m->d_stage[i] = nullptr; }
        if (m->stage_done[i]) { HIP_TRY(hipEventDestroy(m->stage_done[i])); m->stage_done[i] = nullptr; }
        if (m->stage_in[i]) { HIP_TRY(hipEventDestroy(m->stage_in[i])); m->stage_in[i] = nullptr; }
    }
    if (!m->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    m->stage_bytes = 0;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipHostMalloc((void **)&m->h_stage[i], bytes, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&m->d_stage[i], bytes));
        HIP_TRY(hipEventCreateWithFlags(&m->stage_done[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->stage_in[i], hipEventDisableTiming));
    }
    m->stage_bytes = bytes;
    return TSX_HIP_OK;
}

// Pageable -> pinned staging copy on several host threads: one thread moves ~10 GB/s,
// the PCIe link ~55 GB/s.  TSX_HIP_COPY_THREADS overrides the default (hardware threads - 2,
// at most 12).
static void parallel_memcpy(uint8_t *dst, const char *src, size_t len) {
    static unsigned maxt = 0;
    if (!maxt) {
        const unsigned hw = std::thread::hardware_concurrency();
        maxt = std::min(12u, std::max(2u, hw > 2 ? hw - 2 : 2u));
        if (const char *e = getenv("TSX_HIP_COPY_THREADS")) maxt = (unsigned)std::min(64, std::max(1, atoi(e)));
    }
    const size_t MIN_PER_THREAD = (size_t)8 << 20;
    unsigned nthreads = (unsigned)std::min<size_t>(maxt, len / MIN_PER_THREAD);
    if (nthreads <= 1) { memcpy(dst, src, len); return; }
    std::vector<std::thread> th;
    const size_t per = ((len / nthreads) + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < nthreads; ++t) {
        const size_t lo = std::min(len, (size_t)t * per), hi = (t + 1 == nthreads) ? len : std::min(len, lo + per);
        if (hi > lo) th.emplace_back([=]() { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &x : th) x.join();
}

extern "C" int tsx_hip_count_fastq_host(tsx_hip_map *m, const char *text, size_t n) {
    if (!m || (!text && n)) return TSX_HIP_EINVAL;
    if (m->p.lg != m->p.l) return TSX_HIP_EINVAL;   // see tsx_hip_count_fastq_device
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (slab_bits(m) && (m->path == 2 || (m->path == 0 && n * 32 >= m->lay.table_bytes))) {
        // a table built slab by slab walks the whole text once per slab: the text goes to the device in one piece
        uint8_t *d_text = nullptr;
        HIP_TRY(hipMalloc((void **)&d_text, n + 256));
        int rcb = TSX_HIP_OK;
        if (hipMemcpy(d_text, text, n, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_text + n, '\n', 256) != hipSuccess) rcb = TSX_HIP_EHIP;
        if (rcb == TSX_HIP_OK) rcb = tsx_hip_count_fastq_device(m, d_text, n, nullptr);
        if (rcb == TSX_HIP_OK) rcb = tsx_hip_sync(m);
        (void)hipStreamSynchronize(m->stream);
        (void)hipFree(d_text);
        return rcb;
    }
    int rc = ensure_staging(m, n);
    if (rc != TSX_HIP_OK) return rc;
    hipStream_t st = m->stream;
    HIP_TRY(hipMemsetAsync(m->d_carry, 0, 64, st));
    // Pieces own m->piece start positions and carry a k-1 byte halo so that
    // windows beginning near the end of a piece see their last bytes.
    const size_t halo = (size_t)m->p.k - 1;
    int buf = 0;
    bool used[2] = {false, false};
    for (size_t off = 0; off < n; off += m->piece, buf ^= 1) {
        const size_t own = std::min(m->piece, n - off);
        const size_t len = std::min(own + halo, n - off);
        // three legs overlap: this piece's host copy, the previous piece's H2D copy (its own stream)
        // and the kernels of the piece before that
        if (used[buf]) HIP_TRY(hipEventSynchronize(m->stage_done[buf]));
        parallel_memcpy(m->h_stage[buf], text + off, len);
        HIP_TRY(hipMemcpyAsync(m->d_stage[buf], m->h_stage[buf], len, hipMemcpyHostToDevice, m->copy_stream));
        HIP_TRY(hipEventRecord(m->stage_in[buf], m->copy_stream));
        HIP_TRY(hipStreamWaitEvent(st, m->stage_in[buf], 0));
        const int head_open = (off > 0 && text[off - 1] != '\n') ? 1 : 0;
        rc = run_fastq_piece(m, m->d_stage[buf], len, own, head_open, st);
        if (rc != TSX_HIP_OK) return rc;
        HIP_TRY(hipEventRecord(m->stage_done[buf], st));
        used[buf] = true;
    }
    return tsx_hip_sync(m);
}

extern "C" int tsx_hip_add_kmers_device(tsx_hip_map *m, const void *dev_kmers, const void *dev_counts, size_t n,
                                        void *stream) {
    if (!m || (!dev_kmers && n)) return TSX_HIP_EINVAL;
    if (n == 0) return TSX_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    int rcz = ensure_zeroed(m, st);
    if (rcz != TSX_HIP_OK) return rcz;
    const int grid = grid_for(m, n, 8);
    DISPATCH_WK(m, hipLaunchKernelGGL((add_kmers_kernel<WKV>), dim3(grid), dim3(NT), 0, st, m->p,
                                      (const uint64_t *)dev_kmers, (const uint64_t *)dev_counts, (uint64_t)n));
    HIP_TRY(hipGetLastError());
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_add_kmers_host(tsx_hip_map *m, const uint64_t *kmers, const uint64_t *counts, size_t n) {
    if (!m || (!kmers && n)) return TSX_HIP_EINVAL;
    if (n == 0) return TSX_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    uint64_t *dk = nullptr, *dc = nullptr;
    const size_t kb = n * (size_t)m->p.wk * 8;
    HIP_TRY(hipMalloc((void **)&dk, kb));
    int rc = TSX_HIP_OK;
    do {
        if (hipMemcpyAsync(dk, kmers, kb, hipMemcpyHostToDevice, m->stream) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        if (counts) {
            if (hipMalloc((void **)&dc, n * 8) != hipSuccess) { rc = TSX_HIP_ENOMEM; break; }
            if (hipMemcpyAsync(dc, counts, n * 8, hipMemcpyHostToDevice, m->stream) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        }
        rc = tsx_hip_add_kmers_device(m, dk, dc, n, nullptr);
        if (rc == TSX_HIP_OK) rc = tsx_hip_sync(m);
    } while (0);
    (void)hipStreamSynchronize(m->stream);
    (void)hipFree(dk); (void)hipFree(dc);
    return rc;
}

extern "C" int tsx_hip_get_counts_device(tsx_hip_map *m, const void *dev_kmers, size_t n, void *dev_counts_out,
                                         void *stream) {
    if (!m || ((!dev_kmers || !dev_counts_out) && n)) return TSX_HIP_EINVAL;
    if (n == 0) return TSX_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    int rcz = ensure_zeroed(m, st);
    if (rcz != TSX_HIP_OK) return rcz;
    const int grid = grid_for(m, n, 8);
    DISPATCH_WK(m, hipLaunchKernelGGL((get_counts_kernel<WKV>), dim3(grid), dim3(NT), 0, st, m->p,
                                      (const uint64_t *)dev_kmers, (uint64_t)n, (uint64_t *)dev_counts_out,
                                      (uint64_t *)nullptr));
    HIP_TRY(hipGetLastError());
    return TSX_HIP_OK;
}

// Host-side lookups of a handful of k-mers (the reference calls getKmerCount one k-mer at a time,
// main.cpp:285-330) go through a small device scratch kept with the map instead of hipMalloc/hipFree.
static const size_t SMALL_LOOKUP = 256;
static int lookup_host(tsx_hip_map *m, const uint64_t *kmers, size_t n, uint64_t *counts_out, uint64_t *slots_out) {
    if (!m || ((!kmers || !counts_out) && n)) return TSX_HIP_EINVAL;
    if (n == 0) return TSX_HIP_OK;
    HIP_TRY(hipSetDevice(m->device));
    join_foreign(m, false);
    int rcz = ensure_zeroed(m, m->stream);
    if (rcz != TSX_HIP_OK) return rcz;
    const size_t wk = (size_t)m->p.wk, kb = n * wk * 8;
    uint64_t *dk = nullptr, *dc = nullptr, *dp = nullptr;
    const bool small = n <= SMALL_LOOKUP;
    if (small) {
        if (!m->d_small) HIP_TRY(hipMalloc((void **)&m->d_small, SMALL_LOOKUP * (4 + 2) * 8));
        dk = m->d_small; dc = dk + SMALL_LOOKUP * 4; dp = dc + SMALL_LOOKUP;
    } else {
        HIP_TRY(hipMalloc((void **)&dk, kb));
        if (hipMalloc((void **)&dc, n * 16) != hipSuccess) { (void)hipFree(dk); return TSX_HIP_ENOMEM; }
        dp = dc + n;
    }
    int rc = TSX_HIP_OK;
    do {
        if (hipMemcpyAsync(dk, kmers, kb, hipMemcpyHostToDevice, m->stream) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        const int grid = grid_for(m, n, 8);
        DISPATCH_WK(m, hipLaunchKernelGGL((get_counts_kernel<WKV>), dim3(grid), dim3(NT), 0, m->stream, m->p,
                                          (const uint64_t *)dk, (uint64_t)n, dc, slots_out ? dp : (uint64_t *)nullptr));
        if (hipGetLastError() != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        if (hipMemcpyAsync(counts_out, dc, n * 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        if (slots_out && hipMemcpyAsync(slots_out, dp, n * 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess) { rc = TSX_HIP_EHIP; break; }
        if (hipStreamSynchronize(m->stream) != hipSuccess) rc = TSX_HIP_EHIP;
    } while (0);
    if (!small) { (void)hipStreamSynchronize(m->stream); (void)hipFree(dk); (void)hipFree(dc); }
    return rc;
}

extern "C" int tsx_hip_lookup_host(tsx_hip_map *m, const uint64_t *kmers, size_t n, uint64_t *counts_out,
                                   uint64_t *slots_out) {
    if (!slots_out && n) return TSX_HIP_EINVAL;
    return lookup_host(m, kmers, n, counts_out, slots_out);
}

extern "C" int tsx_hip_kmer_starts_host(tsx_hip_map *m, uint8_t *bits_out, size_t nbytes) {
    if (!m || !bits_out || nbytes * 8 < m->lay.slots) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    const uint64_t nb = (m->lay.slots + 7) / 8;
    join_foreign(m, false);
    int rcz = ensure_zeroed(m, m->stream);
    if (rcz != TSX_HIP_OK) return rcz;
    uint8_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, nb));
    hipLaunchKernelGGL(kmer_starts_kernel, dim3(grid_for(m, nb, 8)), dim3(NT), 0, m->stream, m->p, d, nb);
    int rc = (hipGetLastError() == hipSuccess &&
              hipMemcpyAsync(bits_out, d, nb, hipMemcpyDeviceToHost, m->stream) == hipSuccess &&
              hipStreamSynchronize(m->stream) == hipSuccess) ? TSX_HIP_OK : TSX_HIP_EHIP;
    (void)hipFree(d);
    if (rc == TSX_HIP_OK && nbytes > nb) memset(bits_out + nb, 0, nbytes - nb);
    return rc;
}

extern "C" int tsx_hip_get_counts_host(tsx_hip_map *m, const uint64_t *kmers, size_t n, uint64_t *counts_out) {
    return lookup_host(m, kmers, n, counts_out, nullptr);
}

// Diagnostic builds only (TSX_HIP_DEBUG bit 4): the ST_DBG* counters.  Not part of include/tsxcount_hip.h.
extern "C" int tsx_hip_debug_counters(tsx_hip_map *m, uint64_t *out8) {
    if (!m || !out8) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    unsigned long long st[ST_N];
    int rc = read_stats(m, st);
    if (rc != TSX_HIP_OK) return rc;
    for (int i = 0; i < 7; ++i) out8[i] = st[ST_DBG0 + i];
    out8[7] = 0;
    if (m->d_def_n) {   // entries of the deferred list of the last partitioned pass
        unsigned long long dn = 0;
        HIP_TRY(hipMemcpy(&dn, m->d_def_n, 8, hipMemcpyDeviceToHost));
        out8[7] = dn;
    }
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_get_stats(tsx_hip_map *m, tsx_hip_stats *out) {
    if (!m || !out) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    join_foreign(m, false);
    int rcz = ensure_zeroed(m, m->stream);
    if (rcz != TSX_HIP_OK) return rcz;
    HIP_TRY(hipMemsetAsync(m->p.stats + ST_SCRATCH, 0, 2 * sizeof(unsigned long long), m->stream));
    HIP_TRY(hipMemsetAsync(m->p.stats + ST_SCRATCH3, 0, sizeof(unsigned long long), m->stream));
    const int grid = grid_for(m, m->lay.slots, 8);
    hipLaunchKernelGGL(occupied_kernel, dim3(grid), dim3(NT), 0, m->stream, m->p);
    HIP_TRY(hipGetLastError());
    unsigned long long st[ST_N];
    int rc = read_stats(m, st);
    if (rc != TSX_HIP_OK) return rc;
    out->kmers_added = st[ST_KMERS];
    out->insert_failures = st[ST_FAIL];
    out->overflow_carries = st[ST_CARRY];
    out->overflow_failures = st[ST_SECFAIL];
    out->lock_timeouts = st[ST_LOCKTO];
    out->fallback_inserts = st[ST_FALLBACK];
    out->distinct = st[ST_SCRATCH];
    out->overflow_used = st[ST_SCRATCH2];
    out->count_sum = st[ST_SCRATCH3];
    return TSX_HIP_OK;
}

// Entries of the table slots [slot_lo, slot_hi), grouped by owner rank (nranks = 1: plain dump).
static int dump_slots(tsx_hip_map *m, int nranks, uint64_t slot_lo, uint64_t slot_hi, void *dev_kmers_out,
                      void *dev_counts_out, size_t cap, void *dev_seg_counts, void *stream) {
    if (!m || nranks < 1 || nranks > 64 || !dev_kmers_out || !dev_counts_out || !dev_seg_counts) return TSX_HIP_EINVAL;
    if (slot_lo > slot_hi || slot_hi > m->lay.slots) return TSX_HIP_EINVAL;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = pick_stream(m, stream);
    int rcz = ensure_zeroed(m, st);
    if (rcz != TSX_HIP_OK) return rcz;
    const int grid = grid_for(m, std::max<uint64_t>(1, slot_hi - slot_lo), 8);
    unsigned long long *seg = m->d_seg;
    HIP_TRY(hipMemsetAsync(seg, 0, 64 * sizeof(unsigned long long), st));
    DISPATCH_WK(m, hipLaunchKernelGGL((dump_kernel<WKV>), dim3(grid), dim3(NT), 0, st, m->p, nranks, 0,
                                      (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0, seg, slot_lo, slot_hi));
    // segment sizes -> caller; exclusive prefix -> cursors (tiny: done on the host)
    unsigned long long h_seg[64];
    HIP_TRY(hipMemcpyAsync(h_seg, seg, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    unsigned long long cur[64], total = 0;
    for (int r = 0; r < 64; ++r) { cur[r] = total; total += (r < nranks) ? h_seg[r] : 0; }
    if (total > cap) return TSX_HIP_ERANGE;
    HIP_TRY(hipMemcpyAsync(dev_seg_counts, h_seg, (size_t)nranks * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(seg, cur, 64 * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    DISPATCH_WK(m, hipLaunchKernelGGL((dump_kernel<WKV>), dim3(grid), dim3(NT), 0, st, m->p, nranks, 1,
                                      (uint64_t *)dev_kmers_out, (uint64_t *)dev_counts_out, (uint64_t)cap, seg, slot_lo,
                                      slot_hi));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // cur[] lives on this stack frame
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_partition_device(tsx_hip_map *m, int nranks, void *dev_kmers_out, void *dev_counts_out,
                                        size_t cap, void *dev_seg_counts, void *stream) {
    if (!m) return TSX_HIP_EINVAL;
    return dump_slots(m, nranks, 0, m->lay.slots, dev_kmers_out, dev_counts_out, cap, dev_seg_counts, stream);
}

extern "C" int tsx_hip_dump_device(tsx_hip_map *m, void *dev_kmers_out, void *dev_counts_out, size_t cap,
                                   void *dev_n, void *stream) {
    if (!m || !dev_n) return TSX_HIP_EINVAL;
    return dump_slots(m, 1, 0, m->lay.slots, dev_kmers_out, dev_counts_out, cap, dev_n, stream);
}

extern "C" int tsx_hip_dump_range_device(tsx_hip_map *m, uint64_t slot_lo, uint64_t slot_hi, void *dev_kmers_out,
                                         void *dev_counts_out, size_t cap, void *dev_n, void *stream) {
    if (!m || !dev_n) return TSX_HIP_EINVAL;
    return dump_slots(m, 1, slot_lo, slot_hi, dev_kmers_out, dev_counts_out, cap, dev_n, stream);
}

extern "C" int tsx_hip_dump_host(tsx_hip_map *m, uint64_t *kmers_out, uint64_t *counts_out, size_t cap,
                                 size_t *n_out) {
    if (!m || !kmers_out || !counts_out || !n_out) return TSX_HIP_EINVAL;
    tsx_hip_stats s;
    int rc = tsx_hip_get_stats(m, &s);
    if (rc != TSX_HIP_OK) return rc;
    if (s.distinct > cap) return TSX_HIP_ERANGE;
    *n_out = (size_t)s.distinct;
    if (s.distinct == 0) return TSX_HIP_OK;
    uint64_t *dk = nullptr, *dc = nullptr, *dn = nullptr;
    const size_t kb = (size_t)s.distinct * m->p.wk * 8;
    HIP_TRY(hipMalloc((void **)&dk, kb));
    if (hipMalloc((void **)&dc, s.distinct * 8) != hipSuccess) { (void)hipFree(dk); return TSX_HIP_ENOMEM; }
    if (hipMalloc((void **)&dn, 64) != hipSuccess) { (void)hipFree(dk); (void)hipFree(dc); return TSX_HIP_ENOMEM; }
    rc = tsx_hip_dump_device(m, dk, dc, (size_t)s.distinct, dn, nullptr);
    if (rc == TSX_HIP_OK) {
        if (hipMemcpy(kmers_out, dk, kb, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(counts_out, dc, s.distinct * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = TSX_HIP_EHIP;
    }
    (void)hipFree(dk); (void)hipFree(dc); (void)hipFree(dn);
    return rc;
}

extern "C" int tsx_hip_owner_host(const tsx_hip_map *m, const uint64_t *kmer, int nranks) {
    if (!m || !kmer || nranks < 1) return TSX_HIP_EINVAL;
    uint64_t z = 0x243F6A8885A308D3ULL;
    for (int t = 0; t < m->p.wk; ++t) {
        uint64_t x = (t == m->p.wk - 1) ? (kmer[t] & m->p.top_mask) : kmer[t];
        z ^= x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
    }
    return (int)((z >> 32) * (uint64_t)nranks >> 32);
}

extern "C" int tsx_hip_hash_apply(const tsx_hip_map *m, const uint64_t *kmer, uint64_t *key_out) {
    if (!m || !kmer || !key_out) return TSX_HIP_EINVAL;
    std::vector<uint64_t> x(kmer, kmer + m->p.wk);
    x[m->p.wk - 1] &= m->p.top_mask;
    apply_rows(m, m->rows, x.data(), key_out);
    return TSX_HIP_OK;
}
extern "C" int tsx_hip_hash_invert(const tsx_hip_map *m, const uint64_t *key, uint64_t *kmer_out) {
    if (!m || !key || !kmer_out) return TSX_HIP_EINVAL;
    std::vector<uint64_t> x(key, key + m->p.wk);
    x[m->p.wk - 1] &= m->p.top_mask;
    apply_rows(m, m->irows, x.data(), kmer_out);
    return TSX_HIP_OK;
}
extern "C" int tsx_hip_hash_rows(const tsx_hip_map *m, uint64_t *rows_out) {
    if (!m || !rows_out) return TSX_HIP_EINVAL;
    memcpy(rows_out, m->rows.data(), m->rows.size() * 8);
    return TSX_HIP_OK;
}

// ---- synthetic reads -----------------------------------------------------------
static inline int dec_digits(uint64_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

extern "C" int tsx_hip_synth_fastq_device(uint64_t seed, uint64_t first_read, uint64_t n_reads, int k,
                                          void *dev_out, size_t cap, uint64_t *bytes_out, uint64_t *kmers_out,
                                          uint64_t *polya_kmers_out, int device, void *stream) {
    if (k < 1 || k > 127) return TSX_HIP_EINVAL;
    std::vector<uint64_t> offs(n_reads + 1);
    uint64_t total = 0, kmers = 0, polya = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        const uint64_t id = first_read + r;
        const uint32_t nrand = 500u + (uint32_t)(synth_mix(seed, id, 0) % 501u);
        const uint32_t na = 100u + (uint32_t)(synth_mix(seed, id, 1) % 201u);
        const uint64_t L = (uint64_t)nrand + na;
        offs[r] = total;
        total += 4 + dec_digits(id) + 1 + 2 * L + 4;
        if (L >= (uint64_t)k) kmers += L - k + 1;
        if (polya_kmers_out) {
            // all-'A' windows: runs of A inside the random part plus the tail
            uint32_t run = 0;
            for (uint32_t j = 0; j < nrand; ++j) {
                const uint64_t w = synth_mix(seed, id, 2 + (j >> 5));
                if (((w >> (2 * (j & 31))) & 3) == 0) ++run;
                else { if (run >= (uint32_t)k) polya += run - k + 1; run = 0; }
            }
            run += na;
            if (run >= (uint32_t)k) polya += run - k + 1;
        }
    }
    offs[n_reads] = total;
    if (bytes_out) *bytes_out = total;
    if (kmers_out) *kmers_out = kmers;
    if (polya_kmers_out) *polya_kmers_out = polya;
    if (!dev_out) return TSX_HIP_OK;
    if (total > cap) return TSX_HIP_ERANGE;
    if (n_reads == 0) return TSX_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TSX_HIP_ENODEVICE;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    uint64_t *d_off = nullptr;
    HIP_TRY(hipMalloc((void **)&d_off, (n_reads + 1) * 8));
    int rc = TSX_HIP_OK;
    if (hipMemcpyAsync(d_off, offs.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess) rc = TSX_HIP_EHIP;
    if (rc == TSX_HIP_OK) {
        const int grid = (int)std::min<uint64_t>(n_reads, 65536);
        hipLaunchKernelGGL(synth_fill_kernel, dim3(grid), dim3(NT), 0, st, seed, first_read, n_reads,
                           (const uint64_t *)d_off, (uint8_t *)dev_out);
        if (hipGetLastError() != hipSuccess) rc = TSX_HIP_EHIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) rc = TSX_HIP_EHIP;
    (void)hipFree(d_off);
    return rc;
}

// Zipf-skewed reads written straight into device memory (BASELINE config 4; see synth_zipf_kernel).  thr: n_templates
// ascending uint64 thresholds (host).  Sizing call: dev_out == NULL.
extern "C" int tsx_hip_synth_zipf_device(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint32_t n_templates,
                                         const uint64_t *thr, void *dev_out, size_t cap, uint64_t *bytes_out, int device,
                                         void *stream) {
    if (read_len < 1 || read_len > (1u << 20) || n_templates < 1 || !thr) return TSX_HIP_EINVAL;
    std::vector<uint64_t> offs(n_reads + 1);
    uint64_t total = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        offs[r] = total;
        total += 2 + dec_digits(r) + 1 + 2 * (uint64_t)read_len + 4;
    }
    offs[n_reads] = total;
    if (bytes_out) *bytes_out = total;
    if (!dev_out) return TSX_HIP_OK;
    if (total > cap) return TSX_HIP_ERANGE;
    if (n_reads == 0) return TSX_HIP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TSX_HIP_ENODEVICE;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    uint64_t *d_off = nullptr, *d_thr = nullptr;
    HIP_TRY(hipMalloc((void **)&d_off, (n_reads + 1) * 8));
    if (hipMalloc((void **)&d_thr, (size_t)n_templates * 8) != hipSuccess) { (void)hipFree(d_off); return TSX_HIP_ENOMEM; }
    int rc = TSX_HIP_OK;
    if (hipMemcpyAsync(d_off, offs.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_thr, thr, (size_t)n_templates * 8, hipMemcpyHostToDevice, st) != hipSuccess) rc = TSX_HIP_EHIP;
    if (rc == TSX_HIP_OK) {
        const int grid = (int)std::min<uint64_t>(n_reads, 65536);
        hipLaunchKernelGGL(synth_zipf_kernel, dim3(grid), dim3(NT), 0, st, seed, n_reads, read_len, n_templates,
                           (const uint64_t *)d_thr, (const uint64_t *)d_off, (uint8_t *)dev_out);
        if (hipGetLastError() != hipSuccess) rc = TSX_HIP_EHIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) rc = TSX_HIP_EHIP;
    (void)hipFree(d_off); (void)hipFree(d_thr);
    return rc;
}
