// tsx_inflate.h -- device-side inflate for blocked gzip (BGZF) input.
//
// The reference reads `.gz` input through zlib (`gzopen`/`gzgets`, src/fastxutils/FastXReader.h:178-206), one
// stream, one thread.  A deflate stream is serial by construction; what a GPU can use is the way sequencing
// data is actually shipped: BGZF (`bgzip`, the format of BAM and of most compressed FASTQ), a gzip file made
// of independent members of at most 64 KiB each, every member carrying its compressed size in a `BC` extra
// field (so the members can be found without decoding) and its CRC-32 and uncompressed size in its trailer.
// zlib reads such a file as ordinary multi-member gzip; here every member is inflated by ONE LANE
// (RFC 1951: stored, fixed and dynamic Huffman blocks), 64 members per workgroup, the decode tables of a lane
// in LDS (interleaved by lane: conflict-free), the output written straight into the text buffer in HBM that
// the scan kernels read.  The CRC-32 of every member is checked on the device as well.
// What shapes the kernel: the 64 lanes of a wave are in 64 unrelated streams, a wave pays for every path any of
// its lanes takes, and s_waitcnt counts per wave -- so the symbol loop is three phases that every lane passes
// once per iteration (decode one symbol without branches on its bits / land the one load in flight / store or
// issue the next load), see inflate_members_kernel.  The kernel's time is the latency chain of one lane through
// its member (about 20 ms for 64 KiB), whatever the number of members.
// A `.gz` file that is not BGZF is not handled here (tsx_hip_bgzf_index_host says so); the CLI then falls
// back to zlib on the host.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace tsx {

constexpr int INF_NT = 64;         // one wave; lane = member
constexpr int INF_MAXBITS = 15;
constexpr int INF_LCODES = 288, INF_DCODES = 30;

constexpr size_t INF_LDS_BYTES = 8 * 256 * 4 + (size_t)(2 * (INF_MAXBITS + 1) + INF_LCODES + INF_DCODES) * INF_NT * 2 +
                                 (size_t)(INF_LCODES + INF_DCODES + 2) * INF_NT;

enum InflateStatus : uint32_t {
    INF_OK = 0, INF_ETRUNC = 1, INF_EBLOCK = 2, INF_ESTORED = 3, INF_ELENGTHS = 4, INF_ESYMBOL = 5,
    INF_EOUTPUT = 6, INF_ESIZE = 7, INF_ECRC = 8
};

// LSB-first bit reader (RFC 1951 section 3.1.1).  refill() tops the 64-bit buffer up to 56..63 bits from the eight
// bytes at p and moves p by the whole bytes that fitted (the rest is read again next time: or-ing the same bits in
// twice does no harm).  The eight bytes were loaded by the refill BEFORE: s_waitcnt counts per wave, not per lane, so
// a load that is waited for where it is issued -- or at the next of several refill sites -- costs all 64 lanes a
// memory round trip each time.  One refill at the top of an iteration covers the iteration (a match takes at most
// 15 + 5 + 15 + 13 = 48 bits); get() and the decoder refill only when fewer than 32 bits are left (block headers,
// stored blocks).  The reader runs up to 8 bytes past the member's data (the caller pads the buffer); `avail` counts
// the bits that belong to the member, a read beyond them fails as a truncated stream does.
struct InfBits {
    const uint8_t *p;     // the first byte not (completely) in buf
    uint64_t buf, nxt;    // nxt: the eight bytes at p
    uint32_t cnt, avail;
    bool fail;
    __device__ __forceinline__ void init(const uint8_t *start, uint32_t nbytes) {
        p = start;
        buf = 0; cnt = 0; fail = false;
        avail = nbytes * 8u;
        __builtin_memcpy(&nxt, p, 8);
    }
    __device__ __forceinline__ void refill() {
        buf |= nxt << cnt;
        const uint32_t adv = (63u - cnt) >> 3;
        p += adv;
        cnt += adv * 8u;
        __builtin_memcpy(&nxt, p, 8);
    }
    __device__ __forceinline__ void need32() { if (cnt < 32u) refill(); }
    __device__ __forceinline__ void used(uint32_t n) {   // n bits of buf consumed
        if (n > avail) { fail = true; avail = 0; } else avail -= n;
    }
    __device__ __forceinline__ uint32_t get(uint32_t n) {   // n <= 16
        need32();
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        used(n);
        return fail ? 0u : v;
    }
};

// A lane's canonical Huffman code in LDS: count[len] codes of every length, symbols ordered by code.
// Element i of a lane's array sits at [i * INF_NT + lane].
struct InfCode {
    uint16_t *count, *symbol;
};

// Decoding without a loop over the bits.  In a canonical code the codes of length len are the integers
// first[len] .. first[len] + count[len] - 1, first[len + 1] = (first[len] + count[len]) << 1 (RFC 1951 section 3.2.2).
// Read the next 15 stream bits as a 15-bit number v with the FIRST bit on top: a code of length len is a prefix of v
// exactly when v < lim[len] = (first[len] + count[len]) << (15 - len) and v >= lim[len - 1].  The lim[] do not fall,
// so the length of the code in front is 1 + the number of lim[] that v has reached: fifteen independent compares on
// registers instead of fifteen rounds of shift / compare / branch in which every lane of the wave leaves at another
// round.  The symbol is symbol[(v >> (15 - len)) + delta[len]], delta[len] = (symbols with shorter codes) - first[len].
struct InfFast { uint32_t lim[INF_MAXBITS]; };
// Builds lim[] in registers from the counts inf_construct left in LDS and overwrites count[1..15] with delta[].
__device__ __forceinline__ InfFast inf_fast(const InfCode &h) {
    InfFast f;
    uint32_t first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= INF_MAXBITS; ++len) {
        const uint32_t count = h.count[len * INF_NT];
        f.lim[len - 1] = (first + count) << (INF_MAXBITS - len);
        h.count[len * INF_NT] = (uint16_t)(index - first);
        index += count;
        first = (first + count) << 1;
    }
    return f;
}
__device__ __forceinline__ int inf_decode(InfBits &b, const InfCode &h, const InfFast &f) {
    b.need32();
    const uint32_t v = __brev((uint32_t)b.buf) >> (32 - INF_MAXBITS);
    uint32_t n = 0;
#pragma unroll
    for (int len = 1; len <= INF_MAXBITS; ++len) n += (v >= f.lim[len - 1]) ? 1u : 0u;
    if (n >= (uint32_t)INF_MAXBITS) return -1;   // no code starts like this (an incomplete code)
    const uint32_t len = n + 1u;
    const uint32_t idx = ((v >> (INF_MAXBITS - len)) + (uint32_t)h.count[len * INF_NT]) & 0xFFFFu;
    b.buf >>= len;
    b.cnt -= len;
    b.used(len);
    if (b.fail) return -1;
    return (int)h.symbol[idx * INF_NT];
}

// Build count[]/symbol[] from n code lengths (lengths[i * INF_NT]).  Returns < 0 for an over-subscribed set,
// > 0 for an incomplete one (allowed for a single distance code, as zlib allows it), 0 for a complete one.
__device__ inline int inf_construct(const InfCode &h, const uint8_t *lengths, int n) {
    for (int len = 0; len <= INF_MAXBITS; ++len) h.count[len * INF_NT] = 0;
    for (int s = 0; s < n; ++s) {
        const uint32_t l = lengths[s * INF_NT];
        h.count[l * INF_NT] = (uint16_t)(h.count[l * INF_NT] + 1);
    }
    if (h.count[0] == n) return 0;   // no codes at all: complete, but any decode will fail
    int left = 1;
    for (int len = 1; len <= INF_MAXBITS; ++len) {
        left <<= 1;
        left -= (int)h.count[len * INF_NT];
        if (left < 0) return left;
    }
    uint16_t offs[INF_MAXBITS + 1];
    offs[1] = 0;
    for (int len = 1; len < INF_MAXBITS; ++len) offs[len + 1] = (uint16_t)(offs[len] + h.count[len * INF_NT]);
    for (int s = 0; s < n; ++s) {
        const uint32_t l = lengths[s * INF_NT];
        if (l != 0) {
            // (dynamic index into a 16-entry array: lives in scratch, touched only while tables are built)
            h.symbol[(uint32_t)offs[l] * INF_NT] = (uint16_t)s;
            offs[l] = (uint16_t)(offs[l] + 1);
        }
    }
    return left;
}

__device__ __constant__ const uint8_t INF_CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// One member per lane.  in_off/in_len: the raw deflate data of member i inside gz; out_off/out_len: where its
// ISIZE bytes go; crc: the CRC-32 of its trailer.  status[i] = InflateStatus.
__global__ __launch_bounds__(INF_NT) void inflate_members_kernel(const uint8_t *gz, const uint64_t *in_off,
                                                                 const uint32_t *in_len, const uint64_t *out_off,
                                                                 const uint32_t *out_len, const uint32_t *crc,
                                                                 uint32_t nmem, uint8_t *out, uint32_t *status,
                                                                 const uint32_t *crc_table) {
    extern __shared__ uint32_t s_inf[];   // INF_LDS_BYTES: CRC table | count and symbol tables of both codes | code lengths
    uint32_t *s_crc = s_inf;
    uint16_t *s_lcount = reinterpret_cast<uint16_t *>(s_crc + 8 * 256);
    uint16_t *s_lsym = s_lcount + (INF_MAXBITS + 1) * INF_NT;
    uint16_t *s_dcount = s_lsym + INF_LCODES * INF_NT;
    uint16_t *s_dsym = s_dcount + (INF_MAXBITS + 1) * INF_NT;
    uint8_t *s_len = reinterpret_cast<uint8_t *>(s_dsym + INF_DCODES * INF_NT);
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 8 * 256; i += INF_NT) s_crc[i] = crc_table[i];
    __syncthreads();
    const uint32_t mem = blockIdx.x * INF_NT + lane;
    if (mem >= nmem) return;
    const InfCode lc = {s_lcount + lane, s_lsym + lane}, dc = {s_dcount + lane, s_dsym + lane};
    uint8_t *lengths = s_len + lane;
    InfBits b;
    b.init(gz + in_off[mem], in_len[mem]);
    uint8_t *o = out + out_off[mem];
    const uint32_t olen = out_len[mem];
    uint32_t pos = 0, err = INF_OK;
    // Literals wait in a register, eight to a store: s_waitcnt counts stores as well as loads, so an iteration that
    // stored a byte made the next one wait for that store to land -- a memory round trip per literal.  o[pos - lcnt ..
    // pos) is what waits in lbuf; it is stored before anything reads the output (a match, the CRC) and at the end of
    // a block.  A store of fewer than eight literals writes eight bytes all the same (zeros behind them, in places of
    // this member that later output overwrites), byte by byte only at the very end of the member.
    uint64_t lbuf = 0;
    uint32_t lcnt = 0;
    auto lflush = [&]() {
        if (lcnt == 0u) return;
        if (pos - lcnt + 8u <= olen) __builtin_memcpy(o + pos - lcnt, &lbuf, 8);
        else for (uint32_t i = 0; i < lcnt; ++i) o[pos - lcnt + i] = (uint8_t)(lbuf >> (8u * i));
        lbuf = 0; lcnt = 0;
    };
    // every loop below either consumes input bits or produces output bytes; `budget` bounds the whole member
    // anyway (a corrupt stream must never keep a wave alive)
    uint32_t budget = olen + 8u * in_len[mem] + 1024u;
    for (bool last = false; !last && err == INF_OK;) {
        last = b.get(1) != 0;
        const uint32_t type = b.get(2);
        if (b.fail) { err = INF_ETRUNC; break; }
        if (type == 0) {   // stored: skip to the byte boundary, LEN, NLEN, LEN bytes
            lflush();
            b.used(b.cnt & 7u);
            b.buf >>= (b.cnt & 7u);
            b.cnt &= ~7u;
            const uint32_t len = b.get(16), nlen = b.get(16);
            if (b.fail) { err = INF_ETRUNC; break; }
            if ((len ^ 0xFFFFu) != nlen) { err = INF_ESTORED; break; }
            if (pos + len > olen) { err = INF_EOUTPUT; break; }
            for (uint32_t i = 0; i < len; ++i) {
                const uint32_t v = b.get(8);
                if (b.fail) break;
                o[pos++] = (uint8_t)v;
            }
            if (b.fail) { err = INF_ETRUNC; break; }
            continue;
        }
        if (type == 3) { err = INF_EBLOCK; break; }
        if (type == 1) {   // fixed codes (RFC 1951 section 3.2.6)
            for (int s = 0; s < 144; ++s) lengths[s * INF_NT] = 8;
            for (int s = 144; s < 256; ++s) lengths[s * INF_NT] = 9;
            for (int s = 256; s < 280; ++s) lengths[s * INF_NT] = 7;
            for (int s = 280; s < INF_LCODES; ++s) lengths[s * INF_NT] = 8;
            inf_construct(lc, lengths, INF_LCODES);
            for (int s = 0; s < INF_DCODES; ++s) lengths[s * INF_NT] = 5;
            inf_construct(dc, lengths, INF_DCODES);
        } else {           // dynamic codes (section 3.2.7)
            const uint32_t nlen = b.get(5) + 257, ndist = b.get(5) + 1, ncode = b.get(4) + 4;
            if (b.fail) { err = INF_ETRUNC; break; }
            if (nlen > (uint32_t)INF_LCODES || ndist > (uint32_t)INF_DCODES) { err = INF_ELENGTHS; break; }
            for (uint32_t i = 0; i < 19; ++i) lengths[(uint32_t)INF_CLORDER[i] * INF_NT] = (i < ncode) ? (uint8_t)b.get(3) : 0;
            if (b.fail) { err = INF_ETRUNC; break; }
            if (inf_construct(lc, lengths, 19) != 0) { err = INF_ELENGTHS; break; }   // the code-length code must be complete
            const InfFast cc = inf_fast(lc);
            uint32_t idx = 0;
            while (idx < nlen + ndist && err == INF_OK) {
                const int sym = inf_decode(b, lc, cc);
                if (sym < 0) { err = b.fail ? INF_ETRUNC : INF_ESYMBOL; break; }
                if (sym < 16) { lengths[(idx++) * INF_NT] = (uint8_t)sym; continue; }
                uint32_t prev = 0, rep;
                if (sym == 16) {
                    if (idx == 0) { err = INF_ELENGTHS; break; }
                    prev = lengths[(idx - 1) * INF_NT];
                    rep = 3 + b.get(2);
                } else if (sym == 17) rep = 3 + b.get(3);
                else rep = 11 + b.get(7);
                if (b.fail) { err = INF_ETRUNC; break; }
                if (idx + rep > nlen + ndist) { err = INF_ELENGTHS; break; }
                while (rep--) lengths[(idx++) * INF_NT] = (uint8_t)prev;
            }
            if (err != INF_OK) break;
            if (lengths[256 * INF_NT] == 0) { err = INF_ELENGTHS; break; }   // no end-of-block code
            // the distance lengths follow the literal/length lengths in the same array
            const int rl = inf_construct(lc, lengths, (int)nlen);
            if (rl < 0 || (rl > 0 && nlen - (uint32_t)lc.count[0] != 1)) { err = INF_ELENGTHS; break; }
            const int rd = inf_construct(dc, lengths + nlen * INF_NT, (int)ndist);
            if (rd < 0 || (rd > 0 && ndist - (uint32_t)dc.count[0] != 1)) { err = INF_ELENGTHS; break; }
        }
        // ---- the compressed data of the block.
        // The 64 lanes of the wave are at 64 different places of 64 different streams, and the wave pays for every
        // path any of its lanes takes: with "decode a symbol, then copy the whole match" an iteration cost the
        // longest copy loop among the lanes plus every special case some lane was in (about five memory round trips
        // per symbol).  So an iteration of the wave has three phases, each at most once:
        //   A  decode: a lane that has no match in progress decodes ONE symbol (LDS and registers only);
        //   B  land:   the ONE load a lane issued in the previous iteration is waited for here -- behind the decode, the
        //              round trip runs while the next symbol is decoded -- and stored (copy) or turned into `pat`;
        //   C  act:    store the literal, or do one step of the match: issue the load for 8 bytes of a copy (stored in
        //              the next B), store up to 64 bytes of a repeating pattern, ...
        // Output is written in order: the store of B may run up to 7 bytes past its match (bytes of this member that
        // later output overwrites), so nothing of C may be stored before it.
        const InfFast lcn = inf_fast(lc), dcn = inf_fast(dc);
        uint32_t clen = 0, cdist = 0;
        uint64_t pat = 0;        // distances below 8: the bytes to repeat, replicated to 8 bytes
        uint64_t pw = 0;         // the load in flight
        uint32_t pend = 0;       // 0 none, 1 copy of 8 bytes to o + ppos, 2 the 8 bytes in front of a pattern match
        uint32_t ppos = 0;
        bool have_pat = false;
        for (bool eob = false; !eob;) {
            if (budget-- == 0) { err = INF_EOUTPUT; break; }
            int lit = -1;
            if (clen == 0) {   // ---- A
                b.refill();   // the one refill of the iteration: >= 56 bits
                const int sym = inf_decode(b, lc, lcn);
                if (sym < 0) { err = b.fail ? INF_ETRUNC : INF_ESYMBOL; break; }
                if (sym < 256) {
                    if (pos >= olen) { err = INF_EOUTPUT; break; }
                    lit = sym;
                } else if (sym == 256) {
                    eob = true;
                } else {
                    const uint32_t ls = (uint32_t)sym - 257u;
                    if (ls >= 29u) { err = INF_ESYMBOL; break; }
                    // base and extra bits of a length / distance symbol by arithmetic (RFC 1951 section 3.2.5: after
                    // the first eight lengths / four distances the number of extra bits grows by one every four / two
                    // symbols) -- a lookup in a table in memory is a load the next step waits for, twice per match
                    uint32_t lext = 0, lbase = 3u + ls;
                    if (ls >= 8u) { lext = (ls - 4u) >> 2; lbase = 3u + ((4u + (ls & 3u)) << lext); }
                    if (ls == 28u) { lext = 0; lbase = 258u; }
                    const uint32_t len = lbase + b.get(lext);
                    const int ds = inf_decode(b, dc, dcn);
                    if (ds < 0 || ds >= INF_DCODES) { err = b.fail ? INF_ETRUNC : INF_ESYMBOL; break; }
                    uint32_t dext = 0, dbase = 1u + (uint32_t)ds;
                    if (ds >= 4) { dext = ((uint32_t)ds - 2u) >> 1; dbase = 1u + ((2u + ((uint32_t)ds & 1u)) << dext); }
                    const uint32_t dist = dbase + b.get(dext);
                    if (b.fail) { err = INF_ETRUNC; break; }
                    if (dist > pos || pos + len > olen) { err = INF_EOUTPUT; break; }
                    lflush();   // (no copy is in flight here: the last one landed in B before the literals were decoded)
                    clen = len;
                    cdist = dist;
                    have_pat = false;
                }
            }
            // ---- B.  The wait is spelled out, for all lanes: with a load possibly in flight the compiler guards every
            // later write of a register it shares with `pw` by a wait of its own -- between the pattern stores of C that
            // was a memory round trip PER STORE.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            if (pend == 1u) {
                __builtin_memcpy(o + ppos, &pw, 8);
            } else if (pend == 2u) {   // byte j of pat = pattern[j mod dist], the pattern = the last `dist` bytes of the output
                uint64_t w = pw >> (8u * (8u - cdist));
                for (uint32_t have = cdist; have < 8u; have *= 2u) w |= w << (8u * have);
                pat = w;
                have_pat = true;
            }
            pend = 0;
            // ---- C: what stores first, what issues the iteration's load last
            const bool small = cdist < 8u;
            uint32_t todo = 0;   // 1: load for a copy of 8 bytes, 2: load of the 8 bytes in front of a pattern match
            if (lit >= 0) {
                lbuf |= (uint64_t)(uint32_t)lit << (8u * lcnt);
                ++lcnt; ++pos;
                if (lcnt == 8u) { __builtin_memcpy(o + pos - 8u, &lbuf, 8); lbuf = 0; lcnt = 0; }
            } else if (clen != 0) {
                if (small && have_pat && pos + 64u <= olen) {   // up to eight stores, no load: long runs in few iterations
                    const uint32_t step = (8u / cdist) * cdist;   // whole periods per store
                    const uint32_t n = min(clen, 8u * step);
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j)
                        if (j * step < n) __builtin_memcpy(o + pos + j * step, &pat, 8);
                    pos += n;
                    clen -= n;
                } else if (small && !have_pat && pos >= 8u) {
                    todo = 2u;
                } else if (!small && cdist >= 32u && clen >= 32u) {   // source and destination of a 32-byte step cannot overlap
                    uint64_t w[4];
                    __builtin_memcpy(w, o + pos - cdist, 32);
                    __builtin_memcpy(o + pos, w, 32);
                    pos += 32u;
                    clen -= 32u;
                } else if (!small && pos + 8u <= olen) {
                    todo = 1u;
                } else {   // the first or last bytes of the member
                    for (; clen; --clen, ++pos) o[pos] = o[pos - cdist];
                }
            }
            if (todo) {   // ONE load instruction for both kinds (two would wait for each other: same destination register)
                __builtin_memcpy(&pw, o + pos - (todo == 1u ? cdist : 8u), 8);
                pend = todo;
                if (todo == 1u) {
                    ppos = pos;
                    const uint32_t n = min(clen, 8u);
                    pos += n;
                    clen -= n;
                }
            }
        }
        if (pend == 1u) __builtin_memcpy(o + ppos, &pw, 8);   // (left by a break)
        lflush();
    }
    if (err == INF_OK && pos != olen) err = INF_ESIZE;
    if (err == INF_OK) {
        // CRC-32 of the member (gzip trailer).  Byte by byte -- load, look up, xor, each waiting for the one before --
        // this was most of the kernel (a memory round trip per byte).  Eight bytes per step instead: the text comes
        // back 32 bytes per load with the next 32 already on their way, and the eight table lookups of a step
        // (tables for a byte followed by 0..7 zero bytes) do not depend on each other.
        uint32_t c = 0xFFFFFFFFu, i = 0;
        auto step8 = [&](uint64_t w) {
            const uint32_t lo = (uint32_t)w ^ c, hi = (uint32_t)(w >> 32);
            c = s_crc[7 * 256 + (lo & 0xFFu)] ^ s_crc[6 * 256 + ((lo >> 8) & 0xFFu)] ^ s_crc[5 * 256 + ((lo >> 16) & 0xFFu)] ^
                s_crc[4 * 256 + (lo >> 24)] ^ s_crc[3 * 256 + (hi & 0xFFu)] ^ s_crc[2 * 256 + ((hi >> 8) & 0xFFu)] ^
                s_crc[1 * 256 + ((hi >> 16) & 0xFFu)] ^ s_crc[hi >> 24];
        };
        if (olen >= 32u) {
            uint64_t w[4], nx[4];
            __builtin_memcpy(nx, o, 32);
            for (; i + 32u <= olen; i += 32u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = nx[j];
                if (i + 64u <= olen) __builtin_memcpy(nx, o + i + 32u, 32);
#pragma unroll
                for (int j = 0; j < 4; ++j) step8(w[j]);
            }
        }
        for (; i < olen; ++i) c = s_crc[(c ^ o[i]) & 0xFFu] ^ (c >> 8);
        if ((c ^ 0xFFFFFFFFu) != crc[mem]) err = INF_ECRC;
    }
    status[mem] = err;
}

}  // namespace tsx
