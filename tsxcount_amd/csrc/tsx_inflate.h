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
// A `.gz` file that is not BGZF is not handled here (tsx_hip_bgzf_index_host says so); the CLI then falls
// back to zlib on the host.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace tsx {

constexpr int INF_NT = 64;         // one wave; lane = member
constexpr int INF_MAXBITS = 15;
constexpr int INF_LCODES = 288, INF_DCODES = 30;

constexpr size_t INF_LDS_BYTES = 256 * 4 + (size_t)(2 * (INF_MAXBITS + 1) + INF_LCODES + INF_DCODES) * INF_NT * 2 +
                                 (size_t)(INF_LCODES + INF_DCODES + 2) * INF_NT;

enum InflateStatus : uint32_t {
    INF_OK = 0, INF_ETRUNC = 1, INF_EBLOCK = 2, INF_ESTORED = 3, INF_ELENGTHS = 4, INF_ESYMBOL = 5,
    INF_EOUTPUT = 6, INF_ESIZE = 7, INF_ECRC = 8
};

// LSB-first bit reader over [p, end) (RFC 1951 section 3.1.1); four bytes per refill where they exist
struct InfBits {
    const uint8_t *p, *end;
    uint64_t buf;
    uint32_t cnt;
    bool fail;
    __device__ __forceinline__ void refill() {
        if (cnt <= 32u) {
            if (p + 4 <= end) {
                uint32_t w;
                __builtin_memcpy(&w, p, 4);
                buf |= (uint64_t)w << cnt;
                cnt += 32u;
                p += 4;
            } else {
                while (cnt <= 56u && p < end) { buf |= (uint64_t)(*p++) << cnt; cnt += 8u; }
            }
        }
    }
    __device__ __forceinline__ uint32_t get(uint32_t n) {   // n <= 16
        refill();
        if (cnt < n) { fail = true; return 0; }
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        return v;
    }
};

// A lane's canonical Huffman code in LDS: count[len] codes of every length, symbols ordered by code.
// Element i of a lane's array sits at [i * INF_NT + lane].
struct InfCode {
    uint16_t *count, *symbol;
};

// The 16 counts of a code, two per register: the decode loop below is unrolled, so they stay in registers and a
// symbol costs ONE LDS read (the symbol itself) instead of one per bit of its code.
struct InfCounts { uint32_t w[8]; };
__device__ __forceinline__ InfCounts inf_counts(const InfCode &h) {
    InfCounts c;
#pragma unroll
    for (int i = 0; i < 8; ++i) c.w[i] = (uint32_t)h.count[(2 * i) * INF_NT] | ((uint32_t)h.count[(2 * i + 1) * INF_NT] << 16);
    return c;
}

// Decode one symbol, bit by bit (at most 15 steps): the codes of one length are consecutive integers, the
// first code of length len+1 is (first code of len + count[len]) << 1 (RFC 1951 section 3.2.2).
__device__ __forceinline__ int inf_decode(InfBits &b, const InfCode &h, const InfCounts &c) {
    b.refill();
    uint32_t code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    uint32_t left = b.cnt;
#pragma unroll
    for (uint32_t len = 1; len <= (uint32_t)INF_MAXBITS; ++len) {
        if (left == 0) { b.fail = true; return -1; }
        code |= (uint32_t)bits & 1u;
        bits >>= 1;
        --left;
        const uint32_t count = (c.w[len >> 1] >> ((len & 1u) * 16u)) & 0xFFFFu;
        if (code < first + count) {
            b.buf = bits;
            b.cnt = left;
            return (int)h.symbol[(index + (code - first)) * INF_NT];
        }
        index += count;
        first = (first + count) << 1;
        code <<= 1;
    }
    return -1;
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

__device__ __constant__ const uint16_t INF_LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                                         35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ __constant__ const uint8_t INF_LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                                       3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ __constant__ const uint16_t INF_DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                                         257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                                         8193, 12289, 16385, 24577};
__device__ __constant__ const uint8_t INF_DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
                                                       7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
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
    uint16_t *s_lcount = reinterpret_cast<uint16_t *>(s_crc + 256);
    uint16_t *s_lsym = s_lcount + (INF_MAXBITS + 1) * INF_NT;
    uint16_t *s_dcount = s_lsym + INF_LCODES * INF_NT;
    uint16_t *s_dsym = s_dcount + (INF_MAXBITS + 1) * INF_NT;
    uint8_t *s_len = reinterpret_cast<uint8_t *>(s_dsym + INF_DCODES * INF_NT);
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += INF_NT) s_crc[i] = crc_table[i];
    __syncthreads();
    const uint32_t mem = blockIdx.x * INF_NT + lane;
    if (mem >= nmem) return;
    const InfCode lc = {s_lcount + lane, s_lsym + lane}, dc = {s_dcount + lane, s_dsym + lane};
    uint8_t *lengths = s_len + lane;
    InfBits b;
    b.p = gz + in_off[mem];
    b.end = b.p + in_len[mem];
    b.buf = 0; b.cnt = 0; b.fail = false;
    uint8_t *o = out + out_off[mem];
    const uint32_t olen = out_len[mem];
    uint32_t pos = 0, err = INF_OK;
    // every loop below either consumes input bits or produces output bytes; `budget` bounds the whole member
    // anyway (a corrupt stream must never keep a wave alive)
    uint32_t budget = olen + 8u * in_len[mem] + 1024u;
    for (bool last = false; !last && err == INF_OK;) {
        last = b.get(1) != 0;
        const uint32_t type = b.get(2);
        if (b.fail) { err = INF_ETRUNC; break; }
        if (type == 0) {   // stored: skip to the byte boundary, LEN, NLEN, LEN bytes
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
            const InfCounts cc = inf_counts(lc);
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
        // ---- the compressed data of the block
        const InfCounts lcn = inf_counts(lc), dcn = inf_counts(dc);
        for (;;) {
            if (budget-- == 0) { err = INF_EOUTPUT; break; }
            const int sym = inf_decode(b, lc, lcn);
            if (sym < 0) { err = b.fail ? INF_ETRUNC : INF_ESYMBOL; break; }
            if (sym < 256) {
                if (pos >= olen) { err = INF_EOUTPUT; break; }
                o[pos++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256) break;
            const uint32_t ls = (uint32_t)sym - 257u;
            if (ls >= 29u) { err = INF_ESYMBOL; break; }
            const uint32_t len = (uint32_t)INF_LBASE[ls] + b.get(INF_LEXT[ls]);
            const int ds = inf_decode(b, dc, dcn);
            if (ds < 0 || ds >= INF_DCODES) { err = b.fail ? INF_ETRUNC : INF_ESYMBOL; break; }
            const uint32_t dist = (uint32_t)INF_DBASE[ds] + b.get(INF_DEXT[ds]);
            if (b.fail) { err = INF_ETRUNC; break; }
            if (dist > pos || pos + len > olen) { err = INF_EOUTPUT; break; }
            // The copy, eight bytes at a time: a byte-wise `o[pos] = o[pos - dist]` makes every byte wait for the store
            // before it (same-thread load after store: ~2.7 us per byte measured, the whole inflate).  ONE unaligned
            // 8-byte load of the source; distances below 8 repeat: the first `dist` bytes are replicated into a
            // pattern of the largest multiple of dist that fits 8 bytes.  Wide stores may run up to 7 bytes past the
            // match -- bytes of this member that later output overwrites -- so they stop 8 bytes before its end.
            const uint32_t stop = pos + len;
            if (dist >= 8u) {
                while (pos + 8u <= stop) {
                    uint64_t w;
                    __builtin_memcpy(&w, o + pos - dist, 8);
                    __builtin_memcpy(o + pos, &w, 8);
                    pos += 8u;
                }
                if (pos < stop && pos + 8u <= olen) {   // the tail in one more wide copy
                    uint64_t w;
                    __builtin_memcpy(&w, o + pos - dist, 8);
                    __builtin_memcpy(o + pos, &w, 8);
                    pos = stop;
                }
            } else if (stop + 8u <= olen) {
                uint64_t w = 0;
                for (uint32_t i = 0; i < dist; ++i) w |= (uint64_t)o[pos - dist + i] << (8u * i);
                for (uint32_t have = dist; have < 8u; have *= 2u) w |= w << (8u * have);   // byte j = pattern[j mod dist]
                const uint32_t step = (8u / dist) * dist;
                for (; pos < stop; pos += step) __builtin_memcpy(o + pos, &w, 8);
                pos = stop;
            }
            for (; pos < stop; ++pos) o[pos] = o[pos - dist];
        }
    }
    if (err == INF_OK && pos != olen) err = INF_ESIZE;
    if (err == INF_OK) {   // CRC-32 of the member (gzip trailer), byte by byte through the LDS copy of the table
        uint32_t c = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < olen; ++i) c = s_crc[(c ^ o[i]) & 0xFFu] ^ (c >> 8);
        if ((c ^ 0xFFFFFFFFu) != crc[mem]) err = INF_ECRC;
    }
    status[mem] = err;
}

}  // namespace tsx
