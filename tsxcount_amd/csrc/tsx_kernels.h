// tsx_kernels.h -- the HIP kernels of the counting path (gfx950, wave64).
//
//   line_count_kernel   FASTQ pass 1: non-empty line terminators per tile
//   line_*scan* kernels FASTQ pass 2: exclusive scan -> line index at tile start
//   count_fastq_kernel  FASTQ pass 3, atomic path: scan + 2-bit encode + hash + dedup + insert
//   strip_desc_kernel   FASTQ pass 3, partitioned path: the tile front end -> 16-byte strip descriptions
//   walk_log_kernel     ... walked with every lane busy: rolling hash, per-wave key log (tsx_partition.h takes it
//                       from there; walk_part_kernel there is the walk fused with radix level 1); *_wide_*: k > 32
//   add_kmers_kernel    addKmer for a batch of encoded k-mers
//   get_counts_kernel   getKmerCount(kmer) for a batch
//   occupied_kernel     getKmerCount() (occupied slots)
//   dump_kernel         getAllKmers + counts, optionally grouped by owner rank
//   synth_fill_kernel   synthetic FASTQ text (generateFakeSequences.py shape)
#pragma once
#include "tsx_device.h"

namespace tsx {

constexpr int NT = 256;          // threads per workgroup (4 waves)
constexpr int TILE = 4096;       // FASTQ bytes (k-mer start positions) per tile
constexpr int HALO = 128;        // bytes past the tile a window may reach (k <= 128)
constexpr int BATCH = 2048;      // start positions per dedup round (8 per thread)
constexpr int PER_THREAD = BATCH / NT;
constexpr int DSLOTS = 4096;     // LDS dedup slots per round
constexpr int DPROBES = 8;

__device__ __forceinline__ bool is_nl(uint32_t b) { return b == (uint32_t)'\n'; }
// SequenceUtils.h:98-125: A=0 C=1 G=2 T=3; same formula for every other byte
__device__ __forceinline__ uint32_t base_code(uint32_t b) { return ((b >> 1) ^ (b >> 2)) & 3u; }

// 16 bytes -> 16 newline flags, 16 line-end flags (newline that closes a
// non-empty line, FastXReader.h:365-370 drops empty lines), 32 code bits.
__device__ __forceinline__ void classify16(const uint4 v, bool prev_nl, uint32_t &nl16, uint32_t &le16,
                                           uint32_t &code32) {
    // Four bytes at a time (the byte-wise form cost 134 vector instructions per 16 bytes, a sixth of the scan).
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    nl16 = 0; code32 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // base_code of every byte in bits 0-1 of its byte, then the four 2-bit fields side by side
        const uint32_t t = ((w[i] >> 1) ^ (w[i] >> 2)) & 0x03030303u;
        const uint32_t x = t | (t >> 6);
        code32 |= ((x | (x >> 12)) & 0xFFu) << (8 * i);
        // bit 7 of a byte of nz is set iff the byte differs from '\n'
        const uint32_t z = w[i] ^ 0x0A0A0A0Au;
        const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
        uint32_t u = (~nz & 0x80808080u) >> 7;
        u |= u >> 7;
        nl16 |= ((u | (u >> 14)) & 0xFu) << (4 * i);
    }
    const uint32_t prev = (nl16 << 1) | (prev_nl ? 1u : 0u);  // bit i = byte i-1 is a newline
    le16 = nl16 & ~prev & 0xFFFFu;
}

// Loads 16 bytes at byte offset off (multiple of 16) of a text of n bytes;
// bytes at or past n read as '\n' (the last line may lack its terminator).
__device__ __forceinline__ uint4 load16(const uint8_t *buf, uint64_t off, uint64_t n) {
    if (off + 16 <= n) return *reinterpret_cast<const uint4 *>(buf + off);
    uint64_t lo = 0x0A0A0A0A0A0A0A0AULL, hi = lo;   // the text's last, partial 16 bytes: a small rolled loop
#pragma unroll 1
    for (uint32_t i = 0; i < 16u; ++i) {
        if (off + i >= n) break;
        const uint64_t b = buf[off + i], sh = 8u * (i & 7u);
        if (i < 8u) lo = (lo & ~(0xFFULL << sh)) | (b << sh);
        else hi = (hi & ~(0xFFULL << sh)) | (b << sh);
    }
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
// Is the byte before offset off a newline?  Offset 0 counts as a line start
// unless the caller says a previous piece ended mid-line (head_open = 1) or that the
// piece is a window into a longer device text whose previous byte can be read (head_open < 0).
__device__ __forceinline__ bool prev_is_nl(const uint8_t *buf, uint64_t off, uint64_t n, int head_open) {
    if (off == 0) return head_open < 0 ? (*(buf - 1) == (uint8_t)'\n') : !head_open;
    if (off - 1 >= n) return true;
    return buf[off - 1] == (uint8_t)'\n';
}

// Inclusive prefix sum over the wave in six DPP adds (rows of 16 by row_shr 1/2/4/8, then lane 15 of a row
// into the next row and lane 31 into the upper half); the shuffle form went through the LDS pipe six times.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ unsigned long long wave_incl_scan64(unsigned long long v) {   // small kernels only
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Pass 1: tile_cnt[t] = number of non-empty line terminators in tile t.
__global__ __launch_bounds__(NT) void line_count_kernel(const uint8_t *buf, uint64_t n, uint64_t own_end,
                                                        int head_open, uint32_t *tile_cnt, uint64_t ntiles) {
    __shared__ uint32_t s_w[NT / 64];
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t off = tile * TILE + (uint64_t)threadIdx.x * 16;
        uint32_t nl, le, code;
        classify16(load16(buf, off, n), prev_is_nl(buf, off, n, head_open), nl, le, code);
        // terminators at or past own_end belong to the next piece
        if (off + 16 > own_end) le &= (off >= own_end) ? 0u : ((1u << (own_end - off)) - 1u);
        uint32_t c = __popc(le);
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) tile_cnt[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// Pass 2: tile_cnt -> line index at each tile start (exclusive scan, in place), starting
// from *carry (lines seen in earlier pieces) and leaving the running total there.
// Three small launches: chunk sums, scan of the chunk sums, scan inside each chunk.
constexpr int SCAN_CHUNK = 1024;
__global__ __launch_bounds__(SCAN_CHUNK) void line_chunk_sum_kernel(const uint32_t *tile_cnt, uint64_t ntiles,
                                                                    uint32_t *chunk_sum) {
    __shared__ uint32_t s_w[SCAN_CHUNK / 64];
    const uint64_t i = (uint64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x;
    uint32_t v = (i < ntiles) ? tile_cnt[i] : 0;
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < SCAN_CHUNK / 64; ++w) t += s_w[w];
        chunk_sum[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void line_chunk_scan_kernel(uint32_t *chunk_sum, uint64_t nchunks, uint32_t *carry) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_base;
    if (threadIdx.x == 0) s_base = *carry;
    __syncthreads();
    for (uint64_t start = 0; start < nchunks; start += 1024) {
        const uint64_t i = start + threadIdx.x;
        const uint32_t v = (i < nchunks) ? chunk_sum[i] : 0;
        const uint32_t inc = wave_incl_scan(v);
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += s_w[w];
        const uint32_t base = s_base;
        if (i < nchunks) chunk_sum[i] = base + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_base = base + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *carry = s_base;
}
__global__ __launch_bounds__(SCAN_CHUNK) void line_scan_kernel(uint32_t *tile_cnt, uint64_t ntiles,
                                                               const uint32_t *chunk_base) {
    __shared__ uint32_t s_w[SCAN_CHUNK / 64];
    const uint64_t i = (uint64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x;
    const uint32_t v = (i < ntiles) ? tile_cnt[i] : 0;
    const uint32_t inc = wave_incl_scan(v);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = chunk_base[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += s_w[w];
    if (i < ntiles) tile_cnt[i] = woff + inc - v;
}

// Extract the k-mer that starts at byte position p of the tile from the packed
// 2-bit code array in LDS (UBigInt layout: base i in bits 2i,2i+1).
template <int WK>
__device__ __forceinline__ void extract_kmer(const uint64_t *s_codes, uint32_t p, uint64_t top_mask,
                                             uint64_t (&x)[WK]) {
    const uint32_t w = p >> 5, o = (2 * p) & 63;
#pragma unroll
    for (int t = 0; t < WK; ++t) {
        uint64_t v = s_codes[w + t] >> o;
        if (o) v |= s_codes[w + t + 1] << (64 - o);
        x[t] = v;
    }
    x[WK - 1] &= top_mask;
}
template <int WK>
__device__ __forceinline__ bool kmer_eq(const uint64_t (&a)[WK], const uint64_t (&b)[WK]) {
    bool e = true;
#pragma unroll
    for (int t = 0; t < WK; ++t) e &= (a[t] == b[t]);
    return e;
}

// Pass 3: the hot path.  Per tile of TILE bytes:
//   1. 16 B/lane coalesced loads; classify into newline / line-end bit masks
//      and a packed 2-bit code array in LDS (no byte ever re-read from HBM
//      except the k-1 byte halo);
//   2. workgroup prefix sum of line ends -> line index of every byte, so that
//      "second line of a 4-line record" (FastXReader.h:71-77) is a bit test;
//   3. per start position: window test against the newline mask
//      (createKMers, testExecution.h:15-36), funnel-shift extract of the 2k
//      bits (fromSequence), run-length merge of equal neighbours across the
//      wave (shuffle + ballot), LUT hash, claim-or-accumulate in an LDS dedup
//      table;
//   4. after a barrier the claimant of each dedup slot issues ONE global
//      insert carrying the slot's total.
template <int WK>
__global__ __launch_bounds__(NT, WK == 1 ? 3 : 2) void count_fastq_kernel(TableParams p, const uint8_t *buf, uint64_t n,
                                                         uint64_t own_end, int head_open,
                                                         const uint32_t *tile_line, uint64_t ntiles, int dbg) {
    __shared__ uint64_t s_codes[(TILE + HALO) / 32 + 2];
    __shared__ uint64_t s_nl[(TILE + HALO) / 64 + 3];
    __shared__ uint64_t s_le[TILE / 64];
    __shared__ uint8_t s_lb[TILE / 16];
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_dpos[DSLOTS];
    __shared__ uint32_t s_dcnt[DSLOTS];
    extern __shared__ uint64_t s_lut[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int lut_words = p.groups * (1 << p.g) * WK;
    for (int i = tid; i < lut_words; i += NT) s_lut[i] = p.lut[i];
    for (int i = tid; i < DSLOTS; i += NT) { s_dpos[i] = 0; s_dcnt[i] = 0; }
    if (tid < 3) s_nl[(TILE + HALO) / 64 + tid] = ~0ULL;
    if (tid < 2) s_codes[(TILE + HALO) / 32 + tid] = 0;
    unsigned long long added = 0;
    const uint32_t k = (uint32_t)p.k;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t base = tile * TILE;
        lds_barrier();  // previous tile's LDS fully consumed
        {
            const uint64_t off = base + (uint64_t)tid * 16;
            uint32_t nl, le, code;
            classify16(load16(buf, off, n), prev_is_nl(buf, off, n, head_open), nl, le, code);
            reinterpret_cast<uint32_t *>(s_codes)[tid] = code;
            reinterpret_cast<uint16_t *>(s_nl)[tid] = (uint16_t)nl;
            reinterpret_cast<uint16_t *>(s_le)[tid] = (uint16_t)le;
            if (tid < HALO / 16) {
                const uint64_t hoff = base + TILE + (uint64_t)tid * 16;
                uint32_t hnl, hle, hcode;
                classify16(load16(buf, hoff, n), false, hnl, hle, hcode);
                reinterpret_cast<uint32_t *>(s_codes)[TILE / 16 + tid] = hcode;
                reinterpret_cast<uint16_t *>(s_nl)[TILE / 16 + tid] = (uint16_t)hnl;
            }
            const uint32_t c = __popc(le);
            const uint32_t inc = wave_incl_scan(c);
            if (lane == 63) s_wsum[tid >> 6] = inc;
            lds_barrier();
            uint32_t woff = tile_line[tile];
            for (int w = 0; w < (tid >> 6); ++w) woff += s_wsum[w];
            s_lb[tid] = (uint8_t)((woff + inc - c) & 3u);
        }
        lds_barrier();

        for (int round = 0; round < TILE / BATCH; ++round) {
            uint64_t hk[PER_THREAD][WK];
            int slot_of[PER_THREAD];      // >=0: claimed LDS slot, -1: nothing to insert, -2: direct
            uint32_t direct_cnt[PER_THREAD];
#pragma unroll
            for (int j = 0; j < PER_THREAD; ++j) {
                const uint32_t pp = (uint32_t)(round * BATCH + j * NT + tid);
                const uint64_t gpos = base + pp;
                // line index of this byte: group base + line ends before it in the group
                const uint32_t grp = pp >> 4;
                const uint32_t le_before = reinterpret_cast<const uint16_t *>(s_le)[grp] & ((1u << (pp & 15)) - 1u);
                const uint32_t line = (uint32_t)s_lb[grp] + __popc(le_before);
                // no newline inside [pp, pp+k)
                const uint32_t w = pp >> 6, o = pp & 63;
                uint64_t m0 = s_nl[w] >> o, m1 = s_nl[w + 1] >> o;
                if (o) { m0 |= s_nl[w + 1] << (64 - o); m1 |= s_nl[w + 2] << (64 - o); }
                const uint64_t need0 = (k >= 64) ? ~0ULL : ((1ULL << k) - 1ULL);
                const uint64_t need1 = (k > 64) ? ((k >= 128) ? ~0ULL : ((1ULL << (k - 64)) - 1ULL)) : 0ULL;
                const bool valid = ((line & p.line_mask) == 1u) && ((m0 & need0) == 0) && ((m1 & need1) == 0) &&
                                   (gpos + k <= n) && (gpos < own_end);
                slot_of[j] = -1; direct_cnt[j] = 0;
                // header, '+' and quality lines make up half of a FASTQ text: a wave whose 64
                // consecutive positions hold no k-mer start skips extraction, hashing and dedup
                if (__ballot(valid) == 0ULL) continue;
                uint64_t x[WK];
                extract_kmer<WK>(s_codes, pp, p.top_mask, x);
                // run-length merge across the wave: lanes hold consecutive positions
                uint64_t xp[WK];
#pragma unroll
                for (int t = 0; t < WK; ++t) xp[t] = __shfl_up((unsigned long long)x[t], 1, 64);
                const bool prev_valid = __shfl_up((int)valid, 1, 64) != 0;
                const bool leader = valid && (lane == 0 || !prev_valid || !kmer_eq<WK>(x, xp));
                const unsigned long long bnd = __ballot(leader || !valid);
                const unsigned long long above = (lane == 63) ? 0ULL : (bnd >> (lane + 1));
                const uint32_t runlen = (above ? (uint32_t)__builtin_ctzll(above) : (uint32_t)(63 - lane)) + 1u;
                added += valid ? 1ULL : 0ULL;
                if (leader) {
                    hash_apply<WK>(p, (const uint64_t *)s_lut, x, hk[j]);
                    uint32_t slot = (uint32_t)(mix64(hk[j][0] ^ (WK > 1 ? hk[j][WK - 1] : 0)) >> 40) & (DSLOTS - 1);
                    slot_of[j] = -2; direct_cnt[j] = runlen;
                    for (int pr = 0; pr < DPROBES; ++pr) {
                        const uint32_t old = atomicCAS(&s_dpos[slot], 0u, pp + 1u);
                        if (old == 0u) {
                            atomicAdd(&s_dcnt[slot], runlen);
                            slot_of[j] = (int)slot;
                            break;
                        }
                        uint64_t y[WK];
                        extract_kmer<WK>(s_codes, old - 1u, p.top_mask, y);
                        if (kmer_eq<WK>(x, y)) {
                            atomicAdd(&s_dcnt[slot], runlen);
                            slot_of[j] = -1;
                            break;
                        }
                        slot = (slot + 1) & (DSLOTS - 1);
                    }
                }
            }
            lds_barrier();
            // Phase B.  Totals are final now: the claimant of each dedup slot issues ONE global
            // insert carrying the slot's total (one CAS for a new key, CAS + add otherwise).
#pragma unroll
            for (int j = 0; j < PER_THREAD; ++j) {
                if (slot_of[j] == -1) continue;
                uint64_t d = direct_cnt[j];
                if (slot_of[j] >= 0) {
                    d = s_dcnt[slot_of[j]];
                    s_dcnt[slot_of[j]] = 0;
                    s_dpos[slot_of[j]] = 0;
                }
                if (!(dbg & 1)) insert_key<WK>(p, hk[j], d);
                else if (d == 0xFFFFFFFFFFULL) p.stats[ST_SCRATCH] = hk[j][0];  // keep the hash alive in ablation runs
            }
            lds_barrier();
        }
    }
    for (int d = 32; d > 0; d >>= 1) added += __shfl_down(added, d, 64);
    if (lane == 0 && added) atomicAdd(&p.stats[ST_KMERS], added);
}

// ---- the scan in two kernels: describe the strips, then walk them ------------------------------------------
// Lines of a read file are about as long as a wave's share of a tile (1 KiB), so nearly every wave of a kernel that
// scans AND walks (the one-kernel forms of rounds 1 and 2) holds some sequence bytes and walks its 16 positions with half
// of its lanes in '+'/quality lines: the roll and append instructions -- two thirds of such a kernel -- run at 50 % lane use.
// strip_desc_kernel does the tile front end only (classify, line index, validity mask) and writes a 16-byte
// DESCRIPTION of every strip that holds a k-mer start -- the 48 bases from its first position as 2-bit codes and
// the 16 validity bits -- to its wave's region of a descriptor array; walk_part_kernel (tsx_partition.h) reads
// descriptions, one per lane, every lane busy, and does first window + rolls + level-1 rings.  k <= 32.
// HOMOUT (the minimizer exchange, 20 <= k <= 32, short descriptions): homopolymer k-mers leave the validity bits here and
// are counted per base in hom_out[0..3] -- a strip that lies wholly in a poly-A tail is then not described at all.
template <bool HOMOUT = false>
__global__ __launch_bounds__(NT, 5) void strip_desc_kernel(TableParams p, const uint8_t *buf, uint64_t n, uint64_t own_end,
                                                           int head_open, const uint32_t *tile_line, uint64_t ntiles,
                                                           uint4 *desc, uint64_t desc_cap, unsigned long long *desc_cnt,
                                                           unsigned long long *kmer_sum, int long_desc,
                                                           unsigned long long *hom_out = nullptr) {
    // long_desc (the exchange of a sharded run): FOUR neighbouring strips in one description of 32 bytes -- 96 bases
    // + 64 validity bits, half the bytes per start position (the 32 bases behind a strip's own 16 are shared).
    __shared__ uint64_t s_codes[(TILE + HALO) / 32 + 2];
    __shared__ uint64_t s_nl[(TILE + HALO) / 64 + 3];
    __shared__ uint64_t s_le[TILE / 64];
    __shared__ uint8_t s_lb[TILE / 16];
    __shared__ uint32_t s_wsum[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) s_nl[(TILE + HALO) / 64 + tid] = ~0ULL;
    if (tid < 2) s_codes[(TILE + HALO) / 32 + tid] = 0;
    unsigned long long added = 0;
    unsigned long long hacc = 0;   // HOMOUT: homopolymer occurrences, 16 bits per base (flushed before a field can fill up)
    const uint32_t k = (uint32_t)p.k;
    const uint32_t region = blockIdx.x * (NT / 64) + wave;
    uint4 *my = desc + (uint64_t)region * desc_cap * (long_desc ? 2u : 1u);
    uint32_t fill = 0;  // wave-uniform, in descriptions
    const uint64_t start_lim = min(own_end, (n >= k) ? n - k + 1 : 0ULL);
    const uint64_t lt = (1ULL << lane) - 1ULL;

    uint4 cur = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au), hcur = cur;
    bool cur_pnl = true;
    uint32_t cur_line = 0;
    if ((uint64_t)blockIdx.x < ntiles) {
        const uint64_t off = (uint64_t)blockIdx.x * TILE + (uint64_t)tid * 16;
        cur_line = tile_line[blockIdx.x];
        cur = load16(buf, off, n);
        cur_pnl = prev_is_nl(buf, off, n, head_open);
        if (tid < HALO / 16) hcur = load16(buf, (uint64_t)blockIdx.x * TILE + TILE + (uint64_t)tid * 16, n);
    }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t base = tile * TILE;
        lds_barrier();  // previous tile's LDS fully consumed
        {
            uint32_t nl, le, code;
            classify16(cur, cur_pnl, nl, le, code);
            reinterpret_cast<uint32_t *>(s_codes)[tid] = code;
            reinterpret_cast<uint16_t *>(s_nl)[tid] = (uint16_t)nl;
            reinterpret_cast<uint16_t *>(s_le)[tid] = (uint16_t)le;
            if (tid < HALO / 16) {
                uint32_t hnl, hle, hcode;
                classify16(hcur, false, hnl, hle, hcode);
                reinterpret_cast<uint32_t *>(s_codes)[TILE / 16 + tid] = hcode;
                reinterpret_cast<uint16_t *>(s_nl)[TILE / 16 + tid] = (uint16_t)hnl;
            }
            const uint32_t c = __popc(le);
            const uint32_t inc = wave_incl_scan(c);
            if (lane == 63) s_wsum[wave] = inc;
            lds_barrier();
            uint32_t woff = cur_line;
            for (int w = 0; w < wave; ++w) woff += s_wsum[w];
            s_lb[tid] = (uint8_t)((woff + inc - c) & 3u);
        }
        {
            const uint64_t nt = tile + gridDim.x;
            if (nt < ntiles) {
                const uint64_t off = nt * TILE + (uint64_t)tid * 16;
                cur_line = tile_line[nt];
                cur = load16(buf, off, n);
                cur_pnl = prev_is_nl(buf, off, n, head_open);
                if (tid < HALO / 16) hcur = load16(buf, nt * TILE + TILE + (uint64_t)tid * 16, n);
            }
        }
        lds_barrier();
        // ---- this lane's strip: start positions s .. s+15 of the tile -----------
        const uint32_t s0 = (uint32_t)tid * 16;
        const uint32_t *codes32 = reinterpret_cast<const uint32_t *>(s_codes);
        const uint32_t *nl32 = reinterpret_cast<const uint32_t *>(s_nl);
        uint64_t m;
        {
            const uint32_t w = (uint32_t)tid >> 1, sh = ((uint32_t)tid & 1u) * 16u;
            const uint32_t a0 = nl32[w], a1 = nl32[w + 1], a2 = nl32[w + 2];
            m = (uint64_t)__funnelshift_r(a0, a1, sh) | ((uint64_t)__funnelshift_r(a1, a2, sh) << 32);
        }
        uint64_t r = m;
        uint32_t span = 1;
        while (span * 2 <= k) { r |= r >> span; span *= 2; }
        if (span < k) r |= r >> (k - span);
        const uint32_t e16 = reinterpret_cast<const uint16_t *>(s_le)[tid];
        const uint32_t lb = s_lb[tid];
        uint32_t c0 = e16 << 1; c0 ^= c0 << 1; c0 ^= c0 << 2; c0 ^= c0 << 4; c0 ^= c0 << 8;
        uint32_t c1 = (e16 & c0) << 1; c1 ^= c1 << 1; c1 ^= c1 << 2; c1 ^= c1 << 4; c1 ^= c1 << 8;
        const uint32_t l0 = (lb & 1u) ? 0xFFFFu : 0u, l1 = (lb & 2u) ? 0xFFFFu : 0u;
        const uint32_t b0 = c0 ^ l0, b1 = c1 ^ l1 ^ (c0 & l0);
        const uint64_t g0 = base + s0;
        const uint32_t jmax = (start_lim > g0) ? (uint32_t)min((uint64_t)16, start_lim - g0) : 0u;
        const uint32_t nb1 = (p.line_mask & 2u) ? ~b1 : ~0u;
        uint32_t vm = ~(uint32_t)r & b0 & nb1 & ((1u << jmax) - 1u);
        added += (unsigned long long)__popc(vm);
        if constexpr (HOMOUT) {
            if (vm) {   // ne: bit 2j = base j differs from base j + 1; window i is a homopolymer iff none in i .. i + k - 2
                const uint32_t c0 = codes32[tid], c1 = codes32[tid + 1], c2 = codes32[tid + 2];
                const uint64_t lo = (uint64_t)c0 | ((uint64_t)c1 << 32), hi = c2;
                const uint64_t xl = lo ^ ((lo >> 2) | (hi << 62)), xh = hi ^ (hi >> 2);
                const uint64_t nl = (xl | (xl >> 1)) & 0x5555555555555555ULL, nh = (xh | (xh >> 1)) & 0x5555555555555555ULL;
                const bool core_ok = ((nl >> 30) & ((1ULL << (2u * (k - 16u))) - 1ULL)) == 0ULL;
              if (core_ok) {   // (bases 15 .. k - 1, which every window of the strip holds, are one base: rare outside poly-A)
                const uint32_t z = (uint32_t)nl & 0x3FFFFFFFu;
                const uint32_t a = z ? ((31u - (uint32_t)__clz((int)z)) >> 1) + 1u : 0u;
                const uint32_t s2 = 2u * (k - 1u);
                const uint32_t zz = (uint32_t)((nl >> s2) | (nh << (64u - s2))) & 0x3FFFFFFFu;
                const uint32_t b = zz ? (uint32_t)(__ffs((int)zz) - 1) >> 1 : 15u;
                const uint32_t homm = (a <= b) ? (((2u << b) - 1u) & ~((1u << a) - 1u)) & 0xFFFFu : 0u;
                hacc += (unsigned long long)__popc(vm & homm) << (16u * ((c0 >> 30) & 3u));
                vm &= ~homm;
              }
            }
        }
        if (!long_desc) {
            const unsigned long long hb = __ballot(vm != 0u);
            if (hb) {
                // (bit 16: the description before this one is the strip before this one in the text -- desc_owner_split_kernel)
                const uint32_t nbr = (uint32_t)(((hb << 1) >> lane) & 1ULL) << 16;
                if (vm) my[fill + (uint32_t)__builtin_popcountll(hb & lt)] = make_uint4(codes32[tid], codes32[tid + 1], codes32[tid + 2], vm | nbr);
                fill += (uint32_t)__builtin_popcountll(hb);
            }
        } else {   // lanes 4q .. 4q+3 hold 64 consecutive start positions: lane 4q writes for all four
            const uint32_t v1 = __shfl_down(vm, 1, 64), v2 = __shfl_down(vm, 2, 64), v3 = __shfl_down(vm, 3, 64);
            const uint32_t vlo = vm | (v1 << 16), vhi = v2 | (v3 << 16);
            const bool has = (lane & 3) == 0 && (vlo | vhi) != 0u;
            const unsigned long long hb = __ballot(has);
            if (hb) {
                if (has) {
                    uint4 *o = my + (uint64_t)(fill + (uint32_t)__builtin_popcountll(hb & lt)) * 2u;
                    o[0] = make_uint4(codes32[tid], codes32[tid + 1], codes32[tid + 2], codes32[tid + 3]);
                    o[1] = make_uint4(codes32[tid + 4], codes32[tid + 5], vlo, vhi);
                }
                fill += (uint32_t)__builtin_popcountll(hb);
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) added += __shfl_down(added, d, 64);
    if (lane == 0) {
        if (added) atomicAdd(&p.stats[ST_KMERS], added);
        if (added && kmer_sum) atomicAdd(kmer_sum, added);   // sharded runs: checked against what the walks keep
        desc_cnt[region] = fill;
    }
    if constexpr (HOMOUT) {   // (a lane sees one strip per tile of its workgroup: < 4096 strips of 16 for texts below 4 GiB)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            unsigned long long t = (hacc >> (16 * b)) & 0xFFFFULL;
            for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d, 64);
            if (lane == 0 && t && hom_out) atomicAdd(&hom_out[b], t);
        }
    }
}

// walk_log_kernel: the walk into a key log, fed from strip descriptions (strip_desc_kernel above) -- one description per lane, every lane busy, no barrier after the set-up (a wave reads descriptor
// regions w, w + G, ... and appends to its own log region; the histogram is the wave's).  Used where the keys must
// come out as a packed log: sharded scans (histogram by owner) and tables that need one radix level.
__global__ __launch_bounds__(NT, 4) void walk_log_kernel(TableParams p, const uint4 *desc, uint64_t desc_cap,
                                                         const unsigned long long *desc_cnt, uint32_t nregions, int dbg,
                                                         uint64_t *log, uint64_t log_cap, unsigned long long *log_cnt,
                                                         uint32_t *hist, uint32_t hist_nb, uint32_t hist_shift,
                                                         uint64_t n_packed, int long_desc, int own_only,
                                                         unsigned long long *emit_sum) {
    // Sharded runs at larger world sizes (tsx_hip_shard_filter_device): the descriptions are ONE packed array of n_packed
    // entries (cut into runs of desc_cap, one per wave), long_desc: four strips per 32-byte entry, own_only: only the keys
    // this GPU owns are logged (emit_sum += their number).  This GPU keeps one key in N, so what is left of the kernel
    // is first window + rolls + owner test: without LDS rings it runs at 20 waves per CU.
    constexpr int HOT_N = 8;
    __shared__ uint64_t s_hot_key[(NT / 64) * HOT_N];
    __shared__ uint32_t s_hot_cnt[(NT / 64) * HOT_N];
    __shared__ uint32_t s_hist[(NT / 64) * 512];  // fan-out <= 512
    __shared__ uint64_t s_roll[64];
    __shared__ uint64_t s_homh[4];
    extern __shared__ uint64_t s_lut[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lut_words = p.groups * (1 << p.g);
    for (int i = tid; i < lut_words; i += NT) s_lut[i] = p.lut[i];
    for (int i = tid; i < (NT / 64) * 512; i += NT) s_hist[i] = 0;
    if (tid < 64) s_roll[tid] = p.roll[tid];
    if (tid < 4) {
        const uint64_t x[1] = {(0x5555555555555555ULL * (uint64_t)tid) & p.top_mask};
        uint64_t hh[1];
        hash_apply<1>(p, p.lut, x, hh);
        s_homh[tid] = hh[0];
    }
    if (tid < (NT / 64) * HOT_N) { s_hot_key[tid] = 0; s_hot_cnt[tid] = 0; }
    const uint32_t k = (uint32_t)p.k;
    const uint32_t G = gridDim.x * (NT / 64), region = blockIdx.x * (NT / 64) + wave;
    uint64_t *my_log = log + (uint64_t)region * log_cap;
    uint32_t *my_hist = s_hist + wave * 512;
    uint32_t fill = 0;  // wave-uniform
    const uint32_t cap32 = (uint32_t)min(log_cap, (uint64_t)0xFFFFFFFFu);
    const TableParams *pk = (const TableParams *)__builtin_amdgcn_kernarg_segment_ptr();
    auto side_insert = [&](uint64_t hkey, uint64_t d) {
        if (dbg & 1) return;
        defer_append1(pk, hkey, d);
    };
    auto is_mine = [&](uint64_t hk) -> bool {
        if (!own_only) return true;
        const uint64_t h1[1] = {hk};
        return owner_shard<1>(p, h1) == p.shard;
    };
    unsigned long long emitted = 0;
    lds_barrier();
    for (uint32_t r = region; r < nregions; r += G) {
        const uint32_t nr = n_packed ? (uint32_t)(((uint64_t)r * desc_cap < n_packed) ? min(desc_cap, n_packed - (uint64_t)r * desc_cap) : 0ULL)
                                     : (uint32_t)min((uint64_t)desc_cnt[r], desc_cap);
        const uint32_t per = long_desc ? 16u : 64u, me = long_desc ? (uint32_t)lane >> 2 : (uint32_t)lane;
        const uint4 *rd = desc + (uint64_t)r * desc_cap * (long_desc ? 2u : 1u);
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4 dn = z4, dn2 = z4;
        if (me < nr) { if (long_desc) { dn = rd[2u * me]; dn2 = rd[2u * me + 1u]; } else dn = rd[me]; }
        for (uint32_t base = 0; base < nr; base += per) {
            const uint4 d = dn, d2 = dn2;
            if (base + per < nr) {
                dn = z4; dn2 = z4;
                if (base + per + me < nr) {
                    if (long_desc) { dn = rd[2u * (base + per + me)]; dn2 = rd[2u * (base + per + me) + 1u]; }
                    else dn = rd[base + per + me];
                }
            }
            uint32_t cw0 = d.x, cw1 = d.y, cw2 = d.z, vm = d.w & 0xFFFFu;   // (bit 16: strip_desc_kernel's neighbour mark)
            if (long_desc) {   // strip t of the four: bases 16 t .. 16 t + 47, validity bits 16 t .. 16 t + 15
                const uint32_t t = (uint32_t)lane & 3u;
                const uint32_t w0 = d.x, w1 = d.y, w2 = d.z, w3 = d.w, w4 = d2.x, w5 = d2.y;
                cw0 = (t == 0u) ? w0 : (t == 1u) ? w1 : (t == 2u) ? w2 : w3;
                cw1 = (t == 0u) ? w1 : (t == 1u) ? w2 : (t == 2u) ? w3 : w4;
                cw2 = (t == 0u) ? w2 : (t == 1u) ? w3 : (t == 2u) ? w4 : w5;
                const uint32_t vw = (t < 2u) ? d2.z : d2.w;
                vm = (vw >> ((t & 1u) * 16u)) & 0xFFFFu;
            }
            uint64_t h = 0;
            if (vm) {
                const uint64_t x[1] = {((uint64_t)cw0 | ((uint64_t)cw1 << 32)) & p.top_mask};
                uint64_t hh[1];
                hash_apply<1>(p, (const uint64_t *)s_lut, x, hh);
                h = hh[0];
            }
            uint32_t inc;
            {
                const uint32_t o = 2u * k, ws = o >> 5, sh = o & 31u;
                const uint32_t w0 = (ws == 0u) ? cw0 : (ws == 1u) ? cw1 : cw2;
                const uint32_t w1 = (ws == 0u) ? cw1 : (ws == 1u) ? cw2 : 0u;
                inc = __funnelshift_r(w0, w1, sh);
            }
            uint32_t homm;   // bit j: the k-mer at strip position j is a homopolymer
            {
                const uint64_t lo = (uint64_t)cw0 | ((uint64_t)cw1 << 32), hi = cw2;
                const uint64_t dlo = lo ^ ((lo >> 2) | (hi << 62)), dhi = hi ^ (hi >> 2);
                uint64_t rlo = (dlo | (dlo >> 1)) & 0x5555555555555555ULL, rhi = (dhi | (dhi >> 1)) & 0x5555555555555555ULL;
                uint32_t span = 1;   // bases covered by the smear so far; k - 1 adjacent pairs must agree
                while (span * 2 <= k - 1) {
                    const uint32_t sh = 2u * span;
                    rlo |= (rlo >> sh) | (rhi << (64u - sh));
                    rhi |= rhi >> sh;
                    span *= 2;
                }
                if (span < k - 1) {
                    const uint32_t sh = 2u * (k - 1 - span);
                    rlo |= (rlo >> sh) | (rhi << (64u - sh));
                }
                uint32_t x = ~(uint32_t)rlo & 0x55555555u;   // even bits -> 16 contiguous bits
                x = (x | (x >> 1)) & 0x33333333u;
                x = (x | (x >> 2)) & 0x0F0F0F0Fu;
                x = (x | (x >> 4)) & 0x00FF00FFu;
                homm = (x | (x >> 8)) & 0xFFFFu;
            }
            const uint32_t hv = vm & homm;
            const uint32_t single = vm & ~homm;
            if (__ballot(hv != 0u)) {
                for (uint32_t b = 0; b < 4; ++b) {   // which of the four: the base at the position
                    uint32_t e = cw0 ^ (0x55555555u * b);
                    uint32_t y = ~(e | (e >> 1)) & 0x55555555u;
                    y = (y | (y >> 1)) & 0x33333333u;
                    y = (y | (y >> 2)) & 0x0F0F0F0Fu;
                    y = (y | (y >> 4)) & 0x00FF00FFu;
                    y = (y | (y >> 8)) & 0xFFFFu;
                    uint32_t tot = (uint32_t)__popc(hv & y);
                    if (__ballot(tot != 0u) == 0ULL) continue;
                    for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d, 64);
                    if (lane == 0 && is_mine(s_homh[b])) {
                            emitted += tot;
                        const uint64_t key = s_homh[b];
                        uint64_t *hkey = s_hot_key + wave * HOT_N;
                        uint32_t *hcnt = s_hot_cnt + wave * HOT_N;
                        int at = -1;
                        for (int q = 0; q < HOT_N; ++q)
                            if (hcnt[q] && hkey[q] == key) { at = q; break; }
                        if (at < 0)
                            for (int q = 0; q < HOT_N; ++q)
                                if (!hcnt[q]) { at = q; hkey[q] = key; break; }
                        if (at >= 0 && (uint64_t)hcnt[at] + tot < 0xFFFFFFF0ULL) hcnt[at] += tot;
                        else side_insert(key, tot);
                    }
                }
            }
            // The strip is walked in two halves of 8 positions (the hash rolls on across them): pass A
            // rolls and keeps the 8 hashes, pass B appends them to this wave's log region, one contiguous
            // piece per position.
            for (uint32_t j0 = 0; j0 < 16; j0 += 8) {
                const uint32_t s8 = (single >> j0) & 0xFFu;
                if (__ballot(s8 != 0u) == 0ULL) {   // nothing to log in this half: only roll on
                    if (j0 == 0) {
    #pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const uint32_t idx = ((uint32_t)h & 3u) | (__builtin_amdgcn_ubfe(cw0, 2u * j, 2u) << 2) |
                                                 (__builtin_amdgcn_ubfe(inc, 2u * j, 2u) << 4);
                            h = (h >> 2) ^ s_roll[idx];
                        }
                    }
                    continue;
                }
                uint64_t hs[8];
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    hs[j] = h;
                    if (j0 + j < 15) {
                        const uint32_t idx = ((uint32_t)h & 3u) | (__builtin_amdgcn_ubfe(cw0, 2u * (j0 + j), 2u) << 2) |
                                             (__builtin_amdgcn_ubfe(inc, 2u * (j0 + j), 2u) << 4);
                        h = (h >> 2) ^ s_roll[idx];
                    }
                }
                if (fill + 64u * 8u <= cap32) {   // the usual case: the region has room for whatever this half logs
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool em = ((s8 >> j) & 1u) && is_mine(hs[j]);
                        const unsigned long long mk = __ballot(em);
                        if (mk) {
                            if (em) {
                                ++emitted;
                                const uint64_t key = hs[j];
                                my_log[fill + (uint32_t)__builtin_popcountll(mk & ((1ULL << lane) - 1ULL))] = key;
                                atomicAdd(&my_hist[(uint32_t)(key >> hist_shift) & (hist_nb - 1)], 1u);
                            }
                            fill += (uint32_t)__builtin_popcountll(mk);
                        }
                    }
                } else {
                    for (int j = 0; j < 8; ++j) {
                        uint64_t kj = hs[0];
#pragma unroll
                        for (int t = 1; t < 8; ++t) kj = (j == t) ? hs[t] : kj;
                        const bool em = ((s8 >> j) & 1u) && is_mine(kj);
                        const unsigned long long mk = __ballot(em);
                        if (mk) {
                            if (em) {
                                ++emitted;
                                uint64_t key = hs[0];
    #pragma unroll
                                for (int t = 1; t < 8; ++t) key = (j == t) ? hs[t] : key;
                                const uint32_t at = fill + (uint32_t)__builtin_popcountll(mk & ((1ULL << lane) - 1ULL));
                                if (at < cap32) {
                                    my_log[at] = key;
                                    atomicAdd(&my_hist[(uint32_t)(key >> hist_shift) & (hist_nb - 1)], 1u);
                                } else {
                                    side_insert(key, 1);  // region full: deferred list (or the exchanged hot list)
                                }
                            }
                            fill += (uint32_t)__builtin_popcountll(mk);
                        }
                    }
                }
            }
        }
    }
    lds_barrier();
    if (tid < (NT / 64) * HOT_N && s_hot_cnt[tid]) side_insert(s_hot_key[tid], s_hot_cnt[tid]);
    if (emit_sum) {
        for (int d = 32; d > 0; d >>= 1) emitted += __shfl_down(emitted, d, 64);
        if (lane == 0 && emitted) atomicAdd(emit_sum, emitted);
    }
    if (lane == 0) log_cnt[region] = min(fill, cap32);   // (k-mers are counted by strip_desc_kernel)
    for (uint32_t b = lane; b < hist_nb; b += 64) hist[(size_t)b * G + region] = my_hist[b];
}

// Records of the partitioned path are RW 64-bit words: the WK limbs of the hashed key, padded to a power of
// two so that 128-byte bursts hold whole records (k = 65..96: three limbs travel as four words).
template <int WK> struct RecWords { static constexpr int value = (WK == 3) ? 4 : WK; };

// r |= r >> s over three 64-bit words (1 <= s <= 64): the smear step of "a newline anywhere in [p, p+k)".
__device__ __forceinline__ void shr_or3(uint64_t (&r)[3], uint32_t s) {
    if (s >= 64) { r[0] |= r[1]; r[1] |= r[2]; return; }
    r[0] |= (r[0] >> s) | (r[1] << (64u - s));
    r[1] |= (r[1] >> s) | (r[2] << (64u - s));
    r[2] |= r[2] >> s;
}

// ---- the two-kernel scan for multi-limb keys (k > 32) -------------------------------------------------------
// As strip_desc_kernel / walk_log_kernel.  What the walk of a strip needs of the text is its FIRST k-mer (WK limbs),
// the 16 bases that enter behind it and the validity bits (the bases that leave are the first 16 of the k-mer): a
// description is 2 WK + 2 words, padded to whole 16-byte units.
template <int WK> struct WideDescU4 { static constexpr int value = (2 * WK + 2 + 3) / 4; };   // uint4 per description

template <int WK>
__global__ __launch_bounds__(NT, 4) void strip_desc_wide_kernel(TableParams p, const uint8_t *buf, uint64_t n, uint64_t own_end,
                                                                int head_open, const uint32_t *tile_line, uint64_t ntiles,
                                                                uint4 *desc, uint64_t desc_cap, unsigned long long *desc_cnt) {
    constexpr int DU = WideDescU4<WK>::value;
    __shared__ uint64_t s_codes[(TILE + HALO) / 32 + 2];
    __shared__ uint64_t s_nl[(TILE + HALO) / 64 + 3];
    __shared__ uint64_t s_le[TILE / 64];
    __shared__ uint8_t s_lb[TILE / 16];
    __shared__ uint32_t s_wsum[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) s_nl[(TILE + HALO) / 64 + tid] = ~0ULL;
    if (tid < 2) s_codes[(TILE + HALO) / 32 + tid] = 0;
    unsigned long long added = 0;
    const uint32_t k = (uint32_t)p.k;
    const uint32_t region = blockIdx.x * (NT / 64) + wave;
    uint4 *my = desc + (uint64_t)region * desc_cap * DU;
    uint32_t fill = 0;  // wave-uniform, in descriptions
    const uint64_t start_lim = min(own_end, (n >= k) ? n - k + 1 : 0ULL);
    const uint64_t lt = (1ULL << lane) - 1ULL;

    // text loaded one tile ahead, as in strip_desc_kernel
    uint4 cur = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au), hcur = cur;
    bool cur_pnl = true;
    uint32_t cur_line = 0;
    if ((uint64_t)blockIdx.x < ntiles) {
        const uint64_t off = (uint64_t)blockIdx.x * TILE + (uint64_t)tid * 16;
        cur_line = tile_line[blockIdx.x];
        cur = load16(buf, off, n);
        cur_pnl = prev_is_nl(buf, off, n, head_open);
        if (tid < HALO / 16) hcur = load16(buf, (uint64_t)blockIdx.x * TILE + TILE + (uint64_t)tid * 16, n);
    }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t base = tile * TILE;
        lds_barrier();  // previous tile's LDS fully consumed
        {
            uint32_t nl, le, code;
            classify16(cur, cur_pnl, nl, le, code);
            reinterpret_cast<uint32_t *>(s_codes)[tid] = code;
            reinterpret_cast<uint16_t *>(s_nl)[tid] = (uint16_t)nl;
            reinterpret_cast<uint16_t *>(s_le)[tid] = (uint16_t)le;
            if (tid < HALO / 16) {
                uint32_t hnl, hle, hcode;
                classify16(hcur, false, hnl, hle, hcode);
                reinterpret_cast<uint32_t *>(s_codes)[TILE / 16 + tid] = hcode;
                reinterpret_cast<uint16_t *>(s_nl)[TILE / 16 + tid] = (uint16_t)hnl;
            }
            const uint32_t c = __popc(le);
            const uint32_t inc = wave_incl_scan(c);
            if (lane == 63) s_wsum[wave] = inc;
            lds_barrier();
            uint32_t woff = cur_line;   // lines before the tile, loaded a tile ahead like its text
            for (int w = 0; w < wave; ++w) woff += s_wsum[w];
            s_lb[tid] = (uint8_t)((woff + inc - c) & 3u);
        }
        {
            const uint64_t nt = tile + gridDim.x;
            if (nt < ntiles) {
                const uint64_t off = nt * TILE + (uint64_t)tid * 16;
                cur_line = tile_line[nt];
                cur = load16(buf, off, n);
                cur_pnl = prev_is_nl(buf, off, n, head_open);
                if (tid < HALO / 16) hcur = load16(buf, nt * TILE + TILE + (uint64_t)tid * 16, n);
            }
        }
        lds_barrier();

        const uint32_t s0 = (uint32_t)tid * 16;
        const uint32_t *codes32 = reinterpret_cast<const uint32_t *>(s_codes);
        const uint32_t *nl32 = reinterpret_cast<const uint32_t *>(s_nl);
        // newline flags of bytes [s0, s0 + 160): a window of position j <= 15 reaches byte s0 + 15 + k - 1 <= s0 + 141
        uint64_t r[3];
        {
            const uint32_t w = (uint32_t)tid >> 1, sh = ((uint32_t)tid & 1u) * 16u;
            const uint32_t a0 = nl32[w], a1 = nl32[w + 1], a2 = nl32[w + 2], a3 = nl32[w + 3], a4 = nl32[w + 4],
                           a5 = nl32[w + 5];
            r[0] = (uint64_t)__funnelshift_r(a0, a1, sh) | ((uint64_t)__funnelshift_r(a1, a2, sh) << 32);
            r[1] = (uint64_t)__funnelshift_r(a2, a3, sh) | ((uint64_t)__funnelshift_r(a3, a4, sh) << 32);
            r[2] = (uint64_t)__funnelshift_r(a4, a5, sh);
        }
        {   // bad_j = a newline in [s0+j, s0+j+k): OR of the mask shifted by 0..k-1, by doubling
            uint32_t span = 1;
            while (span * 2 <= k) { shr_or3(r, span); span *= 2; }
            if (span < k) shr_or3(r, k - span);
        }
        const uint32_t e16 = reinterpret_cast<const uint16_t *>(s_le)[tid];
        const uint32_t lb = s_lb[tid];
        uint32_t c0 = e16 << 1; c0 ^= c0 << 1; c0 ^= c0 << 2; c0 ^= c0 << 4; c0 ^= c0 << 8;
        uint32_t c1 = (e16 & c0) << 1; c1 ^= c1 << 1; c1 ^= c1 << 2; c1 ^= c1 << 4; c1 ^= c1 << 8;
        const uint32_t l0 = (lb & 1u) ? 0xFFFFu : 0u, l1 = (lb & 2u) ? 0xFFFFu : 0u;
        const uint32_t b0 = c0 ^ l0, b1 = c1 ^ l1 ^ (c0 & l0);       // bits 0 and 1 of the line index
        const uint64_t g0 = base + s0;
        const uint32_t jmax = (start_lim > g0) ? (uint32_t)min((uint64_t)16, start_lim - g0) : 0u;
        const uint32_t nb1 = (p.line_mask & 2u) ? ~b1 : ~0u;   // FASTA: every second line is a sequence
        const uint32_t vm = ~(uint32_t)r[0] & b0 & nb1 & ((1u << jmax) - 1u);   // sequence line, no newline, in range
        added += (unsigned long long)__popc(vm);
        const unsigned long long hb = __ballot(vm != 0u);
        if (hb) {
            if (vm) {
                uint64_t x[WK];
                extract_kmer<WK>(s_codes, s0, p.top_mask, x);
                uint32_t inc;
                {
                    const uint32_t o = 2u * k, ws = o >> 5, sh = o & 31u;
                    inc = __funnelshift_r(codes32[tid + ws], codes32[tid + ws + 1], sh);
                }
                uint32_t w[DU * 4];
#pragma unroll
                for (int t = 0; t < DU * 4; ++t) w[t] = 0;
#pragma unroll
                for (int t = 0; t < WK; ++t) { w[2 * t] = (uint32_t)x[t]; w[2 * t + 1] = (uint32_t)(x[t] >> 32); }
                w[2 * WK] = inc;
                w[2 * WK + 1] = vm;
                uint4 *o = my + (uint64_t)(fill + (uint32_t)__builtin_popcountll(hb & lt)) * DU;
#pragma unroll
                for (int u = 0; u < DU; ++u) o[u] = make_uint4(w[4 * u], w[4 * u + 1], w[4 * u + 2], w[4 * u + 3]);
            }
            fill += (uint32_t)__builtin_popcountll(hb);
        }
    }
    for (int d = 32; d > 0; d >>= 1) added += __shfl_down(added, d, 64);
    if (lane == 0) {
        if (added) atomicAdd(&p.stats[ST_KMERS], added);
        desc_cnt[region] = fill;
    }
}

template <int WK>
__global__ __launch_bounds__(NT, 2) void walk_log_wide_kernel(TableParams p, const uint4 *desc, uint64_t desc_cap,
                                                              const unsigned long long *desc_cnt, uint32_t nregions, int dbg,
                                                              uint64_t *log, uint64_t log_cap, unsigned long long *log_cnt,
                                                              uint32_t *hist, uint32_t hist_nb, uint32_t hist_shift) {
    constexpr int RW = RecWords<WK>::value;
    constexpr int DU = WideDescU4<WK>::value;
    constexpr int HOT_N = 8;
    __shared__ uint64_t s_hot_key[(NT / 64) * HOT_N * WK];
    __shared__ uint32_t s_hot_cnt[(NT / 64) * HOT_N];
    __shared__ uint32_t s_hist[(NT / 64) * 512];
    __shared__ uint64_t s_roll[64 * WK];
    __shared__ uint64_t s_homh[4 * WK];
    extern __shared__ uint64_t s_lut[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lut_words = p.groups * (1 << p.g) * WK;
    for (int i = tid; i < lut_words; i += NT) s_lut[i] = p.lut[i];
    for (int i = tid; i < (NT / 64) * 512; i += NT) s_hist[i] = 0;
    for (int i = tid; i < 64 * WK; i += NT) s_roll[i] = p.roll[i];
    if (tid < 4) {
        uint64_t x[WK], hh[WK];
#pragma unroll
        for (int t = 0; t < WK; ++t) x[t] = 0x5555555555555555ULL * (uint64_t)tid;
        x[WK - 1] &= p.top_mask;
        hash_apply<WK>(p, p.lut, x, hh);
#pragma unroll
        for (int t = 0; t < WK; ++t) s_homh[tid * WK + t] = hh[t];
    }
    if (tid < (NT / 64) * HOT_N) s_hot_cnt[tid] = 0;
    const uint32_t G = gridDim.x * (NT / 64), region = blockIdx.x * (NT / 64) + wave;
    uint64_t *my_log = log + (uint64_t)region * log_cap * RW;
    uint32_t *my_hist = s_hist + wave * 512;
    uint32_t fill = 0;  // wave-uniform, in records
    const uint32_t cap32 = (uint32_t)min(log_cap, (uint64_t)0xFFFFFFFFu);
    const TableParams *pk = (const TableParams *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint64_t lt = (1ULL << lane) - 1ULL;
    lds_barrier();
    for (uint32_t r = region; r < nregions; r += G) {
        const uint32_t nr = (uint32_t)min((uint64_t)desc_cnt[r], desc_cap);
        const uint4 *rd = desc + (uint64_t)r * desc_cap * DU;
        for (uint32_t base = 0; base < nr; base += 64u) {
            uint32_t w[DU * 4];
#pragma unroll
            for (int t = 0; t < DU * 4; ++t) w[t] = 0;
            if (base + (uint32_t)lane < nr) {
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    const uint4 q = rd[(uint64_t)(base + lane) * DU + u];
                    w[4 * u] = q.x; w[4 * u + 1] = q.y; w[4 * u + 2] = q.z; w[4 * u + 3] = q.w;
                }
            }
            const uint32_t cw0 = w[0], inc = w[2 * WK], vm = w[2 * WK + 1];
            uint64_t h[WK];
#pragma unroll
            for (int t = 0; t < WK; ++t) h[t] = 0;
            if (vm) {
                uint64_t x[WK];
#pragma unroll
                for (int t = 0; t < WK; ++t) x[t] = (uint64_t)w[2 * t] | ((uint64_t)w[2 * t + 1] << 32);
                hash_apply<WK>(p, (const uint64_t *)s_lut, x, h);
            }
            uint32_t homcnt = 0;   // four 8-bit counters: homopolymer k-mers of base b seen in this strip
            for (uint32_t j = 0; j < 16; ++j) {
                const bool valid = (vm >> j) & 1u;
                const uint32_t ob = __builtin_amdgcn_ubfe(cw0, 2u * j, 2u);
                bool hom = valid && h[0] == s_homh[ob * WK];
                if (hom) {
    #pragma unroll
                    for (int t = 1; t < WK; ++t) hom &= (h[t] == s_homh[ob * WK + t]);
                }
                homcnt += hom ? (1u << (8u * ob)) : 0u;
                const bool em = valid && !hom;
                const unsigned long long mk = __ballot(em);
                if (mk) {
                    if (em) {
                        const uint32_t at = fill + (uint32_t)__builtin_popcountll(mk & lt);
                        if (at < cap32) {
                            uint64_t *o = my_log + (uint64_t)at * RW;
    #pragma unroll
                            for (int t = 0; t < RW; ++t) o[t] = (t < WK) ? h[t < WK ? t : 0] : 0ULL;
                            atomicAdd(&my_hist[(uint32_t)(h[0] >> hist_shift) & (hist_nb - 1)], 1u);
                        } else if (!(dbg & 1)) {
                            uint64_t rec[RW];
    #pragma unroll
                            for (int t = 0; t < RW; ++t) rec[t] = (t < WK) ? h[t < WK ? t : 0] : 0ULL;
                            defer_append<RW>(pk, rec, 1);   // region full
                        }
                    }
                    fill += (uint32_t)__builtin_popcountll(mk);
                }
                if (j < 15) {
                    const uint32_t idx = ((uint32_t)h[0] & 3u) | (ob << 2) | (__builtin_amdgcn_ubfe(inc, 2u * j, 2u) << 4);
    #pragma unroll
                    for (int t = 0; t < WK; ++t) {
                        uint64_t v = h[t] >> 2;
                        if (t + 1 < WK) v |= h[t + 1] << 62;
                        h[t] = v ^ s_roll[idx * WK + t];
                    }
                }
            }
            if (__ballot(homcnt != 0u)) {
                for (uint32_t b = 0; b < 4; ++b) {
                    uint32_t tot = (homcnt >> (8u * b)) & 0xFFu;
                    if (__ballot(tot != 0u) == 0ULL) continue;
                    for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d, 64);
                    if (lane == 0) {
                        uint64_t *hkey = s_hot_key + (size_t)wave * HOT_N * WK;
                        uint32_t *hcnt = s_hot_cnt + wave * HOT_N;
                        int at = -1;
                        for (int q = 0; q < HOT_N && at < 0; ++q) {
                            if (!hcnt[q]) continue;
                            bool same = true;
                            for (int t = 0; t < WK; ++t) same &= (hkey[q * WK + t] == s_homh[b * WK + t]);
                            if (same) at = q;
                        }
                        if (at < 0)
                            for (int q = 0; q < HOT_N; ++q)
                                if (!hcnt[q]) {
                                    at = q;
                                    for (int t = 0; t < WK; ++t) hkey[q * WK + t] = s_homh[b * WK + t];
                                    break;
                                }
                        if (at >= 0 && (uint64_t)hcnt[at] + tot < 0xFFFFFFF0ULL) hcnt[at] += tot;
                        else if (!(dbg & 1)) {
                            uint64_t rec[RW];
                            for (int t = 0; t < RW; ++t) rec[t] = (t < WK) ? s_homh[b * WK + (t < WK ? t : 0)] : 0ULL;
                            defer_append<RW>(pk, rec, tot);
                        }
                    }
                }
            }
        }
    }
    lds_barrier();
    if (tid < (NT / 64) * HOT_N && s_hot_cnt[tid] && !(dbg & 1)) {
        uint64_t rec[RW];
        for (int t = 0; t < RW; ++t) rec[t] = (t < WK) ? s_hot_key[(size_t)tid * WK + (t < WK ? t : 0)] : 0ULL;
        defer_append<RW>(pk, rec, s_hot_cnt[tid]);
    }
    if (lane == 0) log_cnt[region] = min(fill, cap32);   // (k-mers are counted by strip_desc_wide_kernel)
    for (uint32_t b = lane; b < hist_nb; b += 64) hist[(size_t)b * G + region] = my_hist[b];
}


// addKmer for encoded k-mers already on the device (API batches, merge inserts).
template <int WK>
__global__ __launch_bounds__(NT) void add_kmers_kernel(TableParams p, const uint64_t *kmers,
                                                       const uint64_t *counts, uint64_t n) {
    unsigned long long added = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * NT) {
        uint64_t x[WK], h[WK];
#pragma unroll
        for (int t = 0; t < WK; ++t) x[t] = kmers[i * WK + t];
        x[WK - 1] &= p.top_mask;
        const uint64_t d = counts ? counts[i] : 1ULL;
        if (d == 0) continue;
        hash_apply<WK>(p, p.lut, x, h);
        if (p.lg != p.l && owner_shard<WK>(p, h) != p.shard) continue;  // another GPU's slot range
        insert_key<WK>(p, h, d);
        added += d;
    }
    for (int d = 32; d > 0; d >>= 1) added += __shfl_down(added, d, 64);
    if ((threadIdx.x & 63) == 0 && added) atomicAdd(&p.stats[ST_KMERS], added);
}

template <int WK>
__global__ __launch_bounds__(NT) void get_counts_kernel(TableParams p, const uint64_t *kmers, uint64_t n,
                                                        uint64_t *out, uint64_t *pos_out) {
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * NT) {
        uint64_t x[WK], h[WK];
#pragma unroll
        for (int t = 0; t < WK; ++t) x[t] = kmers[i * WK + t];
        x[WK - 1] &= p.top_mask;
        hash_apply<WK>(p, p.lut, x, h);
        out[i] = lookup_key<WK>(p, h, pos_out ? pos_out + i : nullptr);
    }
}

// getKmerStarts (TSXHashMap.h:650-658) as a bitmap: bit (i & 7) of byte (i >> 3) = slot i is occupied.
__global__ __launch_bounds__(NT) void kmer_starts_kernel(TableParams p, uint8_t *bits, uint64_t nbytes) {
    const uint64_t slots = p.slot_mask + 1;
    for (uint64_t b = (uint64_t)blockIdx.x * NT + threadIdx.x; b < nbytes; b += (uint64_t)gridDim.x * NT) {
        uint32_t v = 0;
        for (uint32_t j = 0; j < 8; ++j) {
            const uint64_t i = b * 8 + j;
            if (i < slots && p.table[i * (uint64_t)p.W] != 0) v |= 1u << j;
        }
        bits[b] = (uint8_t)v;
    }
}

// getKmerCount(): occupied primary slots -> stats[ST_SCRATCH]; occupied
// secondary slots -> stats[ST_SCRATCH2]; sum of every stored count (in-slot counters plus
// carries << C) -> stats[ST_SCRATCH3].
__global__ __launch_bounds__(NT) void occupied_kernel(TableParams p) {
    unsigned long long c = 0, c2 = 0, cs = 0;
    const uint64_t slots = p.slot_mask + 1, sslots = p.sec_mask + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * NT) {
        const uint64_t v = p.table[i * (uint64_t)p.W];
        c += (v != 0) ? 1ULL : 0ULL;
        cs += v >> p.cshift;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * NT + threadIdx.x; i < sslots; i += (uint64_t)gridDim.x * NT) {
        const bool used = p.sec_keys[i] != 0;
        c2 += used ? 1ULL : 0ULL;
        cs += used ? ((unsigned long long)p.sec_cnt[i] << p.C) : 0ULL;
    }
    for (int d = 32; d > 0; d >>= 1) {
        c += __shfl_down(c, d, 64); c2 += __shfl_down(c2, d, 64); cs += __shfl_down(cs, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (c) atomicAdd(&p.stats[ST_SCRATCH], c);
        if (c2) atomicAdd(&p.stats[ST_SCRATCH2], c2);
        if (cs) atomicAdd(&p.stats[ST_SCRATCH3], cs);
    }
}

// Rebuild the k-mer stored in slot `pos` (TSXHashMap::getAllKmers,
// TSXHashMap.h:660-722): hashed key = func bits | (pos - i(i+1)/2 mod 2^l),
// then the inverse mapping.
template <int WK>
__device__ __forceinline__ void slot_to_kmer(const TableParams &p, uint64_t pos, uint64_t (&x)[WK],
                                             uint64_t &count) {
    const uint64_t *e = p.table + pos * (uint64_t)p.W;
    const uint64_t v = e[0];
    const uint32_t i = (uint32_t)(v & ((1ULL << p.R) - 1ULL));
    const uint64_t pos0 = (pos & ~p.seg_mask) | ((pos - (((uint64_t)i * (i + 1)) >> 1)) & p.seg_mask);
    // fr = key bits above the reprobe field: limb-0 part, then limbs 1..W-1 placed at bit K0
    uint64_t fr[6] = {0, 0, 0, 0, 0, 0};
    fr[0] = v & p.k0mask & ~((1ULL << p.R) - 1ULL);
    for (int t = 1; t < p.W; ++t) {
        const uint64_t hv = e[t];
        fr[t - 1] |= hv << p.K0;
        fr[t] |= hv >> (64 - p.K0);
    }
    // func = fr >> R ; h = (func << l) | pos0
    uint64_t h[WK];
    uint64_t func[6];
#pragma unroll
    for (int t = 0; t < 5; ++t) func[t] = (fr[t] >> p.R) | (fr[t + 1] << (64 - p.R));
    func[5] = 0;
#pragma unroll
    for (int t = 0; t < WK; ++t) {
        uint64_t hv = func[t] << p.lg;
        if (t > 0) hv |= func[t - 1] >> (64 - p.lg);
        h[t] = hv;
    }
    h[0] |= pos0 | ((uint64_t)p.shard << p.l);
    h[WK - 1] &= p.top_mask;
    hash_apply<WK>(p, p.ilut, h, x);
    count = (v >> p.cshift) + (sec_get(p, pos) << p.C);
}

__device__ __forceinline__ int owner_of(const uint64_t *x, int wk, int nranks) {
    uint64_t z = 0x243F6A8885A308D3ULL;
    for (int t = 0; t < wk; ++t) z = mix64(z ^ x[t]);
    return (int)((z >> 32) * (uint64_t)nranks >> 32);
}

// mode 0: count occupied slots per owner into seg[0..nranks)
// mode 1: write k-mer + count at seg_cursor[owner]++ (cursors preset to segment starts)
// One atomic per (wave, owner present in the wave): the lanes that share an owner are found
// with ballots and take consecutive places behind the leader's atomicAdd.
template <int WK>
__global__ __launch_bounds__(NT) void dump_kernel(TableParams p, int nranks, int mode, uint64_t *kmers_out,
                                                  uint64_t *counts_out, uint64_t cap,
                                                  unsigned long long *seg, uint64_t slot_lo, uint64_t slot_hi) {
    const uint64_t slots = slot_hi;
    const int lane = threadIdx.x & 63;
    const uint64_t lt = (1ULL << lane) - 1ULL;
    // wave-uniform trip count: every lane of a wave runs the same iterations
    for (uint64_t base = slot_lo + (uint64_t)blockIdx.x * NT + (threadIdx.x & ~63u); base < slots; base += (uint64_t)gridDim.x * NT) {
        const uint64_t pos = base + lane;
        const bool occ = pos < slots && p.table[pos * (uint64_t)p.W] != 0;
        uint64_t x[WK], c = 0;
        int own = 0;
        if (occ) {
            slot_to_kmer<WK>(p, pos, x, c);
            own = (nranks > 1) ? owner_of(x, WK, nranks) : 0;
        }
        uint64_t todo = __ballot(occ);
        unsigned long long at = 0;
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int o = __shfl(own, lead, 64);
            const uint64_t grp = __ballot(occ && own == o);
            unsigned long long b = 0;
            if (lane == lead) b = atomicAdd(&seg[o], (unsigned long long)__popcll(grp));
            b = __shfl(b, lead, 64);
            if (occ && own == o) at = b + (unsigned long long)__popcll(grp & lt);
            todo &= ~grp;
        }
        if (mode == 1 && occ && at < cap) {
#pragma unroll
            for (int t = 0; t < WK; ++t) kmers_out[at * WK + t] = x[t];
            counts_out[at] = c;
        }
    }
}

// Synthetic FASTQ: one workgroup per read, record layout
// "@seq<i>\n" bases "\n+\n" qualities "\n"  (generateFakeSequences.py:17-20).
__device__ __host__ inline uint64_t synth_mix(uint64_t seed, uint64_t i, uint64_t c) {
    uint64_t z = seed + i * 0x9E3779B97F4A7C15ULL + c * 0xD1B54A32D192ED03ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(NT) void synth_fill_kernel(uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                                        const uint64_t *offsets, uint8_t *out) {
    for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint64_t id = first_read + r;
        const uint32_t nrand = 500u + (uint32_t)(synth_mix(seed, id, 0) % 501u);
        const uint32_t na = 100u + (uint32_t)(synth_mix(seed, id, 1) % 201u);
        const uint32_t L = nrand + na;
        uint8_t *o = out + offsets[r];
        // header
        char digits[24]; int nd = 0; uint64_t v = id;
        do { digits[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
        const uint32_t hl = 4u + (uint32_t)nd + 1u;
        if (threadIdx.x == 0) {
            o[0] = '@'; o[1] = 's'; o[2] = 'e'; o[3] = 'q';
            for (int d = 0; d < nd; ++d) o[4 + d] = (uint8_t)digits[nd - 1 - d];
            o[4 + nd] = '\n';
        }
        uint8_t *seq = o + hl;
        for (uint32_t j = threadIdx.x; j < L; j += NT) {
            uint8_t ch = 'A';
            if (j < nrand) {
                const uint64_t w = synth_mix(seed, id, 2 + (j >> 5));
                ch = (uint8_t)"ACGT"[(w >> (2 * (j & 31))) & 3];
            }
            seq[j] = ch;
            seq[L + 3 + j] = '&';
        }
        if (threadIdx.x == 0) { seq[L] = '\n'; seq[L + 1] = '+'; seq[L + 2] = '\n'; seq[2 * L + 3] = '\n'; }
    }
}

// Zipf-skewed reads (BASELINE config 4): n_templates random template sequences of 2 * read_len bases, read r is the
// window [start_r, start_r + read_len) of template pick_r, pick_r drawn with Zipf rank weights: thr[t] = the upper end of
// template t's share of [0, 2^64) (thr is ascending, the last entry 2^64 - 1), pick_r = number of thr entries < u_r for
// u_r = synth_mix(seed, r, 0).  Record: "@z<r>\n" bases "\n+\n" 'I' * read_len "\n".  tsxcount_amd/synth.py has the numpy twin.
__device__ __host__ inline uint32_t zipf_pick(const uint64_t *thr, uint32_t n_templates, uint64_t u) {
    uint32_t lo = 0, hi = n_templates - 1;   // first t with thr[t] >= u
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (thr[mid] >= u) hi = mid; else lo = mid + 1;
    }
    return lo;
}
__global__ __launch_bounds__(NT) void synth_zipf_kernel(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint32_t n_templates,
                                                        const uint64_t *thr, const uint64_t *offsets, uint8_t *out) {
    for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint32_t t = zipf_pick(thr, n_templates, synth_mix(seed, r, 0));
        const uint32_t start = (uint32_t)(synth_mix(seed, r, 1) % (uint64_t)(read_len + 1));
        uint8_t *o = out + offsets[r];
        char digits[24]; int nd = 0; uint64_t v = r;
        do { digits[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
        const uint32_t hl = 2u + (uint32_t)nd + 1u;
        if (threadIdx.x == 0) {
            o[0] = '@'; o[1] = 'z';
            for (int d = 0; d < nd; ++d) o[2 + d] = (uint8_t)digits[nd - 1 - d];
            o[2 + nd] = '\n';
        }
        uint8_t *seq = o + hl;
        const uint64_t tid64 = 0x5A495046ULL + (uint64_t)t;   // template t draws its bases from a stream of its own
        for (uint32_t j = threadIdx.x; j < read_len; j += NT) {
            const uint32_t q = start + j;
            const uint64_t w = synth_mix(seed ^ 0x7E3779B97F4A7C15ULL, tid64, 2 + (q >> 5));
            seq[j] = (uint8_t)"ACGT"[(w >> (2 * (q & 31))) & 3];
            seq[read_len + 3 + j] = 'I';
        }
        if (threadIdx.x == 0) { seq[read_len] = '\n'; seq[read_len + 1] = '+'; seq[read_len + 2] = '\n'; seq[2 * read_len + 3] = '\n'; }
    }
}

}  // namespace tsx
