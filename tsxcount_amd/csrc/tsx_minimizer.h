// tsx_minimizer.h -- multi-GPU counting with owner = f(minimizer of the k-mer): the sender side.
//
// A table sharded by home-slot range (tsx_hip_shard_*) scatters consecutive k-mers of a read over all GPUs, so either every
// key travels (8 B per occurrence) or every GPU rolls over every GPU's text (description exchange: N x the walk).  Here the
// owner of a k-mer is a function of its MINIMIZER -- the m-mer of the k-mer with the smallest hash, m = min(11, k - 15) --
// so runs of consecutive k-mers (12 on random sequence at k = 31) share an owner: what travels is strip descriptions masked
// per owner, 16 bytes per (strip, owner present in it), about 1.65 per strip of 16 starts at 8 GPUs, and every GPU walks only
// what it owns into a table of its own (no slot-range split, no merge; a lookup goes to mz_owner_of_kmer(kmer)).
//
//   desc_owner_split_kernel    strip descriptions (strip_desc_kernel<true>'s wave regions, a share of them per call) -> one
//                              list per owner GPU, chunk by chunk
//   desc_owner_finish_kernel   the lists' lengths; chunks no workgroup reached get descriptions without a valid start
//
// Homopolymer k-mers (poly-A tails: a sixth of all occurrences in the reference's synthetic reads, ONE key, one owner) are
// taken out of the descriptions and counted per base when the text is described (strip_desc_kernel<true>, tsx_kernels.h); the
// first share reports the totals, the caller sends them to their owners.
// 20 <= k <= 32 (the 16 windows of a strip share the m-mers 15 .. k - m of the strip: w = k - m + 1 >= 16).
#pragma once
#include "tsx_kernels.h"

namespace tsx {

constexpr uint32_t MZ_MULT = 0x9E3779u;    // odd, < 2^24: x -> x * MZ_MULT + MZ_SALT (mod 4^m) is a bijection of the m-mers,
constexpr uint32_t MZ_SALT = 0x2B5A3Du;    //   the ORDER of its values is the minimizer order (poly-A is nothing special in it)
constexpr uint32_t MZ_MULT2 = 0xC2B2AFu;   // owner = a middle slice of (value * MZ_MULT2), scaled to the number of GPUs
constexpr int MZ_MAX_RANKS = 16;
constexpr uint32_t MZ_CHUNK = 64;          // an owner's list is made of chunks of this many places: chunk j * G + g is workgroup g's j-th
constexpr int MZ_NT = 1024;               // one workgroup of 1024 per CU: a list is as long as its busiest workgroup made it, and the
constexpr int MZ_WG_PER_CU = 1;           //   shares of 256 big workgroups vary less than those of 1024 small ones (5 % holes instead of 11 %)
constexpr int MZ_ROUND = 5;                // owners of a strip placed per round (more -- 0.05 % of the strips at 8 GPUs -- : another round)
static_assert(MZ_NT % (int)MZ_CHUNK == 0, "the hole filler takes whole chunks");

__host__ __device__ inline bool mz_supported(uint32_t k) { return k >= 20u && k <= 32u; }
__host__ __device__ inline uint32_t mz_m(uint32_t k) { return (k - 15u < 11u) ? k - 15u : 11u; }
// the value of an m-mer (x: its 2m bits, first base lowest): bit 31 = it starts or ends with AAA (such m-mers -- poly-A
// tails and what borders them, the same few in every read -- come after all others), below it the bijective mix
__host__ __device__ inline uint32_t mz_key(uint32_t x, uint32_t mbits) {
    const uint32_t mmask = (1u << mbits) - 1u;
    const uint32_t pen = ((x & 63u) == 0u || (x >> (mbits - 6u)) == 0u) ? 0x80000000u : 0u;
    return pen | (((x * MZ_MULT + MZ_SALT) & mmask) << (31u - mbits));
}
__host__ __device__ inline uint32_t mz_owner(uint32_t key, uint32_t mbits, uint32_t nranks) {
    const uint32_t t = key >> (31u - mbits);      // < 2^23
    const uint32_t u = ((t * MZ_MULT2) >> 8) & 0xFFFFu;
    return (u * nranks) >> 16;
}
// the reference form: owner of one k-mer (k <= 32, base i at bits 2i)
__host__ __device__ inline uint32_t mz_owner_of_kmer(uint64_t x, uint32_t k, uint32_t nranks) {
    const uint32_t m = mz_m(k), mbits = 2u * m, w = k - m + 1u;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < w; ++j) {
        const uint32_t key = mz_key((uint32_t)(x >> (2u * j)) & ((1u << mbits) - 1u), mbits);
        best = key < best ? key : best;
    }
    return mz_owner(best, mbits, nranks);
}

// v_mul_u32_u24 (full rate; the compiler takes the quarter-rate 32-bit multiply where it cannot see the operands' width)
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(MZ_NT) void desc_owner_split_kernel(TableParams p, const uint4 *desc, uint64_t desc_cap,
                                                                    const unsigned long long *desc_cnt, uint32_t nregions,
                                                                    uint32_t nranks, uint4 *out, uint64_t out_cap,
                                                                    uint32_t *used, unsigned long long *hom_cnt, int merge,
                                                                    uint32_t part, uint32_t nparts,
                                                                    const unsigned long long *hom_pre) {
    __shared__ uint32_t s_cnt[MZ_MAX_RANKS];    // descriptions of this round per owner
    __shared__ uint32_t s_base[MZ_MAX_RANKS];   // descriptions of this workgroup for the owner before this round
    __shared__ uint32_t s_fill[MZ_MAX_RANKS];   //   ... including it
    __shared__ uint32_t s_more[2];              // some lane has owners left for another round
    __shared__ uint4 *s_list[MZ_MAX_RANKS];     // where each owner's list begins
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t G = gridDim.x, g = blockIdx.x;
    if (tid < MZ_MAX_RANKS) { s_cnt[tid] = 0; s_fill[tid] = 0; s_list[tid] = out + (uint64_t)tid * out_cap; }
    if (tid < 2) s_more[tid] = 0;
    const uint32_t k = (uint32_t)p.k;
    const uint32_t m = mz_m(k), mbits = 2u * m, w = k - m + 1u, sh = 31u - mbits;
    const bool pow2 = (nranks & (nranks - 1u)) == 0u;
    const uint32_t own_sh = 24u - (uint32_t)__builtin_ctz(nranks | 0x10000u);   // (power of two: log2)
    unsigned long long lost = 0;
    uint32_t round = 0;
    lds_barrier();
    // mz_key of the m-mer at `bit` of (lo, hi); pa: bit 2j of it = the m-mer at base j of lo starts or ends with AAA
    auto keyat = [&](uint32_t lo, uint32_t hi, uint32_t pa, uint32_t bit) -> uint32_t {
        const uint32_t x = __funnelshift_r(lo, hi, bit);   // (what lies above the m-mer only reaches product bits >= 2m)
        const uint32_t v = (__umul24(x, MZ_MULT) + MZ_SALT) << sh;      // (bits from 2m on leave at the top or land on bit 31)
        return (v & 0x7FFFFFFFu) | ((pa << (31u - bit)) & 0x80000000u);   // one v_bfi_b32
    };
    // place n of this workgroup's descriptions for owner o: its chunks are g, G + g, 2 G + g, ... of the list
    // (32-bit arithmetic: a list holds less than 2^32 descriptions -- a text is described in pieces below 4 GiB)
    const uint32_t cap32 = (uint32_t)min(out_cap, (uint64_t)0xFFFFFFFFu);
    auto place = [&](uint32_t n) -> uint32_t { return ((n / MZ_CHUNK) * G + g) * MZ_CHUNK + (n % MZ_CHUNK); };
    auto put = [&](uint32_t o, uint32_t at, const uint4 v) {
        if (at < cap32) s_list[o][at] = v;
        else ++lost;
    };
    for (uint32_t r = g; r < nregions; r += G) {
        // share `part` of `nparts` of every region (a text is described once and split window by window: the exchange of
        // one window runs while the next is split and the one before is walked)
        const uint32_t nall = (uint32_t)min((uint64_t)desc_cnt[r], desc_cap);
        const uint32_t lo = (uint32_t)((uint64_t)nall * part / nparts), nr = (uint32_t)((uint64_t)nall * (part + 1u) / nparts) - lo;
        const uint4 *rd = desc + (uint64_t)r * desc_cap + lo;
        uint4 dn = make_uint4(0, 0, 0, 0);
        if (tid < nr) dn = rd[tid];
        for (uint32_t base = 0; base < nr; base += MZ_NT) {
            const uint4 d = dn;
            dn = make_uint4(0, 0, 0, 0);
            if (base + MZ_NT + tid < nr) dn = rd[base + MZ_NT + tid];
            const uint32_t c0 = d.x, c1 = d.y, c2 = d.z;
            uint32_t vm = d.w & 0xFFFFu;
            uint32_t P0 = 0, P1 = 0, P2 = 0, P3 = 0;
            if (__ballot(vm != 0u)) {
                // (homopolymer k-mers left the validity bits when the text was described: strip_desc_kernel<true>)
                // ---- the minimizer of each of the 16 windows: min over the m-mers i .. i + w - 1 of the strip =
                //      min(suffix minimum of i .. 14, the shared core 15 .. w - 1, prefix minimum of w .. w + i - 1)
                uint32_t pa0, pa1, pa2;   // bit 2j: the m-mer at base j starts or ends with AAA
                {
                    const uint32_t z0 = ~(c0 | (c0 >> 1)) & 0x55555555u, z1 = ~(c1 | (c1 >> 1)) & 0x55555555u,
                                   z2 = ~(c2 | (c2 >> 1)) & 0x55555555u;     // bit 2j: base j is A
                    const uint32_t a0 = z0 & __funnelshift_r(z0, z1, 2u) & __funnelshift_r(z0, z1, 4u),
                                   a1 = z1 & __funnelshift_r(z1, z2, 2u) & __funnelshift_r(z1, z2, 4u),
                                   a2 = z2 & (z2 >> 2) & (z2 >> 4);          // bit 2j: bases j, j + 1, j + 2 are A
                    const uint32_t e = 2u * (m - 3u);                        // 4 .. 16: AAA at the END of the m-mer at base j
                    pa0 = a0 | __funnelshift_r(a0, a1, e);
                    pa1 = a1 | __funnelshift_r(a1, a2, e);
                    pa2 = a2 | (a2 >> e);
                }
                uint32_t suf[15];
                {
                    uint32_t s = 0xFFFFFFFFu;
#pragma unroll
                    for (int j = 14; j >= 0; --j) { s = min(s, keyat(c0, c1, pa0, 2u * (uint32_t)j)); suf[j] = s; }
                }
                uint32_t core = keyat(c0, c1, pa0, 30u);   // m-mer 15
#pragma unroll
                for (uint32_t j = 16; j < 22; ++j) {
                    const uint32_t kj = keyat(c1, c2, pa1, 2u * j - 32u);
                    core = (j < w) ? min(core, kj) : core;
                }
                const uint32_t bsh = 2u * w - 32u;                      // the strip from m-mer w on (50 bits are enough)
                const uint32_t D0 = __funnelshift_r(c1, c2, bsh), D1 = c2 >> bsh, PD = __funnelshift_r(pa1, pa2, bsh);
                uint32_t pre = 0xFFFFFFFFu;
#pragma unroll
                for (uint32_t i = 0; i < 16; ++i) {
                    if (i > 0) pre = min(pre, keyat(D0, D1, PD, 2u * (i - 1u)));
                    uint32_t best = min(core, pre);
                    if (i < 15) best = min(best, suf[i]);
                    const uint32_t t = best >> sh;
                    const uint32_t pr = mul24(t, MZ_MULT2);                  // (t < 2^23)
                    // mz_owner: bits 8..23 of the product, scaled to the number of GPUs -- for a power of two its top bits
                    const uint32_t o = pow2 ? (pr >> own_sh) & (nranks - 1u) : mul24((pr >> 8) & 0xFFFFu, nranks) >> 16;
                    P0 |= (o & 1u) << i;
                    P1 |= ((o >> 1) & 1u) << i;
                    P2 |= ((o >> 2) & 1u) << i;
                    if (nranks > 8u) P3 |= ((o >> 3) & 1u) << i;
                }
            }
            // ---- a run of one owner that crosses into the next strip of the text (lane + 1, when strip_desc_kernel marked it
            //      as this strip's neighbour) travels as ONE description from its first start, if it is at most 16 starts long:
            //      T = length of the run that ends at start 15, Ld = length of the run that begins at start 0
            uint32_t xm = 0, xo = 0, x0 = 0, x1 = 0, x2 = 0;     // the description of such a run
            if (merge) {
                const uint32_t eqn = ~((P0 ^ (P0 >> 1)) | (P1 ^ (P1 >> 1)) | (P2 ^ (P2 >> 1)) | (P3 ^ (P3 >> 1)));
                const uint32_t cont = eqn & vm & (vm >> 1) & 0x7FFFu;    // bit i: starts i and i + 1 valid, one owner
                uint32_t T = 0, Ld = 0;
                if (vm >> 15) { const uint32_t y = ~cont << 17; T = 1u + (y ? (uint32_t)__clz((int)y) : 15u); }
                if (vm & 1u) { const uint32_t nz = ~cont & 0x7FFFu; Ld = 1u + (nz ? (uint32_t)__ffs((int)nz) - 1u : 15u); }
                const uint32_t ot = ((P0 >> 15) & 1u) | (((P1 >> 15) & 1u) << 1) | (((P2 >> 15) & 1u) << 2) | (((P3 >> 15) & 1u) << 3);
                const uint32_t ol = (P0 & 1u) | ((P1 & 1u) << 1) | ((P2 & 1u) << 2) | ((P3 & 1u) << 3);
                const uint32_t adj = (d.w >> 16) & 1u;
                const uint32_t lead = Ld | (ol << 8) | (adj << 16), tail = T | (ot << 8);
                const uint32_t lead_n = __shfl_down(lead, 1, 64), c2_n = __shfl_down(c2, 1, 64), tail_p = __shfl_up(tail, 1, 64);
                const uint32_t L_n = lead_n & 0xFFu, T_p = tail_p & 0xFFu;
                const bool fwd = lane < 63u && (lead_n >> 16) && T > 0u && L_n > 0u && ot == ((lead_n >> 8) & 0xFFu) && T + L_n <= 16u;
                const bool bwd = lane > 0u && adj && T_p > 0u && Ld > 0u && (tail_p >> 8) == ol && T_p + Ld <= 16u;
                if (bwd) vm &= ~((1u << Ld) - 1u);            // the strip before this one took them
                if (fwd) {
                    vm &= ~((0xFFFFu << (16u - T)) & 0xFFFFu);
                    const uint32_t sft = 2u * (16u - T);      // 2 .. 30
                    x0 = __funnelshift_r(c0, c1, sft); x1 = __funnelshift_r(c1, c2, sft); x2 = __funnelshift_r(c2, c2_n, sft);
                    xm = (1u << (T + L_n)) - 1u;
                    xo = ot;
                }
            }
            // ---- one description per owner present in the strip: places by LDS counters, MZ_ROUND owners per round
            uint32_t left = vm;
            uint32_t again;
            uint32_t xi = 0;
            if (xm) xi = atomicAdd(&s_cnt[xo], 1u);
            do {
                uint32_t mo[MZ_ROUND], oo[MZ_ROUND], ix[MZ_ROUND];
#pragma unroll
                for (int t = 0; t < MZ_ROUND; ++t) {
                    mo[t] = 0; oo[t] = 0; ix[t] = 0;
                    if (left) {
                        const uint32_t i0 = (uint32_t)__ffs((int)left) - 1u;
                        const uint32_t b0 = (P0 >> i0) & 1u, b1 = (P1 >> i0) & 1u, b2 = (P2 >> i0) & 1u, b3 = (P3 >> i0) & 1u;
                        uint32_t msk = left;
                        msk &= P0 ^ (b0 - 1u);   // (bit set: the plane itself, else its complement)
                        msk &= P1 ^ (b1 - 1u);
                        msk &= P2 ^ (b2 - 1u);
                        msk &= P3 ^ (b3 - 1u);
                        const uint32_t o = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3);
                        mo[t] = msk; oo[t] = o;
                        ix[t] = atomicAdd(&s_cnt[o], 1u);
                        left &= ~msk;
                    }
                }
                const uint32_t par = round & 1u;
                ++round;
                if (left) s_more[par] = 1u;
                lds_barrier();
                if (tid < nranks) {
                    const uint32_t c = s_cnt[tid], f = s_fill[tid];
                    s_cnt[tid] = 0; s_base[tid] = f; s_fill[tid] = f + c;
                }
                if (tid == 0) s_more[par ^ 1u] = 0;
                again = s_more[par];
                lds_barrier();
#pragma unroll
                for (int t = 0; t < MZ_ROUND; ++t)
                    if (mo[t]) {
                        put(oo[t], place(s_base[oo[t]] + ix[t]), make_uint4(c0, c1, c2, mo[t]));
                    }
                if (xm) {
                    put(xo, place(s_base[xo] + xi), make_uint4(x0, x1, x2, xm));
                    xm = 0;
                }
            } while (again);
        }
    }
    lds_barrier();
    // the rest of the workgroup's last chunks: descriptions without a valid start (the lists stay packed runs)
    for (uint32_t o = 0; o < nranks; ++o) {
        const uint32_t f = s_fill[o], end = (f + MZ_CHUNK - 1u) / MZ_CHUNK * MZ_CHUNK;
        for (uint32_t n = f + tid; n < end; n += MZ_NT) {
            const uint32_t at = place(n);
            if (at < cap32) s_list[o][at] = make_uint4(0, 0, 0, 0);
        }
    }
    if (tid < nranks) used[(uint64_t)tid * G + g] = (s_fill[tid] + MZ_CHUNK - 1u) / MZ_CHUNK;
    // (what strip_desc_kernel<true> took out when the text was described: reported with the first share)
    if (hom_pre && part == 0u && blockIdx.x == 0 && tid < 4 && hom_pre[tid]) atomicAdd(&hom_cnt[tid], hom_pre[tid]);
    {
        unsigned long long t = lost;
        for (int dd = 32; dd > 0; dd >>= 1) t += __shfl_down(t, dd, 64);
        if (lane == 0 && t) atomicAdd(&p.stats[ST_FAIL], t);
    }
}

// Workgroup g of desc_owner_split_kernel filled chunks g, G + g, 2 G + g, ... of every list, used[o * G + g] of them.  Rows
// (a row = the G chunks j * G .. j * G + G - 1) below the least busy workgroup's count are full; the rows above it, up to the
// busiest workgroup's count, are ragged -- about 5 % of a list would be chunks nobody reached.  One workgroup per list closes
// them up: the R used chunks of the ragged rows come to lie in its first R places (the holes among those places take the used
// chunks that lie behind them: as many of the one as of the other, sources and destinations apart), and the list is
// min * G + R chunks long, every one of them used.  count[o] = its descriptions.  (Ragged rows beyond MZ_RAGGED places, or
// more than 1024 workgroups: the holes get descriptions without a valid start instead, as in the first version.)
constexpr uint32_t MZ_RAGGED = 8192;
__global__ __launch_bounds__(MZ_NT) void desc_owner_finish_kernel(const uint32_t *used, uint32_t G, uint32_t nranks, uint4 *out,
                                                                  uint64_t out_cap, unsigned long long *count,
                                                                  unsigned long long *stats) {
    __shared__ uint32_t s_used[1024];
    __shared__ uint16_t s_hole[MZ_RAGGED], s_move[MZ_RAGGED];
    __shared__ uint32_t s_min, s_max, s_sum, s_nh, s_nm;
    const uint32_t o = blockIdx.x, tid = threadIdx.x;
    uint4 *list = out + (uint64_t)o * out_cap;
    if (tid == 0) { s_min = 0xFFFFFFFFu; s_max = 0; s_sum = 0; s_nh = 0; s_nm = 0; }
    __syncthreads();
    {
        uint32_t mn = 0xFFFFFFFFu, mx = 0, sum = 0;
        for (uint32_t g = tid; g < G; g += MZ_NT) {
            const uint32_t u = used[(uint64_t)o * G + g];
            if (g < 1024u) s_used[g] = u;
            mn = min(mn, u); mx = max(mx, u); sum += u;
        }
        atomicMin(&s_min, mn); atomicMax(&s_max, mx); atomicAdd(&s_sum, sum);
    }
    __syncthreads();
    const uint32_t mn = s_min, mx = s_max;
    const uint32_t T = (mx - mn) * G;          // places of the ragged rows
    const uint32_t R = s_sum - mn * G;         // used chunks among them
    const bool close_up = G <= 1024u && T <= MZ_RAGGED;
    unsigned long long total = (unsigned long long)(close_up ? mn * G + R : mx * G) * MZ_CHUNK;
    if (total > out_cap) {
        if (tid == 0 && stats) atomicAdd(&stats[ST_FAIL], total - out_cap);
        total = out_cap;
    }
    if (tid == 0) count[o] = total;
    if (!close_up) {   // holes stay: descriptions without a valid start
        for (uint32_t g = 0; g < G; ++g)
            for (uint32_t j = used[(uint64_t)o * G + g] + tid / MZ_CHUNK; j < mx; j += MZ_NT / MZ_CHUNK) {
                const unsigned long long at = ((unsigned long long)j * G + g) * MZ_CHUNK + tid % MZ_CHUNK;
                if (at < out_cap) list[at] = make_uint4(0, 0, 0, 0);
            }
        return;
    }
    // place s of the ragged rows = chunk (mn + s / G) * G + s % G; used iff mn + s / G < used[s % G]
    for (uint32_t s0 = 0; s0 < T; s0 += MZ_NT) {
        const uint32_t sidx = s0 + tid;
        if (sidx < T) {
            const bool is_used = mn + sidx / G < s_used[sidx % G];
            if (sidx < R && !is_used) s_hole[atomicAdd(&s_nh, 1u)] = (uint16_t)sidx;
            if (sidx >= R && is_used) s_move[atomicAdd(&s_nm, 1u)] = (uint16_t)sidx;
        }
    }
    __syncthreads();
    const uint32_t npair = min(s_nh, s_nm);    // (equal by construction)
    const uint64_t base = (uint64_t)mn * G * MZ_CHUNK;
    for (uint32_t pr = tid / MZ_CHUNK; pr < npair; pr += MZ_NT / MZ_CHUNK) {
        const uint64_t src = base + (uint64_t)s_move[pr] * MZ_CHUNK + tid % MZ_CHUNK, dst = base + (uint64_t)s_hole[pr] * MZ_CHUNK + tid % MZ_CHUNK;
        if (src < out_cap && dst < out_cap) list[dst] = list[src];
    }
}

}  // namespace tsx
