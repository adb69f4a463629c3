// tsx_device.h -- device-side table layout and the insert / lookup primitives
// of the MI355X k-mer counting hash map (gfx950 only, wave64).
//
// Slot layout (DESIGN.md "Data layout in HBM"): a slot is W uint64 limbs.
//   limb 0, bits [0,R)        reprobe count i (>=1 once occupied) -- the low l
//                             bits of the hashed key are implied by the slot
//                             position, as in TSXHashMap::makeKey
//                             (TSXHashMap.h:1056-1072)
//   limb 0, bits [R,K0)       low func bits of the hashed key (TSXTypes.h:39)
//   limb 0, bit  K0           LOCK, only when W > 1
//   limb 0, bits [64-C,64)    in-slot count (the reference's "value"/storage
//                             bits, TSXHashMap.h:83), wraps modulo 2^C
//   limbs 1..W-1              remaining func bits
// A slot is empty iff limb 0 == 0 (TSXHashMap::positionEmpty, TSXHashMap.h:1121).
// Count wrap-around carries into the secondary array keyed by slot position,
// which replaces the reference's in-table overflow chain
// (TSXHashMapPerf.h:699-881): total = count + (carry << C).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsx {

enum StatIdx {
    ST_KMERS = 0,      // k-mer occurrences handed to insert
    ST_FAIL = 1,       // occurrences that ran out of reprobes
    ST_CARRY = 2,      // carries pushed to the secondary array
    ST_SECFAIL = 3,    // carries that found the secondary array full
    ST_LOCKTO = 4,     // bounded lock spins that expired
    ST_SCRATCH = 5,    // reductions (distinct, ...)
    ST_SCRATCH2 = 6,
    ST_FALLBACK = 7,   // keys the partitioned path handed to insert_key (a list or log region was full)
    ST_SCRATCH3 = 8,   // reduction: sum of all counts
    ST_DBG0 = 9,       // diagnostic builds (TSX_HIP_DEBUG bit 4): wave-rounds of the segment build ...
    ST_DBG1 = 10,      // ... shader cycles inside its insert loop, summed over waves
    ST_DBG2 = 11,      // ... lanes that probed, summed over rounds
    ST_DBG3 = 12,      // ... rounds with fewer than 16 probing lanes
    ST_DBG4 = 13,      // ... cycles of those rounds
    ST_N = 16
};

// Keys that cannot take the fast route of the partitioned path (a full log region, a hot k-mer merged on
// chip, a full sub-list whose overflow queue is full too) wait here with their counts until the segment
// build has finished; deferred_insert_kernel then inserts them like the atomic path would.  Nothing in the
// scan and partition kernels touches the table, which is what lets a freshly cleared table skip its
// memset (the build writes every segment).  In a sharded run the list is the caller's hot (key, count)
// list, exchanged between the GPUs.  An entry is `rw` words of key (1, 2 or 4) and a count.
struct DeferList {
    uint64_t *rec;
    uint64_t *cnt;
    unsigned long long *n;   // entries appended so far (may run past cap: the excess is counted as lost)
    uint64_t cap;
};

struct TableParams {
    uint64_t *table;            // slots * W limbs
    uint64_t *sec_keys;         // secondary array: slot position + 1
    uint64_t *sec_cnt;          // secondary array: carry count
    unsigned long long *stats;  // ST_N device counters
    const uint64_t *lut;        // hash LUT   [groups][1<<g][WK]
    const uint64_t *ilut;       // inverse    [groups][1<<g][WK]
    const uint64_t *roll;       // one-limb keys: sliding-window update table [64] (the walk kernels), else null
    uint64_t slot_mask;         // 2^l - 1
    uint64_t seg_mask;          // 2^S - 1: probing never leaves the 2^S-slot segment of its home slot
    uint8_t *seg_dirty;         // one byte per segment: 1 once the segment holds anything
    uint64_t sec_mask;
    uint64_t k0mask;            // key bits of limb 0
    uint64_t lock_bit;          // 0 when W == 1
    uint64_t top_mask;          // valid bits of the top key limb
    int k, l, n, wk, W;
    int R, F, C, K0, cshift;    // cshift = 64 - C
    int S;                      // log2 slots per segment
    int lg;                     // log2 slots of the WHOLE table: l on one GPU, l + shard_bits when the
                                // table is sharded by slot range over 2^shard_bits GPUs
    uint32_t shard;             // this GPU's slot range: home slots [shard << l, (shard + 1) << l)
    int g, groups;              // LUT granularity (4 or 8 bits) and group count
    uint32_t max_reprobes;
    uint32_t line_mask;         // lines per record - 1: 3 = FASTQ (FASTQEntry, FastXReader.h:62-95), 1 = FASTA as
                                // FASTXreader<FASTAEntry> reads it (two lines per record, :97-116); the sequence
                                // is the line with (index & line_mask) == 1
    DeferList defer;            // see DeferList; set per launch by the host
    uint64_t pos_base;          // slot number of table[0] in the table the lookups see: 0, except in the per-slab views of
                                // a table that is built slab by slab (l - S > 18, tsxcount_hip.hip: count_slabs) -- the
                                // secondary array is keyed by that global slot number
};

// Out of line on purpose (rare path; keeps the callers' register budgets): pk points at the kernel's own
// TableParams argument (the kernel-argument segment).  The record travels BY VALUE (in registers): a pointer
// argument would force the caller's copy of the record into scratch memory, on the fast path too.
template <int RW> struct RecVal { uint64_t w[RW]; };
template <int RW>
__device__ __attribute__((noinline)) void defer_append_v(const TableParams *pk, RecVal<RW> r, uint64_t d) {
    const unsigned long long at = atomicAdd(pk->defer.n, 1ULL);
    if (at < pk->defer.cap) {
#pragma unroll
        for (int t = 0; t < RW; ++t) pk->defer.rec[at * RW + t] = r.w[t];
        pk->defer.cnt[at] = d;
    } else {
        atomicAdd(&pk->stats[ST_FAIL], (unsigned long long)d);
    }
}
template <int RW>
__device__ __forceinline__ void defer_append(const TableParams *pk, const uint64_t *r, uint64_t d) {
    RecVal<RW> v;
#pragma unroll
    for (int t = 0; t < RW; ++t) v.w[t] = r[t];
    defer_append_v<RW>(pk, v, d);
}
__device__ __forceinline__ void defer_append1(const TableParams *pk, uint64_t key, uint64_t d) {
    RecVal<1> v;
    v.w[0] = key;
    defer_append_v<1>(pk, v, d);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// every outstanding global load, store and atomic of the wave (s_waitcnt vmcnt(0)),
// which serialises prefetched loads and fire-and-forget stores with the barrier;
// the kernels here exchange data between waves through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Bijective GF(2) mapping by table lookup: A*x = XOR over byte/nibble groups of
// LUT[group][value] (BijectiveKMapping::applyto, BijectiveKMapping.h:202-225,
// evaluates the same product row by row with AND + popcount).
template <int WK, typename LutPtr>
__device__ __forceinline__ void hash_apply(const TableParams &p, LutPtr lut, const uint64_t (&x)[WK],
                                           uint64_t (&h)[WK]) {
#pragma unroll
    for (int t = 0; t < WK; ++t) h[t] = 0;
    const int g = p.g;
    const uint32_t gm = (1u << g) - 1u;
    const int per_limb = 64 / g;
    for (int grp = 0; grp < p.groups; ++grp) {
        const int limb = grp / per_limb;
        const uint32_t v = (uint32_t)(x[limb < WK ? limb : 0] >> ((grp % per_limb) * g)) & gm;
        const uint64_t *e = (const uint64_t *)(lut + ((size_t)((grp << g) + v)) * WK);
#pragma unroll
        for (int t = 0; t < WK; ++t) h[t] ^= e[t];
    }
}

// Secondary (count overflow) array: open addressing keyed by slot position.
__device__ inline void sec_add(const TableParams &p, uint64_t pos, uint64_t carry) {
    pos += p.pos_base;
    uint64_t slot = mix64(pos + 0x9E3779B97F4A7C15ULL) & p.sec_mask;
    atomicAdd(&p.stats[ST_CARRY], (unsigned long long)carry);
    for (int probe = 0; probe < 256; ++probe) {
        unsigned long long old = atomicCAS((unsigned long long *)&p.sec_keys[slot], 0ULL,
                                           (unsigned long long)(pos + 1));
        if (old == 0ULL || old == pos + 1) {
            atomicAdd((unsigned long long *)&p.sec_cnt[slot], (unsigned long long)carry);
            return;
        }
        slot = (slot + 1) & p.sec_mask;
    }
    atomicAdd(&p.stats[ST_SECFAIL], (unsigned long long)carry);
}
__device__ inline uint64_t sec_get(const TableParams &p, uint64_t pos) {
    uint64_t slot = mix64(pos + 0x9E3779B97F4A7C15ULL) & p.sec_mask;
    for (int probe = 0; probe < 256; ++probe) {
        uint64_t key = p.sec_keys[slot];
        if (key == pos + 1) return p.sec_cnt[slot];
        if (key == 0) return 0;
        slot = (slot + 1) & p.sec_mask;
    }
    return 0;
}

// Probe sequence: the reference's pos = (key + i(i+1)/2) mod 2^l
// (TSXHashMap.h:759-778,1046-1054) with the wrap-around taken inside the
// 2^S-slot segment of the home slot, so that one workgroup can own a segment
// (build_segments_stream_kernel) and still probe exactly like the atomic path.
// For l <= S this is the reference's formula.
__device__ __forceinline__ uint64_t probe_pos(const TableParams &p, uint64_t pos0, uint32_t i) {
    return (pos0 & ~p.seg_mask) | ((pos0 + (((uint64_t)i * (i + 1)) >> 1)) & p.seg_mask);
}

// Split a hashed key into what a slot stores: e0 = limb-0 key bits without the
// reprobe count, hi[] = the func bits that spill into limbs 1..W-1.
template <int WK>
__device__ __forceinline__ void split_key(const TableParams &p, const uint64_t (&h)[WK], uint64_t &pos0,
                                          uint64_t &e0, uint64_t (&hi)[4]) {
    pos0 = h[0] & p.slot_mask;
    // fr = (h >> l) << R, as WK+1 limbs
    uint64_t f[WK + 1];
#pragma unroll
    for (int t = 0; t < WK; ++t) {
        uint64_t v = h[t] >> p.lg;
        if (t + 1 < WK) v |= h[t + 1] << (64 - p.lg);
        f[t] = v;
    }
    f[WK] = 0;
    uint64_t fr[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = WK; t >= 0; --t) {
        uint64_t v = f[t] << p.R;
        if (t > 0) v |= f[t - 1] >> (64 - p.R);
        fr[t] = v;
    }
    e0 = fr[0] & p.k0mask;
    // hi = fr >> K0 (1 <= K0 <= 63)
#pragma unroll
    for (int t = 0; t < 4; ++t) hi[t] = (fr[t] >> p.K0) | (fr[t + 1] << (64 - p.K0));
}

// Which shard owns a hashed key: the bits of its home slot above the local l bits.
template <int WK>
__device__ __forceinline__ uint32_t owner_shard(const TableParams &p, const uint64_t (&h)[WK]) {
    return (uint32_t)((h[0] >> p.l) & ((1ULL << (p.lg - p.l)) - 1ULL));
}

// addKmer (TSXHashMap.h:182-350, CAS form TSXHashMapCAS.h:268-508) for one
// hashed key with d occurrences.  One 64-bit CAS claims an empty slot, one
// atomic add bumps an existing one; probing is the reference's
// pos = (key + i(i+1)/2) mod 2^l, i = 1,2,... (TSXHashMap.h:759-778,1046-1054).
template <int WK>
__device__ inline bool insert_key(const TableParams &p, const uint64_t (&h)[WK], uint64_t d) {
    uint64_t pos0, e0, hi[4];
    split_key<WK>(p, h, pos0, e0, hi);
    const uint64_t dlow = d << p.cshift;
    const int W = p.W;
    uint32_t i = 1;
    uint32_t spins = 0;
    // `state`: 0 = still probing, 1 = placed, 2 = out of reprobes.  The loop has a
    // single exit and no early return on purpose: a lane that claims a slot must
    // publish its limbs and drop the lock INSIDE the iteration, before the wave
    // branches back for the lanes that are still waiting on that very lock
    // (wave64 lanes share one program counter; an exit path would be run only
    // after every lane has left the loop).
    int state = 0;
    uint64_t carry = 0, carry_pos = 0;
    while (state == 0) {
        const uint64_t pos = probe_pos(p, pos0, i);
        unsigned long long *e = (unsigned long long *)(p.table + pos * (uint64_t)W);
        const uint64_t key0 = e0 | i;
        const unsigned long long old = atomicCAS(e, 0ULL, (unsigned long long)(key0 | p.lock_bit | dlow));
        bool next = false;
        if (old == 0ULL) {
            p.seg_dirty[pos >> p.S] = 1;  // idempotent plain store
            if (W > 1) {
                // Publish the remaining limbs and drop the lock with write-through
                // (agent-scope) stores instead of atomics: nobody else touches a locked
                // slot (adders wait for the lock to clear), so the unlocked value of limb 0
                // is known, and 3 atomics per new key become 1 atomic + W stores.  The
                // limb stores must have landed before the unlock store is issued.
                for (int t = 1; t < W; ++t)
                    __hip_atomic_store(e + t, (unsigned long long)hi[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(e, (unsigned long long)(key0 | dlow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            carry = d >> p.C; carry_pos = pos;
            state = 1;
        } else if ((old & p.k0mask) != key0) {
            next = true;
        } else if (W > 1 && (old & p.lock_bit)) {
            // claimed, limbs not published yet: look again on the next iteration
            if (++spins > (1u << 22)) { atomicAdd(&p.stats[ST_LOCKTO], 1ULL); spins = 0; next = true; }
        } else {
            bool same = true;
            for (int t = 1; t < W; ++t) same &= (atomicOr(e + t, 0ULL) == hi[t - 1]);
            if (same) {
                const unsigned long long prev = atomicAdd(e, (unsigned long long)dlow);
                carry = ((prev >> p.cshift) + d) >> p.C; carry_pos = pos;
                state = 1;
            } else {
                next = true;
            }
        }
        if (next) { ++i; if (i > p.max_reprobes) state = 2; }
    }
    if (state == 1) {
        if (carry) sec_add(p, carry_pos, carry);
        return true;
    }
    atomicAdd(&p.stats[ST_FAIL], (unsigned long long)d);
    return false;
}

// getKmerCount(kmer) (TSXHashMap.h:548-638).  Plain loads: runs in its own
// launch after every insert kernel has finished.
// pos_out (optional): the slot the k-mer was found in, ~0 when it is not in the table
// (KmerCountDebug::iFirstPos of getKmerCountDebug, TSXHashMap.h:477-545).
template <int WK>
__device__ inline uint64_t lookup_key(const TableParams &p, const uint64_t (&h)[WK], uint64_t *pos_out = nullptr) {
    if (pos_out) *pos_out = ~0ULL;
    if (p.lg != p.l && owner_shard<WK>(p, h) != p.shard) return 0;  // lives on another GPU
    uint64_t pos0, e0, hi[4];
    split_key<WK>(p, h, pos0, e0, hi);
    const int W = p.W;
    for (uint32_t i = 1; i <= p.max_reprobes; ++i) {
        const uint64_t pos = probe_pos(p, pos0, i);
        const uint64_t *e = p.table + pos * (uint64_t)W;
        const uint64_t v = e[0];
        if (v == 0) return 0;  // "why would the insertion skip an empty place?" TSXHashMap.h:622-626
        if ((v & p.k0mask) != (e0 | i)) continue;
        bool same = true;
        for (int t = 1; t < W; ++t) same &= (e[t] == hi[t - 1]);
        if (!same) continue;
        if (pos_out) *pos_out = pos;
        return (v >> p.cshift) + (sec_get(p, pos) << p.C);
    }
    return 0;
}

}  // namespace tsx
