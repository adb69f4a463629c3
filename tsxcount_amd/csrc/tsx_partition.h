// tsx_partition.h -- the partitioned insert path.
//
// Random 64-bit global atomics top out at ~17 G/s on MI355X whatever the
// footprint (profiles/round1_atomics_ubench.txt), so for large inputs the table
// is not updated in place.  Instead:
//   strip_desc_kernel + walk_part_kernel  strip descriptions, then the walk with every lane busy, fused with radix
//                          level 1 (one-limb keys); or + walk_log_kernel / walk_log_wide_kernel (tsx_kernels.h): hashed
//                          keys to one log region per wave + a level-1 histogram per region (no atomics)
//   offsets_*_kernel       exclusive scan of the histograms -> exact write offsets
//   partition_ring_kernel  x1 or x2: radix-scatters records by the high bits of their
//                          home slot into one list per table segment, through LDS
//                          ring staging and 128-B bursts
//   build_segments_stream_kernel  one workgroup per 2^S-slot segment: segment in LDS, inserts with LDS
//                          atomics from wave-level key streams, streamed back once
//                          (build_segments_wide_stream_kernel: multi-limb keys and slots)
//   overflow_insert_kernel, deferred_insert_kernel: what did not fit a list, inserted
//                          like the atomic path would -- AFTER the build
//   split_owner_kernel, hist_kernel, add_hashed_kernel: the sharded (multi-GPU) legs
// All HBM traffic is sequential.  A record is RW 64-bit words (RecWords<WK>): the WK limbs of
// the hashed key, padded to 1, 2 or 4 words.  Scan and partition never touch the table:
// whatever cannot take the fast route waits in queues (DeferList, tsx_device.h) until the
// build has written its segments, so a freshly cleared table needs no memset.
#pragma once
#include "tsx_kernels.h"
#include <type_traits>

namespace tsx {

constexpr int PART_NT = 256;
constexpr int RING_NT = 512;    // threads of a partition_ring_kernel workgroup: 16 waves per CU with two workgroups (256: 8 waves, level 2 3.5 ms; 1024: 4.1 ms)
constexpr int PART_WPT = 4;     // words per thread per batch (4 / RW records)
constexpr int PART_FLUSH = 16;   // words per burst (128 B = one L2 line)

template <int RW>
__device__ __forceinline__ void load_rec(const uint64_t *ptr, uint64_t (&r)[RW]) {
    if constexpr (RW == 1) {
        r[0] = *ptr;
    } else {
#pragma unroll
        for (int t = 0; t < RW; t += 2) {
            const uint4 v = *reinterpret_cast<const uint4 *>(ptr + t);
            r[t] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            r[t + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
    }
}
template <int RW>
__device__ __forceinline__ void store_rec(uint64_t *ptr, const uint64_t (&r)[RW]) {
    if constexpr (RW == 1) {
        *ptr = r[0];
    } else {
#pragma unroll
        for (int t = 0; t < RW; t += 2)
            *reinterpret_cast<uint4 *>(ptr + t) = make_uint4((uint32_t)r[t], (uint32_t)(r[t] >> 32), (uint32_t)r[t + 1],
                                                             (uint32_t)(r[t + 1] >> 32));
    }
}

// Atomic-free radix level.  A workgroup (r, c) takes every cpr-th batch of source
// region r and owns the write cursors of its destination lists, which therefore
// live in LDS (returning atomics on shared cursors cost 12 of 17 ms in the first
// version).  Two uses:
//   level 1  region = one scan wave's key log, cpr = 1; cursor[b] starts at the
//            exact offset offs[b * nregions + r] that offsets_kernel derived from the
//            scan kernel's histograms: the output is a packed array ordered by bucket
//   level 2  region = one level-1 bucket (start/count from offsets_kernel); the records
//            of segment (r * nb + b) go to sub-list ((r * nb + b) * cpr + c) with
//            room for dst_cap records; its final size is published to dst_cnt
// Sizes and offsets at the interface count RECORDS; inside, rings, cursors and bursts count
// 64-bit WORDS (a record is RW of them, RW | 16), so the staging logic is the same for every
// key width.  Staging is a ring of 2^capbits words per list (no compaction after a flush).
// A flush is (A) one thread per list: how many words (multiple of PART_FLUSH = one
// 128-B line; 64-B bursts cost 0.8 ms more in level 1) and where; (B) 8 consecutive
// lanes per list copy them, with the LDS reads of all lists an octet serves issued
// before the first store.
// What does not fit a destination list (level 2 sub-lists under skew): almost always a handful of hot
// keys (k-mers that occur once per read, thousands of times in all).  They are folded into a small LDS
// cache and join the deferred list with their totals when the workgroup ends, instead of one queue entry
// per occurrence; ordinary records displaced by them go to the workgroup's overflow queue.
constexpr uint32_t OVF_N = 64;
constexpr uint64_t OVF_SALT = 0x5DEECE66D1CE4E5BULL;

// The record level 2 leaves for build_segments_stream_kernel<.., PRE = true> (one-limb keys and slots, R + F >= 32,
// R + F + S <= 64): the slot image of the key -- func bits above R zero bits where the reprobe count goes -- and, above
// it, the FIRST probe position q_1 = (home + 1) mod 2^S inside the segment.  The build then needs a mask and a shift per
// key instead of two 64-bit shifts, three ands and an add, in the loop that is bound by instruction issue.
__device__ __forceinline__ uint64_t format_record(const TableParams &p, uint64_t key) {
    const uint32_t q1 = ((uint32_t)key + 1u) & (uint32_t)p.seg_mask;
    return ((key >> p.lg) << p.R) | ((uint64_t)q1 << (p.R + p.F));
}

constexpr int PART_MAX_PIECES = 512;   // pieces of a source region one workgroup may have to walk (src_np / cpr)
constexpr int PART_ITER = 3;  // flush jobs an octet serves per pass (128 jobs per pass: a usual round of 256 lists)
// ---- records that find their list full (partition_ring_kernel) --------------------------------------------------------
// What does not fit a destination list: almost always hot keys.  Three stations, all of one workgroup:
//   cache   OVF_N (key -> count) places in LDS, four per key; a key is ADMITTED only when it comes as a combined record
//           (d >= 2: the wave has just seen it more than once) -- once a hot key's list is full the cold keys of its
//           segment spill as well, one each, and would take every place before the hot keys get there;
//   queue   the workgroup's overflow queue in HBM (ovq_cap single records), inserted after the build;
//   chunks  the deferred list, in chunks of dch records that belong to this workgroup alone: one global atomic per
//           chunk, an LDS counter per record (skewed input -- BASELINE config 4 -- spills 4.6e8 records; one atomic each
//           on the list's single counter took 1.8 s).  deferred_insert_kernel finds a bucket's records side by side.
// Only the SKEW form of the kernel holds this code (see there).  The state is one LDS struct of the kernel.
template <int RW> struct SpillState {
    static constexpr uint32_t DCH_MAX = (RW == 4) ? 48 : 96;   // (LDS: two workgroups of 80 KB share a CU)
    uint64_t ovk[OVF_N * RW];               // cached keys (word 0 xor OVF_SALT, 0 = free)
    unsigned long long dch[DCH_MAX];        // first record of chunk ci, + 1 (0: not allocated yet)
    uint32_t ovc[OVF_N];                    // their counts
    uint32_t ovr[OVF_N];                    // RW > 1: the other words of the cached key have been written
    uint32_t ovn;                           // records in the overflow queue
    uint32_t dn;                            // records deferred in chunks
    uint32_t skew;                          // a record of this workgroup has found its list full
};
#define TSX_LDS __attribute__((address_space(3)))
template <int RW>
__device__ __forceinline__ void spill_record(const TableParams *pk, TSX_LDS SpillState<RW> *sp, uint64_t *ovq, uint32_t ovq_cap,
                                             uint32_t dch, RecVal<RW> r, uint32_t d) {
    sp->skew = 1u;
    const uint64_t kk = r.w[0] ^ OVF_SALT;
    if (kk != 0) {
        uint64_t mixin = r.w[0];
#pragma unroll
        for (int t = 1; t < RW; ++t) mixin ^= r.w[t] * 0x9E3779B97F4A7C15ULL;
        uint32_t slot = (uint32_t)(mix64(mixin) >> 40) & (OVF_N - 1);
        for (int pr = 0; pr < 4; ++pr, slot = (slot + 1) & (OVF_N - 1)) {
            TSX_LDS unsigned long long *w0 = (TSX_LDS unsigned long long *)&sp->ovk[slot * RW];
            unsigned long long old = 0ULL;
            if (d >= 2u) __hip_atomic_compare_exchange_strong(w0, &old, (unsigned long long)kk, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_WORKGROUP);
            else old = __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 0ULL) {
                if (d < 2u) break;   // not cached, and not to be
                if constexpr (RW > 1) {   // claimed: publish the other words, then the ready flag
#pragma unroll
                    for (int t = 1; t < RW; ++t) sp->ovk[slot * RW + t] = r.w[t];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __hip_atomic_store(&sp->ovr[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __hip_atomic_fetch_add(&sp->ovc[slot], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return;
            }
            if (old == kk) {
                bool same = true;
                if constexpr (RW > 1) {
                    same = __hip_atomic_load(&sp->ovr[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
                    if (same) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                        for (int t = 1; t < RW; ++t) same &= (sp->ovk[slot * RW + t] == r.w[t]);
                    }
                }
                if (same) { __hip_atomic_fetch_add(&sp->ovc[slot], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
            }
        }
    }
    if (ovq && d == 1u) {
        const uint32_t at = __hip_atomic_fetch_add(&sp->ovn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (at < ovq_cap) { store_rec<RW>(ovq + (size_t)at * RW, r.w); return; }
    }
    if (dch == 0) { defer_append_v<RW>(pk, r, d); return; }
    const uint32_t at = __hip_atomic_fetch_add(&sp->dn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t ci = at / dch, off = at % dch;
    if (ci >= SpillState<RW>::DCH_MAX) { defer_append_v<RW>(pk, r, d); return; }
    if (off == 0) {   // the lane that opens a chunk takes it from the list (its store comes BEFORE the wait below,
                      // also for the other lanes of this wave: no lane waits for a lane that waits)
        const unsigned long long first = atomicAdd(pk->defer.n, (unsigned long long)dch);
        __hip_atomic_store(&sp->dch[ci], first + 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned long long first;
    do { first = __hip_atomic_load(&sp->dch[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (first == 0ULL);
    first -= 1ULL;
    if (first + dch <= pk->defer.cap) {
        store_rec<RW>(pk->defer.rec + (first + off) * RW, r.w);
        pk->defer.cnt[first + off] = (unsigned long long)d;
    } else {
        atomicAdd(&pk->stats[ST_FAIL], (unsigned long long)d);
    }
}

// Which buckets are skewed enough for the SKEW form of level 2?  One workgroup per level-1 bucket looks at SAMPLE records
// of it (runs of 16 at a pseudo-random place of every piece, or one run of a packed bucket), counts them in an LDS table
// by a 64-bit tag of the record and sets flag[bucket] when MANY different records show up HOT times or more: more hot keys
// than the plain form's 64-place cache absorbs.  (A few hot keys per bucket are normal and cost the plain form nothing:
// the reads of the bench end in poly-A tails, and the 4^j k-mers with j = 1..5 bases in front of 26+ A's occur 10^3-10^5
// times each -- one or two of them per bucket.)  A miss costs time -- the plain form is correct for any input --, never a count.
template <int RW>
__global__ __launch_bounds__(256) void skew_probe_kernel(const uint64_t *src, const unsigned long long *src_start,
                                                         const unsigned long long *src_cnt, const unsigned long long *src_pcnt,
                                                         uint32_t src_np, uint64_t src_pcap, uint32_t *flag) {
    constexpr uint32_t TN = 8192, SAMPLE = 4096, HOT = 4, MANY = 8;
    __shared__ uint64_t s_tag[TN];
    __shared__ uint32_t s_num[TN];
    __shared__ uint32_t s_seen, s_mass;
    const uint32_t tid = threadIdx.x, r = blockIdx.x;
    for (uint32_t t = tid; t < TN; t += 256) { s_tag[t] = 0; s_num[t] = 0; }
    if (tid == 0) { s_seen = 0; s_mass = 0; }
    __syncthreads();
    uint32_t seen = 0;
    for (uint32_t i = tid; i < SAMPLE; i += 256) {
        uint64_t at = ~0ULL;   // record index in src
        if (src_pcnt) {        // run i / 16 of the sample comes from piece (i / 16) * src_np / (SAMPLE / 16)
            const uint32_t run = i >> 4, g = (uint32_t)(((uint64_t)run * src_np) / (SAMPLE / 16));
            const uint64_t pa = (uint64_t)r * src_np + g, cnt = min((uint64_t)src_pcnt[pa], src_pcap);
            if (cnt >= 16) at = pa * src_pcap + (mix64(pa + 0x51ED) % (cnt - 15)) + (i & 15u);
        } else {
            const uint64_t cnt = src_cnt[r];
            if (cnt > i) at = (uint64_t)src_start[r] + (cnt > SAMPLE ? (mix64(r + 0x51ED) % (cnt - SAMPLE + 1)) : 0) + i;
        }
        if (at == ~0ULL) continue;
        ++seen;
        uint64_t tag = 0;
#pragma unroll
        for (int t = 0; t < RW; ++t) tag ^= mix64(src[at * RW + t] + (uint64_t)t);
        tag |= 1ULL;
        uint32_t slot = (uint32_t)(tag >> 40) & (TN - 1);
        for (int pr = 0; pr < 8; ++pr, slot = (slot + 1) & (TN - 1)) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&s_tag[slot]), 0ULL, (unsigned long long)tag);
            if (old == 0ULL || old == tag) { atomicAdd(&s_num[slot], 1u); break; }
        }
    }
    atomicAdd(&s_seen, seen);
    __syncthreads();
    uint32_t hot = 0;
    for (uint32_t t = tid; t < TN; t += 256) if (s_num[t] >= HOT) ++hot;
    if (hot) atomicAdd(&s_mass, hot);
    __syncthreads();
    if (tid == 0) flag[r] = (s_mass >= MANY && s_seen >= 256u) ? 1u : 0u;
}

// NT threads: 512 where two workgroups share a CU (<= 80 KB of rings: up to 256 lists); 1024 where the rings of 512 lists
// leave room for one workgroup only -- 16 waves per CU either way (8 waves: level 2 of a 2^18-segment table 7.5 ms per
// 1e9 keys instead of 3.1).
// SKEW: the kernel exists in two forms that are BOTH launched for level 2; *skew_flag (set by skew_probe_kernel from a
// sample of every bucket) says which of them works, the other returns at once.  The plain form is round 2's kernel: what
// does not fit a list goes through a one-place cache, the overflow queue and the deferred list's single counter -- fine
// for the odd hot key, and nothing of the skew machinery costs it an instruction (all of it inlined into one kernel made
// level 2 of uniform input 0.5 ms slower: code size; as a function call: 0.9 ms, the calls constrain the hot loop's
// registers).  The skew form: spill_record, combined records, lists looked at before their rings.
template <int RW, int NT = RING_NT, bool SKEW = false>
__global__ __launch_bounds__(NT) void partition_ring_kernel(
    TableParams p, const uint64_t *src, const unsigned long long *src_start, const unsigned long long *src_cnt,
    uint64_t src_cap, uint32_t nregions, uint32_t cpr, uint64_t *dst, const unsigned long long *offs,
    const unsigned long long *offs_base, unsigned long long *dst_cnt, uint64_t dst_cap, uint32_t nb, uint32_t shift,
    uint32_t capbits, int dbg, uint64_t *ovq_all, uint32_t *ovq_cnt, uint32_t ovq_cap,
    const unsigned long long *src_pcnt, uint32_t src_np, uint64_t src_pcap, int dst_bm, unsigned long long *key_sum,
    uint32_t dst_r0, uint32_t dst_nr, int fmt, uint32_t dch, const uint32_t *skew_flag) {
    constexpr int RPT = (PART_WPT >= RW) ? PART_WPT / RW : 1;   // records per thread per batch
    extern __shared__ uint64_t s_part[];  // rings | cursors | limits | flush descriptors | tails | heads | jobs
    const uint32_t CAP = 1u << capbits, cmask = CAP - 1;
    uint64_t *s_stage = s_part;
    unsigned long long *s_cur = reinterpret_cast<unsigned long long *>(s_part + ((size_t)nb << capbits));
    unsigned long long *s_lim = s_cur + nb;
    unsigned long long *s_meta = s_lim + nb;
    uint32_t *s_tail = reinterpret_cast<uint32_t *>(s_meta + nb);
    uint32_t *s_head = s_tail + nb;
    uint32_t *s_job = s_head + nb;      // lists with something to flush this round, in arrival order
    __shared__ uint32_t s_njobs[2];     // their number; two counters used alternately (reset one round ahead)
    __shared__ SpillState<RW> s_sp;     // records that find their list full: see spill_record
    const uint32_t tid = threadIdx.x;
    const uint32_t r = blockIdx.x / cpr, c = blockIdx.x % cpr;
    if (r >= nregions) return;
    if (skew_flag && (skew_flag[r] != 0u) != SKEW) return;   // the other form of this kernel works on this bucket
    if (tid < OVF_N) { s_sp.ovk[tid * RW] = 0; s_sp.ovc[tid] = 0; s_sp.ovr[tid] = 0; }
    if (tid < 2) s_njobs[tid] = 0;
    if (tid == 0) { s_sp.ovn = 0; s_sp.dn = 0; s_sp.skew = 0; }
    if (tid < SpillState<RW>::DCH_MAX) s_sp.dch[tid] = 0;
    uint64_t *ovq = ovq_all ? ovq_all + (size_t)blockIdx.x * ovq_cap * RW : nullptr;
    for (uint32_t b = tid; b < nb; b += NT) {
        s_tail[b] = 0; s_head[b] = 0;
        if (offs) { s_cur[b] = (offs_base[b] + offs[(size_t)b * nregions + r]) * RW; s_lim[b] = ~0ULL; }
        else {   // own sub-list of the destination list; dst_bm: numbered bucket-major (the lists of one bucket side by
                 // side: the next level reads a bucket as nregions * cpr pieces)
            // (dst_r0, dst_nr: this launch partitions regions dst_r0 .. of dst_nr in all -- one launch per exchange window)
            const uint64_t li = dst_bm ? ((uint64_t)b * dst_nr + dst_r0 + r) * cpr + c : ((uint64_t)r * nb + b) * cpr + c;
            s_cur[b] = li * dst_cap * RW; s_lim[b] = s_cur[b] + dst_cap * RW;
        }
    }
    lds_barrier();
    // The source region r: a contiguous run of records (src_start/src_cnt, or src_cap apart), or -- src_pcnt
    // given -- src_np PIECES of at most src_pcap records each, piece g at ((r * src_np + g) * src_pcap) with
    // src_pcnt[r * src_np + g] records (the sub-lists walk_part_kernel's workgroups keep per level-1 bucket);
    // workgroup c of the region then takes pieces c, c + cpr, ... as one stream.
    const uint64_t n = src_pcnt ? 0 : (src_start ? (uint64_t)src_cnt[r] : min((uint64_t)src_cnt[r], src_cap));   // records
    const uint64_t region_first = src_pcnt ? 0 : (src_start ? (uint64_t)src_start[r] : (uint64_t)r * src_cap);
    constexpr uint32_t BATCH_REC = (uint32_t)NT * RPT;
    const uint64_t stride = (uint64_t)cpr * BATCH_REC;

    // TableParams is the first kernel argument: the slow paths read it from the argument segment
    const TableParams *pk = (const TableParams *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t spilled = 0;  // per thread; one atomic per wave at the end
    // A record that found its list full.  In line: one probe of the spill cache (a cached hot key hits it)
    // and the overflow queue (an ordinary record misses); the deferred list takes what the queue cannot.
    // (d: the occurrences the record stands for: 1, or the equal records a wave sends away together)
    auto spill = [&](const uint64_t (&rec)[RW], uint32_t d = 1u) {
        spilled += d;
        if constexpr (SKEW) {
            RecVal<RW> v;
#pragma unroll
            for (int t = 0; t < RW; ++t) v.w[t] = rec[t];
            spill_record<RW>(pk, (TSX_LDS SpillState<RW> *)&s_sp, ovq, ovq_cap, dch, v, d);
        } else {   // one probe of the cache (a cached hot key hits it), the overflow queue, the deferred list
            const uint64_t kk = rec[0] ^ OVF_SALT;
            if (kk != 0) {
                uint64_t mixin = rec[0];
#pragma unroll
                for (int t = 1; t < RW; ++t) mixin ^= rec[t];
                const uint32_t slot = (uint32_t)(mix64(mixin) >> 40) & (OVF_N - 1);
                const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&s_sp.ovk[slot * RW]), 0ULL,
                                                         (unsigned long long)kk);
                if constexpr (RW == 1) {
                    if (old == 0ULL || old == kk) { atomicAdd(&s_sp.ovc[slot], 1u); return; }
                } else {
                    if (old == 0ULL) {   // claimed: publish the other words, then the ready flag
#pragma unroll
                        for (int t = 1; t < RW; ++t) s_sp.ovk[slot * RW + t] = rec[t];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __hip_atomic_store(&s_sp.ovr[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        atomicAdd(&s_sp.ovc[slot], 1u);
                        return;
                    }
                    if (old == kk && __hip_atomic_load(&s_sp.ovr[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        bool same = true;
#pragma unroll
                        for (int t = 1; t < RW; ++t) same &= (s_sp.ovk[slot * RW + t] == rec[t]);
                        if (same) { atomicAdd(&s_sp.ovc[slot], 1u); return; }
                    }
                }
            }
            if (ovq) {
                const uint32_t at = atomicAdd(&s_sp.ovn, 1u);
                if (at < ovq_cap) { store_rec<RW>(ovq + (size_t)at * RW, rec); return; }
            }
            defer_append<RW>(pk, rec, 1);
        }
    };
    // one word of a burst: word q of the burst that starts at ring index hd of `ring` and at destination `at`
    auto put_word = [&](uint64_t w, const uint64_t *ring, uint32_t hd, uint32_t q, unsigned long long at,
                        unsigned long long lim) {
        if (dbg & 256) return;  // ablation: no stores
        if (at + q < lim) { dst[at + q] = (RW == 1 && fmt) ? format_record(p, w) : w; return; }   // (fmt: see format_record)
        if ((q & (RW - 1)) == 0) {   // sub-list full: the lane that holds the record's first word spills all of it
            uint64_t rec[RW];
#pragma unroll
            for (int t = 0; t < RW; ++t) rec[t] = ring[(hd + q + t) & cmask];
            spill(rec);
        }
    };
    uint32_t round = 0;   // flush rounds so far (workgroup-uniform)
    // `between` runs after phase (A) and its barrier, before the first store of phase (B): the main loop waits for
    // the next batch's loads THERE.  vmcnt counts loads and stores in one in-order queue: waiting for those loads
    // behind the flush (as the first version did) also waited for the flush's own stores -- every batch paid a full
    // store latency, half of the kernel's time.
    auto flush = [&](bool all, auto &&between) {
        const uint32_t par = round & 1u;
        ++round;
        if (tid == 0) s_njobs[par ^ 1u] = 0;   // next round's counter: nobody reads or writes it in this round
        for (uint32_t b = tid; b < nb; b += NT) {  // (A)
            const uint32_t head = s_head[b];
            const uint32_t tail = min(s_tail[b], head + CAP);  // arrivals past the ring went out directly
            const uint32_t avail = tail - head;
            const unsigned long long at = s_cur[b];
            // bursts end on a 128-B line of the destination: after a list's first (short) burst every
            // store instruction of an octet writes one whole, aligned line
            const unsigned long long end = (at + avail) & ~(unsigned long long)(PART_FLUSH - 1);
            const uint32_t nout = all ? avail : (end > at ? (uint32_t)(end - at) : 0u);
            s_meta[b] = (at << 16) | ((unsigned long long)(head & cmask) << 8) | nout;
            s_head[b] = head + nout;
            s_tail[b] = tail;
            s_cur[b] = at + nout;
            if (nout) s_job[atomicAdd(&s_njobs[par], 1u)] = b;   // about half of the lists in a usual round
        }
        lds_barrier();
        between();
        const uint32_t oct = tid >> 3, ol = tid & 7;  // (B): an octet of lanes per job
        if (dbg & 512) return;  // ablation: bookkeeping only
        const uint32_t njobs = s_njobs[par];
        for (uint32_t j0 = 0; j0 < njobs; j0 += PART_ITER * (NT / 8)) {
            unsigned long long meta[PART_ITER], lim[PART_ITER];
            uint32_t bj[PART_ITER];
            uint64_t k0[PART_ITER], k1[PART_ITER];
#pragma unroll
            for (int u = 0; u < PART_ITER; ++u) {
                const uint32_t j = j0 + oct + u * (NT / 8);
                bj[u] = (j < njobs) ? s_job[j] : 0u;
            }
#pragma unroll
            for (int u = 0; u < PART_ITER; ++u) {
                const uint32_t j = j0 + oct + u * (NT / 8);
                meta[u] = (j < njobs) ? s_meta[bj[u]] : 0ULL;
                lim[u] = (j < njobs) ? s_lim[bj[u]] : 0ULL;
            }
#pragma unroll
            for (int u = 0; u < PART_ITER; ++u) {
                const uint32_t nout = (uint32_t)(meta[u] & 0xFF), hd = (uint32_t)(meta[u] >> 8) & 0xFF;
                const uint64_t *ring = s_stage + ((size_t)bj[u] << capbits);
                k0[u] = (ol < nout) ? ring[(hd + ol) & cmask] : 0;
                k1[u] = (ol + 8 < nout) ? ring[(hd + ol + 8) & cmask] : 0;
            }
#pragma unroll
            for (int u = 0; u < PART_ITER; ++u) {
                const uint32_t nout = (uint32_t)(meta[u] & 0xFF), hd = (uint32_t)(meta[u] >> 8) & 0xFF;
                const unsigned long long at = meta[u] >> 16;
                const uint64_t *ring = s_stage + ((size_t)bj[u] << capbits);
                if (ol < nout) put_word(k0[u], ring, hd, ol, at, lim[u]);
                if (ol + 8 < nout) put_word(k1[u], ring, hd, ol + 8, at, lim[u]);
                if (nout > 16)   // only rings deeper than 16 or the final flush get here
                    for (uint32_t q = ol + 16; q < nout; q += 8) put_word(ring[(hd + q) & cmask], ring, hd, q, at, lim[u]);
            }
        }
    };

    // A batch: record j (j = q * NT + tid) comes from record index first + j, or first + j + delta once
    // j >= rem (a batch may run from the end of one piece into the next one); nvalid records in all.
    struct Batch { uint64_t first, delta; uint32_t rem, nvalid; };
    uint64_t base = (uint64_t)c * BATCH_REC;   // contiguous source: next batch
    uint32_t poff = 0;                         // pieces: records consumed of the current piece
    // the sizes of this workgroup's pieces (c, c + cpr, ...) wait in LDS: the stream logic below is on the
    // critical path of every batch and must not go to memory
    __shared__ uint32_t s_pc[PART_MAX_PIECES];
    if (src_pcnt) {
        for (uint32_t i = tid; c + i * cpr < src_np && i < (uint32_t)PART_MAX_PIECES; i += NT)
            s_pc[i] = (uint32_t)min((uint64_t)src_pcnt[(uint64_t)r * src_np + c + i * cpr], src_pcap);
        lds_barrier();
    }
    const uint32_t npw = (src_pcnt && c < src_np) ? min((src_np - c + cpr - 1) / cpr, (uint32_t)PART_MAX_PIECES) : 0u;
    uint32_t pi = 0;   // ordinal of the current piece: piece c + pi * cpr of the region
    auto pcnt = [&](uint32_t i) -> uint32_t { return i < npw ? s_pc[i] : 0u; };
    auto next_batch = [&]() -> Batch {   // workgroup-uniform; describes the next batch and moves on
        Batch d;
        d.delta = 0;
        if (!src_pcnt) {
            d.first = region_first + base;
            d.rem = BATCH_REC;
            d.nvalid = base < n ? (uint32_t)min((uint64_t)BATCH_REC, n - base) : 0u;
            base += stride;
            return d;
        }
        while (pi < npw) {   // pieces that are used up (or empty)
            const uint32_t cg = pcnt(pi);
            if (poff < cg) break;
            poff -= cg;
            ++pi;
        }
        if (pi >= npw) { d.first = 0; d.rem = 0; d.nvalid = 0; return d; }
        const uint32_t left = pcnt(pi) - poff, cb = pcnt(pi + 1);
        const uint64_t pa = (uint64_t)r * src_np + c + (uint64_t)pi * cpr;   // number of the piece among all
        d.first = pa * src_pcap + poff;
        d.rem = min(left, BATCH_REC);
        d.delta = (pa + cpr) * src_pcap - (d.first + left);
        d.nvalid = (uint32_t)min((uint64_t)BATCH_REC, (uint64_t)left + cb);
        if (left > BATCH_REC) poff += BATCH_REC;
        else { ++pi; poff = d.nvalid - left; }
        return d;
    };
    auto load_batch = [&](const Batch &d, uint64_t (&regs)[RPT][RW]) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t j = (uint32_t)q * NT + tid;
            if (j < d.nvalid) load_rec<RW>(src + (d.first + j + (j >= d.rem ? d.delta : 0ULL)) * RW, regs[q]);
            else {
#pragma unroll
                for (int t = 0; t < RW; ++t) regs[q][t] = 0;
            }
        }
    };
    uint64_t cur[RPT][RW], nxt[RPT][RW];
    unsigned long long ksum = 0;   // sum of the keys read (sharded runs: checked against the sum the scans wrote)
    Batch bc = next_batch();
    load_batch(bc, cur);
    // The first batch must have ARRIVED before the loop: the compiler then knows that `cur` is complete at the
    // loop header on both edges and waits for the next batch's loads where they are consumed (the copies at the
    // end of the batch) -- not right behind their issue, which is what it did without this wait: vmcnt(0) in
    // front of the first use of `cur`, i.e. one full HBM latency per batch and no prefetch at all.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    while (bc.nvalid) {
        const Batch bn = next_batch();
        load_batch(bn, nxt);
        // all ring places of the batch are taken before any is used: the returning LDS atomics
        // of a thread are in flight together.
        // Two forms of the append.  The plain one is what uniform input runs.  Once a record of this workgroup has found
        // its list full (s_skew: skewed input, BASELINE config 4) the batches take the second form: a record first looks
        // whether ITS list is full and then skips the ring altogether (two same-address LDS atomics and a burst that
        // would spill anyway), and the wave sends equal homeless records away as ONE entry with their number.
        auto append = [&](auto skew_tag) {
            constexpr bool HOT = decltype(skew_tag)::value;   // lists of this workgroup have run full
            uint32_t bq[RPT], slot[RPT], head[RPT];
            bool pre_full[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t j = (uint32_t)q * NT + tid;
                if (key_sum && j < bc.nvalid) ksum += cur[q][0];
                bq[q] = (uint32_t)(cur[q][0] >> shift) & (nb - 1);
                pre_full[q] = HOT && j < bc.nvalid && s_cur[bq[q]] >= s_lim[bq[q]];
                slot[q] = (j < bc.nvalid && !pre_full[q]) ? atomicAdd(&s_tail[bq[q]], (uint32_t)RW) : 0u;
            }
#pragma unroll
            for (int q = 0; q < RPT; ++q) head[q] = s_head[bq[q]];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t j = (uint32_t)q * NT + tid;
                bool homeless = pre_full[q];   // ring full AND list full, or the list seen full up front
                if (j < bc.nvalid && !pre_full[q]) {
                    const uint32_t b = bq[q];
                    if (slot[q] - head[q] < CAP) {   // RW | CAP and records are RW-aligned: a record never wraps
                        uint64_t *ring = s_stage + ((size_t)b << capbits) + (slot[q] & cmask);
#pragma unroll
                        for (int t = 0; t < RW; ++t) ring[t] = cur[q][t];
                    } else if (!(dbg & 256)) {  // ring full: take the next place of the list directly
                        const unsigned long long at = atomicAdd(&s_cur[b], (unsigned long long)RW);
                        if (at < s_lim[b]) {
                            if (RW == 1 && fmt) dst[at] = format_record(p, cur[q][0]);
                            else store_rec<RW>(dst + at, cur[q]);
                        } else if (HOT) homeless = true;
                        else spill(cur[q]);
                    }
                }
                if constexpr (HOT) {
                    // two rounds of: the first homeless lane's record -- who has the same?  What is left goes one by one.
                    unsigned long long todo = __ballot(homeless);
                    for (int rnd = 0; rnd < 2 && todo != 0ULL; ++rnd) {
                        const int L = __builtin_ctzll(todo);
                        bool same = homeless;
#pragma unroll
                        for (int t = 0; t < RW; ++t) {
                            const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)cur[q][t], L),
                                           hi = __builtin_amdgcn_readlane((uint32_t)(cur[q][t] >> 32), L);
                            same &= (cur[q][t] == (((uint64_t)hi << 32) | lo));
                        }
                        const unsigned long long mm = __ballot(same);
                        if ((int)(tid & 63u) == L) spill(cur[q], (uint32_t)__builtin_popcountll(mm));
                        if (same) homeless = false;
                        todo &= ~mm;
                    }
                    if (homeless) spill(cur[q]);
                }
            }
        };
        if constexpr (SKEW) {
            if (s_sp.skew != 0u) append(std::true_type{});
            else append(std::false_type{});
        } else {
            append(std::false_type{});
        }
        lds_barrier();
        flush(false, [&]() {
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), spelled out: the compiler would sink the copies (and the wait) behind the stores
#pragma unroll
            for (int q = 0; q < RPT; ++q)
#pragma unroll
                for (int t = 0; t < RW; ++t) cur[q][t] = nxt[q][t];
        });
        lds_barrier();
        bc = bn;
    }
    flush(true, []() {});
    lds_barrier();
    {   // the unused tail of this workgroup's last chunk: count 0 = "no record" to deferred_insert_kernel
        const uint32_t dn = SKEW ? s_sp.dn : 0u;
        if (dch && dn % dch != 0u && (dn - 1u) / dch < SpillState<RW>::DCH_MAX) {
            const unsigned long long first = s_sp.dch[(dn - 1u) / dch] - 1ULL;
            if (first + dch <= pk->defer.cap)
                for (uint32_t o = dn % dch + tid; o < dch; o += NT) pk->defer.cnt[first + o] = 0ULL;
        }
    }
    if (dst_cnt)
        for (uint32_t b = tid; b < nb; b += NT) {
            const uint64_t li = dst_bm ? ((uint64_t)b * dst_nr + dst_r0 + r) * cpr + c : ((uint64_t)r * nb + b) * cpr + c;
            dst_cnt[li] = (min(s_cur[b], s_lim[b]) - li * dst_cap * RW) / RW;
        }
    if (tid < OVF_N && s_sp.ovc[tid] && !(dbg & 1)) {   // hot keys: one deferred entry each, with the total
        uint64_t rec[RW];
#pragma unroll
        for (int t = 0; t < RW; ++t) rec[t] = s_sp.ovk[tid * RW + t];
        rec[0] ^= OVF_SALT;
        defer_append<RW>(pk, rec, s_sp.ovc[tid]);
    }
    if (tid == 0 && ovq_cnt) ovq_cnt[blockIdx.x] = min(s_sp.ovn, ovq_cap);
    for (int d = 32; d > 0; d >>= 1) spilled += __shfl_down(spilled, d, 64);
    if ((tid & 63) == 0 && spilled) atomicAdd(&p.stats[ST_FALLBACK], (unsigned long long)spilled);
    if (key_sum) {
        for (int d = 32; d > 0; d >>= 1) ksum += __shfl_down(ksum, d, 64);
        if ((tid & 63) == 0 && ksum) atomicAdd(key_sum, ksum);
    }
}

// ---- the walk fused with radix level 1 (one-limb keys, two-level split): walk_part_kernel below --------------------
// A key log that level 1 reads again costs 12.9 GB and a 3 ms launch per 1e9 k-mers for nothing but a histogram.
// walk_part_kernel keeps the level-1 staging rings next to the walk: the keys of a strip go straight into the ring of
// their level-1 bucket, bursts of 128 B leave for the workgroup's own sub-list of that bucket (fixed capacity dst_cap,
// list (b, g) at ((b * G + g) * dst_cap), its size in dst_cnt[b * G + g]; what does not fit: spill cache / overflow
// queue / deferred list, as in level 2).  Level 2 reads a bucket as the G pieces the workgroups left
// (partition_ring_kernel, src_pcnt).  512 threads, two workgroups per CU: the rings take 64 KiB, the LUT of the first
// window is the 4-bit-group one (2 KiB, 16 lookups per strip instead of 8) so that two workgroups fit.
constexpr int SP_NT = 512;
constexpr uint32_t SP_CAPBITS = 5;   // 32 words per ring: 15 may stay behind a flush, ~7 arrive per half strip

// Descriptions of one text window, packed for the exchange of a sharded run: strip_desc_kernel leaves one region per
// wave; desc_prefix_kernel (one workgroup) turns the region sizes into offsets, desc_pack_kernel copies the regions
// back to back.  total[0] = descriptions in all.
__global__ __launch_bounds__(1024) void desc_prefix_kernel(const unsigned long long *cnt, uint32_t nregions,
                                                           unsigned long long *offs, unsigned long long *total,
                                                           uint64_t desc_cap, uint64_t out_cap, unsigned long long *stats) {
    // (a region that ran past its capacity, or more descriptions than the caller's array holds: the sizes are bounded by
    // construction -- one description per 16 / 64 start positions -- so either is a sizing bug; it is clamped here,
    // consistently with desc_pack_kernel, and reported through the sticky failure counter instead of a short exchange)
    __shared__ unsigned long long s_w[16];
    const uint32_t tid = threadIdx.x, per = (nregions + 1023u) / 1024u;
    unsigned long long sum = 0, lost = 0;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t r = tid * per + i;
        if (r < nregions) { sum += min((unsigned long long)cnt[r], (unsigned long long)desc_cap); lost += cnt[r] > desc_cap ? cnt[r] - desc_cap : 0ULL; }
    }
    const unsigned long long inc = wave_incl_scan64(sum);
    if ((tid & 63) == 63) s_w[tid >> 6] = inc;
    __syncthreads();
    unsigned long long base = inc - sum;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += s_w[w];
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t r = tid * per + i;
        if (r < nregions) { offs[r] = base; base += min((unsigned long long)cnt[r], (unsigned long long)desc_cap); }
    }
    if (lost && stats) atomicAdd(&stats[ST_FAIL], lost);
    if (tid == 1023) {
        if (base > out_cap) { if (stats) atomicAdd(&stats[ST_FAIL], base - out_cap); base = out_cap; }
        total[0] = base;
    }
}
__global__ __launch_bounds__(256) void desc_pack_kernel(const uint4 *desc, uint64_t desc_cap, const unsigned long long *cnt,
                                                        const unsigned long long *offs, uint32_t nregions, uint4 *out,
                                                        uint64_t out_cap, uint32_t du) {   // du: 16-byte units per description
    for (uint32_t r = blockIdx.x; r < nregions; r += gridDim.x) {
        const uint64_t n = min((uint64_t)cnt[r], desc_cap) * du, o = offs[r] * du;
        const uint4 *src = desc + (uint64_t)r * desc_cap * du;
        for (uint64_t i = threadIdx.x; i < n; i += 256)
            if (o + i < out_cap) out[o + i] = src[i];
    }
}

// ---- walk_part_kernel: the second half of the two-kernel scan (strip_desc_kernel, tsx_kernels.h) -----------
// Reads strip descriptions (48 bases as 2-bit codes + 16 validity bits), ONE PER LANE, every lane busy: first
// window by the 4-bit-group LUT, 15 rolls, the keys into the level-1 rings, bursts to the workgroup's sub-lists.
// A batch = 512 strips = up to 8192 keys: four flushes per batch (one per four positions).  Workgroup g takes
// the descriptor regions g, g + G, ... (a region = what one wave of strip_desc_kernel wrote).
template <int NT>   // 512 threads, two workgroups per CU (up to 256 level-1 lists); 1024 threads where 512 lists leave room for one
__global__ __launch_bounds__(NT, 4) void walk_part_kernel(TableParams p, const uint4 *desc, uint64_t desc_cap,
                                                            const unsigned long long *desc_cnt, uint32_t nregions, int dbg,
                                                            uint64_t *dst, uint64_t dst_cap, unsigned long long *dst_cnt,
                                                            uint32_t nb, uint32_t shift, uint64_t *ovq_all,
                                                            uint32_t *ovq_cnt, uint32_t ovq_cap, uint64_t n_packed,
                                                            uint32_t dst_g0, uint32_t dst_gtot, int own_flags,
                                                            unsigned long long *emit_sum, int long_desc, uint32_t flush_q) {
    // own_flags: low two bits = 0 local run, 1 keep only the keys this shard owns, 2 minimizer exchange (below); 4 = this launch
    // APPENDS to the sub-lists and the overflow queue an earlier launch of the same step left (windows of a sharded step)
    const int own_only = own_flags & 3;
    const bool append = (own_flags & 4) != 0;
    constexpr int HOT_N = 8;
    __shared__ uint64_t s_hot_key[(NT / 64) * HOT_N];
    __shared__ uint32_t s_hot_cnt[(NT / 64) * HOT_N];
    __shared__ uint64_t s_roll[64];
    __shared__ uint64_t s_lut4[256];
    __shared__ uint64_t s_homh[4];
    __shared__ uint32_t s_njobs[2];
    __shared__ uint32_t s_ovn;
    __shared__ uint64_t s_ovk[OVF_N];
    __shared__ uint32_t s_ovc[OVF_N];
    extern __shared__ uint64_t s_part[];   // rings | flush descriptors | tail and head of every ring | cursors | jobs
    constexpr uint32_t CAP = 1u << SP_CAPBITS, cmask = CAP - 1;
    uint64_t *s_stage = s_part;
    unsigned long long *s_meta = reinterpret_cast<unsigned long long *>(s_part + ((size_t)nb << SP_CAPBITS));
    unsigned long long *s_th = s_meta + nb;
    uint32_t *s_cur = reinterpret_cast<uint32_t *>(s_th + nb);
    uint32_t *s_job = s_cur + nb;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wg = blockIdx.x;
    const uint32_t cap32 = (uint32_t)min(dst_cap, (uint64_t)0xFFFFFFF0u);
    // Sharded runs (descriptions gathered from every GPU, tsx_hip_shard_walk_device): the descriptions are ONE packed
    // array of n_packed entries cut into runs of desc_cap; only the keys this GPU owns are kept (own_only); the launch
    // fills lists (b, dst_g0 + g) of dst_gtot per bucket.  Local runs: n_packed = 0, dst_g0 = 0, dst_gtot = G.
    // own_only 2 (minimizer exchange, a table of this GPU's own): every key stays and is counted in emit_sum; no homopolymers.
    const uint32_t gl = dst_g0 + wg;
    auto word_of = [&](uint32_t b, uint32_t at) -> uint64_t * { return dst + ((uint64_t)(b * dst_gtot + gl) * (uint64_t)cap32 + at); };
    auto is_mine = [&](uint64_t hk) -> bool {
        if (own_only != 1) return true;
        const uint64_t h1[1] = {hk};
        return owner_shard<1>(p, h1) == p.shard;
    };
    if (tid < 64) s_roll[tid] = p.roll[tid];
    if (tid < 256) s_lut4[tid] = p.roll[64 + tid];
    if (tid < 4) {
        const uint64_t x = (0x5555555555555555ULL * (uint64_t)tid) & p.top_mask;
        uint64_t hh = 0;
        for (int grp = 0; grp < 16; ++grp) hh ^= p.roll[64 + grp * 16 + ((x >> (4 * grp)) & 15u)];
        s_homh[tid] = hh;
    }
    if (tid < (NT / 64) * HOT_N) { s_hot_key[tid] = 0; s_hot_cnt[tid] = 0; }
    if (tid < OVF_N) { s_ovk[tid] = 0; s_ovc[tid] = 0; }
    if (tid < 2) s_njobs[tid] = 0;
    if (tid == 0) s_ovn = (append && ovq_cnt) ? ovq_cnt[blockIdx.x] : 0u;
    for (uint32_t b = tid; b < nb; b += NT) { s_cur[b] = append ? (uint32_t)dst_cnt[(uint64_t)b * dst_gtot + dst_g0 + blockIdx.x] : 0u; s_th[b] = 0; }
    uint64_t *ovq = ovq_all ? ovq_all + (size_t)blockIdx.x * ovq_cap : nullptr;
    unsigned long long added = 0;   // (k-mers are counted by strip_desc_kernel)
    unsigned long long emitted = 0; // sharded runs: k-mer occurrences this GPU kept (sum over GPUs == k-mers scanned)
    uint32_t spilled = 0;
    const uint32_t k = (uint32_t)p.k;
    const uint32_t ngrp = (2u * k + 3u) / 4u;

    const TableParams *pk = (const TableParams *)__builtin_amdgcn_kernarg_segment_ptr();
    auto side_insert = [&](uint64_t hkey, uint64_t d) {
        if (dbg & 1) return;
        defer_append1(pk, hkey, d);
    };
    // a key that found its sub-list full: spill cache (a hot key hits it), overflow queue, deferred list
    auto spill = [&](uint64_t key) {
        ++spilled;
        const uint64_t kk = key ^ OVF_SALT;
        if (kk != 0) {
            const uint32_t slot = (uint32_t)(mix64(key) >> 40) & (OVF_N - 1);
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&s_ovk[slot]), 0ULL,
                                                     (unsigned long long)kk);
            if (old == 0ULL || old == kk) { atomicAdd(&s_ovc[slot], 1u); return; }
        }
        if (ovq) {
            const uint32_t at = atomicAdd(&s_ovn, 1u);
            if (at < ovq_cap) { ovq[at] = key; return; }
        }
        side_insert(key, 1);
    };
    uint32_t round = 0;
    // partition_ring_kernel's flush for one-word records and fixed-capacity lists: (A) one thread per list decides
    // how many words leave (whole 128-B lines of the destination, or everything at the end), (B) an octet of lanes
    // per list copies them.
    // wait_loads: the flush of a batch's first quarter waits for the NEXT batch's descriptions between its phases,
    // i.e. before the batch's first store: vmcnt is one in-order queue of loads and stores, and a wait at the top
    // of the next batch would also wait for this batch's last stores.
    auto flush = [&](bool all, bool wait_loads) {
        const uint32_t par = round & 1u;
        ++round;
        if (tid == 0) s_njobs[par ^ 1u] = 0;
        for (uint32_t b = tid; b < nb; b += NT) {
            const unsigned long long th = s_th[b];
            const uint32_t head = (uint32_t)(th >> 32);
            const uint32_t tail = min((uint32_t)th, head + CAP);   // arrivals past the ring went out directly
            const uint32_t avail = tail - head;
            const uint32_t at = s_cur[b];
            const uint32_t end = (at + avail) & ~(uint32_t)(PART_FLUSH - 1);
            const uint32_t nout = all ? avail : (end > at ? end - at : 0u);
            s_meta[b] = ((unsigned long long)at << 16) | ((unsigned long long)(head & cmask) << 8) | nout;
            s_th[b] = ((unsigned long long)(head + nout) << 32) | tail;
            s_cur[b] = at + nout;
            if (nout) s_job[atomicAdd(&s_njobs[par], 1u)] = b;
        }
        lds_barrier();
        if (wait_loads) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        const uint32_t oct = tid >> 3, ol = tid & 7;
        const uint32_t njobs = s_njobs[par];
        // about 45 % of the lists have a line to send after a half strip: two jobs per octet and pass
        constexpr int ITER = 2;
        for (uint32_t j0 = 0; j0 < njobs; j0 += ITER * (NT / 8)) {
            unsigned long long meta[ITER];
            uint32_t bj[ITER];
            uint64_t k0[ITER], k1[ITER];
#pragma unroll
            for (int u = 0; u < ITER; ++u) {
                const uint32_t j = j0 + oct + u * (NT / 8);
                bj[u] = (j < njobs) ? s_job[j] : 0u;
            }
#pragma unroll
            for (int u = 0; u < ITER; ++u) {
                const uint32_t j = j0 + oct + u * (NT / 8);
                meta[u] = (j < njobs) ? s_meta[bj[u]] : 0ULL;
            }
#pragma unroll
            for (int u = 0; u < ITER; ++u) {
                const uint32_t nout = (uint32_t)meta[u] & 0xFFu, hd = ((uint32_t)meta[u] >> 8) & 0xFFu;
                const uint64_t *ring = s_stage + (bj[u] << SP_CAPBITS);
                k0[u] = (ol < nout) ? ring[((hd + ol) ^ bj[u]) & cmask] : 0;     // (place ^ list: see the append)
                k1[u] = (ol + 8 < nout) ? ring[((hd + ol + 8) ^ bj[u]) & cmask] : 0;
            }
#pragma unroll
            for (int u = 0; u < ITER; ++u) {
                const uint32_t nout = (uint32_t)meta[u] & 0xFFu, hd = ((uint32_t)meta[u] >> 8) & 0xFFu;
                const uint32_t at = (uint32_t)(meta[u] >> 16);
                const uint64_t *ring = s_stage + (bj[u] << SP_CAPBITS);
                uint64_t *out = word_of(bj[u], at);
                if (ol < nout) { if (at + ol < cap32) out[ol] = k0[u]; else spill(k0[u]); }
                if (ol + 8 < nout) { if (at + ol + 8 < cap32) out[ol + 8] = k1[u]; else spill(k1[u]); }
                if (nout > 16)
                    for (uint32_t q = ol + 16; q < nout; q += 8) {
                        const uint64_t w = ring[((hd + q) ^ bj[u]) & cmask];
                        if (at + q < cap32) out[q] = w; else spill(w);
                    }
            }
        }
    };

    lds_barrier();
    for (uint32_t r = blockIdx.x; r < nregions; r += gridDim.x) {
        const uint32_t nr = n_packed ? (uint32_t)(((uint64_t)r * desc_cap < n_packed) ? min(desc_cap, n_packed - (uint64_t)r * desc_cap) : 0ULL)
                                     : (uint32_t)min((uint64_t)desc_cnt[r], desc_cap);
        // long descriptions (32 bytes, four strips each; a batch is NT / 4 of them): lanes 4q .. 4q+3 read the same one
        const uint32_t per = long_desc ? NT / 4 : NT, me = long_desc ? tid >> 2 : tid;
        const uint4 *rd = desc + (uint64_t)r * desc_cap * (long_desc ? 2u : 1u);
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4 dn = z4, dn2 = z4;
        if (me < nr) { if (long_desc) { dn = rd[2u * me]; dn2 = rd[2u * me + 1u]; } else dn = rd[me]; }
        for (uint32_t base = 0; base < nr; base += per) {
            const uint4 d = dn, d2 = dn2;
            if (base + per < nr) {   // next batch
                dn = z4; dn2 = z4;
                if (base + per + me < nr) {
                    if (long_desc) { dn = rd[2u * (base + per + me)]; dn2 = rd[2u * (base + per + me) + 1u]; }
                    else dn = rd[base + per + me];
                }
            }
            uint32_t cw0 = d.x, cw1 = d.y, cw2 = d.z, vm = d.w & 0xFFFFu;   // (bit 16: strip_desc_kernel's neighbour mark)
            if (long_desc) {   // strip t of the four: bases 16 t .. 16 t + 47, validity bits 16 t .. 16 t + 15
                const uint32_t t = tid & 3u;
                const uint32_t w0 = d.x, w1 = d.y, w2 = d.z, w3 = d.w, w4 = d2.x, w5 = d2.y;
                cw0 = (t == 0u) ? w0 : (t == 1u) ? w1 : (t == 2u) ? w2 : w3;
                cw1 = (t == 0u) ? w1 : (t == 1u) ? w2 : (t == 2u) ? w3 : w4;
                cw2 = (t == 0u) ? w2 : (t == 1u) ? w3 : (t == 2u) ? w4 : w5;
                const uint32_t vw = (t < 2u) ? d2.z : d2.w;
                vm = (vw >> ((t & 1u) * 16u)) & 0xFFFFu;
            }
            const bool wave_has = __ballot(vm != 0u) != 0ULL;   // only the last batch of a region has idle waves
            uint64_t h = 0;
            uint32_t inc = 0, single = 0;
            if (wave_has) {
                const uint64_t lo = (uint64_t)cw0 | ((uint64_t)cw1 << 32), hi = cw2;
                if (vm) {
                    const uint64_t x = lo & p.top_mask;
                    for (uint32_t grp = 0; grp < ngrp; ++grp) h ^= s_lut4[grp * 16u + ((uint32_t)(x >> (4u * grp)) & 15u)];
                }
                {
                    const uint32_t o = 2u * k, ws = o >> 5, sh = o & 31u;
                    const uint32_t w0 = (ws == 0u) ? cw0 : (ws == 1u) ? cw1 : cw2;
                    const uint32_t w1 = (ws == 0u) ? cw1 : (ws == 1u) ? cw2 : 0u;
                    inc = __funnelshift_r(w0, w1, sh);
                }
            uint32_t homm = 0;   // bit j: the k-mer at strip position j is a homopolymer
            if (own_only != 2) {   // (own_only 2, minimizer exchange: the sender took them out and counted them)
                const uint64_t dlo = lo ^ ((lo >> 2) | (hi << 62)), dhi = hi ^ (hi >> 2);
                uint64_t rlo = (dlo | (dlo >> 1)) & 0x5555555555555555ULL, rhi = (dhi | (dhi >> 1)) & 0x5555555555555555ULL;
                uint32_t span = 1;
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
                uint32_t x = ~(uint32_t)rlo & 0x55555555u;
                x = (x | (x >> 1)) & 0x33333333u;
                x = (x | (x >> 2)) & 0x0F0F0F0Fu;
                x = (x | (x >> 4)) & 0x00FF00FFu;
                homm = (x | (x >> 8)) & 0xFFFFu;
            }
            const uint32_t hv = vm & homm;
            single = vm & ~homm;
            if (__ballot(hv != 0u)) {
                for (uint32_t b = 0; b < 4; ++b) {
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
        }
            bool waited = false;   // the first flush of a batch waits for the next batch's descriptions
            // ---- the strip in four quarters of 4 positions: roll, append to the rings, flush ----------------------
            for (uint32_t j0 = 0; j0 < 16; j0 += 4) {
                const uint32_t s4 = (single >> j0) & 0xFu;
                if (wave_has) {
                    uint64_t hs[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hs[j] = h;
                        if (j0 + j < 15) {
                            const uint32_t idx = ((uint32_t)h & 3u) | (__builtin_amdgcn_ubfe(cw0, 2u * (j0 + j), 2u) << 2) |
                                                 (__builtin_amdgcn_ubfe(inc, 2u * (j0 + j), 2u) << 4);
                            h = (h >> 2) ^ s_roll[idx];
                        }
                    }
                    uint32_t m4 = s4;   // positions of this quarter whose keys this GPU keeps
                    if (own_only) {
                        if (own_only == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (((s4 >> j) & 1u) && !is_mine(hs[j])) m4 &= ~(1u << j);
                        }
                        emitted += (unsigned long long)__popc(m4);
                    }
                    if (__ballot(m4 != 0u) != 0ULL) {
                        uint32_t bq[4];
                        unsigned long long sl[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            bq[j] = (uint32_t)(hs[j] >> shift) & (nb - 1);
                            sl[j] = ((m4 >> j) & 1u) ? atomicAdd(&s_th[bq[j]], 1ULL) : 0ULL;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if ((m4 >> j) & 1u) {
                                const uint32_t b = bq[j];
                                if ((uint32_t)sl[j] - (uint32_t)(sl[j] >> 32) < CAP) {
                                    s_stage[(b << SP_CAPBITS) + (((uint32_t)sl[j] ^ b) & cmask)] = hs[j];
                                } else {   // ring full: the next place of the list directly
                                    const uint32_t at = atomicAdd(&s_cur[b], 1u);
                                    if (at < cap32) *word_of(b, at) = hs[j];
                                    else spill(hs[j]);
                                }
                            }
                        }
                    }
                }
                // A flush per quarter strip keeps the rings safe when every key stays (7 arrive per ring and quarter).  A
                // GPU of a sharded run keeps one key in N: it flushes every flush_q quarters (N >= 4: once per batch) --
                // three barriers and a pass over the rings less for each one skipped; a ring that fills up all the same
                // sends its keys straight to the list.
                if ((((j0 >> 2) + 1u) % flush_q) == 0u) {
                    lds_barrier();
                    flush(false, !waited);
                    waited = true;
                    lds_barrier();
                }
            }
        }
    }
    lds_barrier();
    flush(true, false);
    lds_barrier();
    for (uint32_t b = tid; b < nb; b += NT) dst_cnt[(uint64_t)b * dst_gtot + gl] = min(s_cur[b], cap32);
    if (tid < OVF_N && s_ovc[tid] && !(dbg & 1)) side_insert(s_ovk[tid] ^ OVF_SALT, s_ovc[tid]);
    if (tid == 0 && ovq_cnt) ovq_cnt[blockIdx.x] = min(s_ovn, ovq_cap);
    if (tid < (NT / 64) * HOT_N && s_hot_cnt[tid]) side_insert(s_hot_key[tid], s_hot_cnt[tid]);
    if (emit_sum) {
        for (int d = 32; d > 0; d >>= 1) emitted += __shfl_down(emitted, d, 64);
        if (lane == 0 && emitted) atomicAdd(emit_sum, emitted);
    }
    for (int d = 32; d > 0; d >>= 1) { added += __shfl_down(added, d, 64); spilled += __shfl_down(spilled, d, 64); }
    if (lane == 0) {
        if (added) atomicAdd(&p.stats[ST_KMERS], added);
        if (spilled) atomicAdd(&p.stats[ST_FALLBACK], (unsigned long long)spilled);
    }
}

// Inserts the overflow queues partition_ring_kernel left behind (one queue of `cap` records per
// workgroup) -- after the segment build, like every insert that goes to the table directly.
template <int WK>
__global__ __launch_bounds__(PART_NT) void overflow_insert_kernel(TableParams p, const uint64_t *ovq_all,
                                                                  const uint32_t *ovq_cnt, uint32_t cap, uint32_t nq) {
    constexpr int RW = RecWords<WK>::value;
    for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        const uint32_t n = ovq_cnt[qi];
        for (uint32_t i = threadIdx.x; i < n; i += PART_NT) {
            uint64_t h[WK];
#pragma unroll
            for (int t = 0; t < WK; ++t) h[t] = ovq_all[((size_t)qi * cap + i) * RW + t];
            insert_key<WK>(p, h, 1);
        }
    }
}

// Inserts the deferred list (DeferList, tsx_device.h): hot k-mers with their totals, records of full log
// regions, records that found sub-list and overflow queue full.  Keys of other shards are skipped (the
// exchanged hot lists of a sharded run reach every GPU).
template <int WK>
__global__ __launch_bounds__(PART_NT) void deferred_insert_kernel(TableParams p, const uint64_t *rec, const uint64_t *cnt,
                                                                  const unsigned long long *n_ptr, uint64_t n_fixed,
                                                                  uint64_t cap) {
    constexpr int RW = RecWords<WK>::value;
    const uint64_t n = min(n_ptr ? (uint64_t)*n_ptr : n_fixed, cap);
    // The list is mostly the same hot k-mers over and over (every scan wave drains its homopolymer cache, every
    // level-2 workgroup its spill cache -- and, on skewed input, the chunks of records that found their sub-list
    // full: the records of ONE level-2 workgroup side by side, i.e. the hot keys of one bucket).  A workgroup sums
    // equal keys of its contiguous share in an LDS table that lives as long as the share: a hot key costs one
    // same-address global atomic per WORKGROUP, not per entry.  A key that finds no place in 16 probes is
    // inserted directly.  Multi-word keys: word 0 is claimed by CAS, the other words published behind a ready flag.
    constexpr uint32_t DN = (RW == 1) ? 2048u : (RW == 2) ? 1024u : 512u;
    __shared__ uint64_t s_k[DN * RW];
    __shared__ unsigned long long s_c[DN];
    __shared__ uint32_t s_r[(RW > 1) ? DN : 1];
    for (uint32_t t = threadIdx.x; t < DN; t += PART_NT) { s_k[t * RW] = 0; s_c[t] = 0; if (RW > 1) s_r[t] = 0; }
    __syncthreads();
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = min(n, (uint64_t)blockIdx.x * per), hi = min(n, lo + per);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += PART_NT) {
        const uint64_t d = cnt ? cnt[i] : 1ULL;
        if (d == 0) continue;
        uint64_t h[WK];
#pragma unroll
        for (int t = 0; t < WK; ++t) h[t] = rec[i * RW + t];
        if (p.lg != p.l && owner_shard<WK>(p, h) != p.shard) continue;
        const uint64_t kk = h[0] ^ OVF_SALT;
        uint64_t mixin = h[0];
#pragma unroll
        for (int t = 1; t < WK; ++t) mixin ^= h[t] * 0x9E3779B97F4A7C15ULL;
        uint32_t slot = (uint32_t)(mix64(mixin) >> 40) & (DN - 1);
        bool done = false;
        for (int pr = 0; pr < 16 && kk != 0 && !done; ++pr) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&s_k[slot * RW]), 0ULL,
                                                     (unsigned long long)kk);
            if constexpr (RW == 1) {
                if (old == 0ULL || old == kk) { atomicAdd(&s_c[slot], (unsigned long long)d); done = true; }
            } else {
                if (old == 0ULL) {
#pragma unroll
                    for (int t = 1; t < WK; ++t) s_k[slot * RW + t] = h[t];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __hip_atomic_store(&s_r[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    atomicAdd(&s_c[slot], (unsigned long long)d);
                    done = true;
                } else if (old == kk && __hip_atomic_load(&s_r[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    bool same = true;
#pragma unroll
                    for (int t = 1; t < WK; ++t) same &= (s_k[slot * RW + t] == h[t]);
                    if (same) { atomicAdd(&s_c[slot], (unsigned long long)d); done = true; }
                }
                // (word 0 equal but not published yet, or another key: next place -- a key may then sit in two places
                // of the table of this workgroup, which costs a second global insert, not a wrong count)
            }
            slot = (slot + 1) & (DN - 1);
        }
        if (!done) insert_key<WK>(p, h, d);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < DN; t += PART_NT)
        if (s_c[t]) {
            uint64_t h[WK];
            h[0] = s_k[t * RW] ^ OVF_SALT;
#pragma unroll
            for (int u = 1; u < WK; ++u) h[u] = s_k[t * RW + u];
            insert_key<WK>(p, h, s_c[t]);
        }
}

// Write offsets from the level-1 histograms hist[b * G + g] (bucket-major), two steps:
//   offsets_rows_kernel    workgroup b: exclusive scan of row b (all regions' keys for
//                          bucket b) -> offs[b * G + g] relative to the bucket, row total
//   offsets_finish_kernel  one workgroup: exclusive scan of the nb row totals -> where
//                          each bucket starts in the packed array, and its size
// A region's absolute write offset for bucket b is bucket_start[b] + offs[b * G + g].
__global__ __launch_bounds__(1024) void offsets_rows_kernel(const uint32_t *hist, unsigned long long *offs, uint32_t G,
                                                            unsigned long long *row_total) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (uint32_t start = 0; start < G; start += 1024) {
        const uint32_t i = start + tid;
        const unsigned long long v = (i < G) ? hist[(size_t)b * G + i] : 0;
        const unsigned long long inc = wave_incl_scan64(v);
        if ((tid & 63) == 63) s_w[tid >> 6] = inc;
        __syncthreads();
        unsigned long long woff = 0;
        for (uint32_t w = 0; w < (tid >> 6); ++w) woff += s_w[w];
        const unsigned long long base = s_base;
        if (i < G) offs[(size_t)b * G + i] = base + woff + inc - v;
        __syncthreads();
        if (tid == 1023) s_base = base + woff + inc;
        __syncthreads();
    }
    if (tid == 0) row_total[b] = s_base;
}
__global__ __launch_bounds__(1024) void offsets_finish_kernel(uint32_t nb, unsigned long long *bucket_start,
                                                              unsigned long long *bucket_cnt) {
    // on entry bucket_cnt[b] = row total (nb <= 1024)
    __shared__ unsigned long long s_w[16];
    const uint32_t tid = threadIdx.x;
    const unsigned long long v = (tid < nb) ? bucket_cnt[tid] : 0;
    const unsigned long long inc = wave_incl_scan64(v);
    if ((tid & 63) == 63) s_w[tid >> 6] = inc;
    __syncthreads();
    unsigned long long woff = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) woff += s_w[w];
    if (tid < nb) bucket_start[tid] = woff + inc - v;
}

// Level 0 of a sharded run: split this workgroup's key log by owner GPU (fan-out
// 2..8) into the packed send buffer, at the exact offsets offsets_kernel derived
// from the scan kernel's per-owner histograms.  With so few lists no staging is
// needed: per destination one ballot gives every lane its place in a contiguous
// run, so consecutive lanes write consecutive keys.
__global__ __launch_bounds__(PART_NT) void split_owner_kernel(const uint64_t *src, const unsigned long long *src_cnt,
                                                              uint64_t src_cap, uint32_t nregions, uint64_t *dst,
                                                              const unsigned long long *offs,
                                                              const unsigned long long *offs_base,
                                                              const unsigned long long *owner_cnt, uint32_t nown,
                                                              uint32_t shift, uint64_t *dst_own, uint32_t own,
                                                              unsigned long long *key_sum) {
    __shared__ unsigned long long s_cur[8];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    // keys of owner `own` (this GPU) go to dst_own, packed from 0: they never travel
    // and the send buffer closes the gap they would have left (owners above `own` move down by their number)
    const unsigned long long own_base = dst_own ? offs_base[own] : 0ULL;
    const unsigned long long own_cnt = dst_own ? owner_cnt[own] : 0ULL;
    unsigned long long sum = 0;   // integrity: sum of every key handed on (mod 2^64), checked against the builders' sums
    for (uint32_t r = blockIdx.x; r < nregions; r += gridDim.x) {
        lds_barrier();
        if (tid < nown) s_cur[tid] = offs_base[tid] + offs[(size_t)tid * nregions + r];
        lds_barrier();
        const uint64_t n = min((uint64_t)src_cnt[r], src_cap);
        const uint64_t *in = src + (uint64_t)r * src_cap;
        // one returning LDS atomic per 64 keys, whatever the fan-out: lane d reserves the run of
        // destination d; the next keys are already in flight while a row is placed
        uint64_t nxt = (tid < n) ? in[tid] : 0;
        for (uint64_t base = 0; base < n; base += PART_NT) {
            const uint64_t i = base + tid;
            const bool have = i < n;
            const uint64_t key = nxt;
            nxt = (i + PART_NT < n) ? in[i + PART_NT] : 0;
            const uint32_t o = have ? ((uint32_t)(key >> shift) & (nown - 1)) : 0xFFFFFFFFu;
            unsigned long long my_mk = 0;
            uint32_t cnt_d = 0;
            for (uint32_t d = 0; d < nown; ++d) {
                const unsigned long long mk = __ballot(o == d);
                if (o == d) my_mk = mk;
                if (lane == d) cnt_d = (uint32_t)__builtin_popcountll(mk);
            }
            unsigned long long at = 0;
            if (lane < nown && cnt_d) at = atomicAdd(&s_cur[lane], (unsigned long long)cnt_d);
            at = __shfl(at, (int)(o & 63u), 64);
            if (have) {
                at += __builtin_popcountll(my_mk & ((1ULL << lane) - 1ULL));
                if (dst_own && o == own) dst_own[at - own_base] = key;
                else dst[at - (o > own ? own_cnt : 0ULL)] = key;
                sum += key;
            }
        }
    }
    if (key_sum) {
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
        if (lane == 0 && sum) atomicAdd(key_sum, sum);
    }
}

// Level-1 histogram of received keys: G regions given by (region_start, region_cnt) -- cuts of the pieces
// a shard received, own keys and one run per exchange window -- into hist[b * G + g], the layout
// offsets_kernel scans.
__global__ __launch_bounds__(PART_NT) void hist_kernel(const uint64_t *keys, uint32_t G, uint32_t nb, uint32_t shift,
                                                       uint32_t *hist, const unsigned long long *region_start,
                                                       const unsigned long long *region_cnt, unsigned long long *key_sum) {
    __shared__ uint32_t s_h[512];
    const uint32_t tid = threadIdx.x;
    unsigned long long sum = 0;   // integrity: sum of every key received (see split_owner_kernel)
    for (uint32_t g = blockIdx.x; g < G; g += gridDim.x) {
        lds_barrier();
        for (uint32_t b = tid; b < nb; b += PART_NT) s_h[b] = 0;
        lds_barrier();
        const uint64_t lo = region_start[g], hi = lo + region_cnt[g];
        // four independent loads per thread and round: the kernel is a stream, not a latency chain
        uint64_t i = lo + tid;
        for (; i + 3 * PART_NT < hi; i += 4 * PART_NT) {
            const uint64_t k0 = keys[i], k1 = keys[i + PART_NT], k2 = keys[i + 2 * PART_NT], k3 = keys[i + 3 * PART_NT];
            atomicAdd(&s_h[(uint32_t)(k0 >> shift) & (nb - 1)], 1u);
            atomicAdd(&s_h[(uint32_t)(k1 >> shift) & (nb - 1)], 1u);
            atomicAdd(&s_h[(uint32_t)(k2 >> shift) & (nb - 1)], 1u);
            atomicAdd(&s_h[(uint32_t)(k3 >> shift) & (nb - 1)], 1u);
            sum += k0 + k1 + k2 + k3;
        }
        for (; i < hi; i += PART_NT) {
            const uint64_t k0 = keys[i];
            atomicAdd(&s_h[(uint32_t)(k0 >> shift) & (nb - 1)], 1u);
            sum += k0;
        }
        lds_barrier();
        for (uint32_t b = tid; b < nb; b += PART_NT) hist[(size_t)b * G + g] = s_h[b];
    }
    if (key_sum) {
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
        if ((tid & 63) == 0 && sum) atomicAdd(key_sum, sum);
    }
}

// addKmer for HASHED keys (tiny sharded tables that cannot be partitioned); keys of other shards
// are skipped.  key_sum: see hist_kernel.
__global__ __launch_bounds__(PART_NT) void add_hashed_kernel(TableParams p, const uint64_t *keys, const uint64_t *counts,
                                                             uint64_t n, unsigned long long *key_sum) {
    unsigned long long sum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * PART_NT + threadIdx.x; i < n; i += (uint64_t)gridDim.x * PART_NT) {
        const uint64_t h[1] = {keys[i]};
        const uint64_t d = counts ? counts[i] : 1ULL;
        sum += h[0];
        if (d == 0 || owner_shard<1>(p, h) != p.shard) continue;
        insert_key<1>(p, h, d);
    }
    if (key_sum) {
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
        if ((threadIdx.x & 63) == 0 && sum) atomicAdd(key_sum, sum);
    }
}

// ---- the segment build (one-limb keys and slots): wave-level key streams --------------------------
// Round 1's form gave every LANE a private stream of keys with a register FIFO of loads in
// flight.  Its disassembly showed what that costs: the FIFO advances by register moves, a move of a load's
// destination needs that load finished, the wave's load counter is in order -- so every round in which ANY
// lane places a key (nearly all of them) waits for the load issued one round earlier: the loop runs at one
// HBM latency per round (~1000 cycles for ~40 instructions), and a lane's 12 keys take 22 rounds on
// average, 40 for the slowest lane of the workgroup.
// Here a WAVE owns a stream: batches of 64 consecutive keys, one key per lane, all of a segment's batches
// loaded up front by coalesced 512-byte reads into registers (BK batches per wave and pass), and the NEXT
// segment's lists are pulled into L2 meanwhile: every lane loads ONE dword of one 128-byte line of them (1024
// lanes = 128 KiB of lists) into a register nobody needs before the next segment starts, so the loop itself
// never waits for HBM.  (LDS-DMA into a dump area would cost no register at all, but the compiler orders
// every later LDS read behind a pending LDS-DMA with s_waitcnt vmcnt(0), which makes it synchronous.)
// A lane that has placed its key takes the next unconsumed key of the wave's stream: a ballot and a
// prefix count give it the stream position.  The two batches the wave is consuming sit in a 1 KiB ring of
// the wave's own behind the segment (written from the batch registers, one ds_write_b64 per 64 keys), so the
// lanes that ask for keys in a round read CONSECUTIVE 8-byte words: one conflict-free ds_read_b64.  (Fetching
// straight from the batch registers with ds_bpermute measured slower than the per-lane FIFO: a
// ds_bpermute_b32 occupies the LDS pipe for ~16 cycles, four of them per round and wave saturate it.)
// All lanes stay busy until the stream runs dry; the tail is one key's probe chain, not the sum of a lane's.
constexpr int BK = 16;   // batches (of 64 keys) a wave holds in registers per pass

__device__ __forceinline__ uint64_t pick_batch(const uint64_t (&B)[BK], uint32_t j) {   // j is wave-uniform
    uint64_t v = 0;
    switch (j) {
        case 0: v = B[0]; break;   case 1: v = B[1]; break;   case 2: v = B[2]; break;   case 3: v = B[3]; break;
        case 4: v = B[4]; break;   case 5: v = B[5]; break;   case 6: v = B[6]; break;   case 7: v = B[7]; break;
        case 8: v = B[8]; break;   case 9: v = B[9]; break;   case 10: v = B[10]; break; case 11: v = B[11]; break;
        case 12: v = B[12]; break; case 13: v = B[13]; break; case 14: v = B[14]; break; case 15: v = B[15]; break;
        default: break;
    }
    return v;
}

// The batches of pass `pass` of one wave: batch j of the pass is batch t = wi + (pass*BK + j)*nw of the
// wave group's list (`mine` keys at `base`); lane L holds key t*64 + L.
__device__ __forceinline__ void load_batches(const uint64_t *base, uint32_t mine, uint32_t wi, uint32_t nw,
                                             uint32_t pass, uint32_t lane, uint64_t (&B)[BK]) {
#pragma unroll
    for (int j = 0; j < BK; ++j) {
        const uint32_t idx = (wi + (pass * BK + (uint32_t)j) * nw) * 64u + lane;
        B[j] = (idx < mine) ? base[idx] : 0ULL;
    }
}
// keys of the wave's stream in that pass
__device__ __forceinline__ uint32_t pass_total(uint32_t mine, uint32_t wi, uint32_t nw, uint32_t pass) {
    uint32_t tot = 0;
    for (uint32_t j = 0; j < (uint32_t)BK; ++j) {
        const uint32_t first = (wi + (pass * BK + j) * nw) * 64u;
        tot += (first < mine) ? min(64u, mine - first) : 0u;
    }
    return tot;
}

// PRE: the lists hold PRE-FORMATTED records (format_record below: level 2 wrote them that way) -- the slot image of the
// key without reprobe count and counter, with the first probe position above it: the hand-out is a mask and a shift.
template <bool DIAG, bool PRE>   // DIAG: the ablation / diagnostic switches of TSX_HIP_DEBUG are compiled in
__global__ __launch_bounds__(1024) void build_segments_stream_kernel(TableParams p, const uint64_t *lists,
                                                                     const unsigned long long *list_start,
                                                                     const unsigned long long *list_cnt,
                                                                     uint64_t list_cap, uint32_t pieces, uint32_t nseg,
                                                                     int dbg_arg, int fresh, int look_ahead) {
    const int dbg = DIAG ? dbg_arg : 0;
    extern __shared__ uint64_t s_seg[];  // 2^S slots, then four batches of 64 keys per wave
    const uint32_t nslots = 1u << p.S;
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t *ring = s_seg + nslots + wave * 256u;   // batch b of the wave's stream sits in ring[(b & 3) * 64 ..]
    // The two shift amounts of split_key live in VECTOR registers on purpose: the kernel is short of scalar
    // registers, and the compiler otherwise re-reads them from the kernel-argument segment inside the probe
    // loop -- two dependent scalar loads (s_load + s_waitcnt lgkmcnt(0)) in every round.
    uint32_t sh_lg, sh_r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(sh_lg) : "s"(p.lg));
    asm volatile("v_mov_b32 %0, %1" : "=v"(sh_r) : "s"(p.R));
    const uint32_t npieces = list_start ? 1u : pieces;
    const uint32_t nw = (nt / 64u) / npieces;          // waves per list (pieces is a power of two <= 8)
    const uint32_t grp = wave / nw, wi = wave % nw;
    const uint32_t smask = (uint32_t)p.seg_mask;
    const uint64_t k0mask = p.k0mask;
    const uint32_t maxr = p.max_reprobes;
    const uint64_t one = 1ULL << p.cshift;
    // One-limb slots: the counter is the top C <= 32 bits, so the key bits (K0 = 64 - C of them) cover the whole low
    // dword -- the loop works on dwords: key_lo = e0_lo | i, the high dword carries the rest of the key and the counter.
    const uint32_t one_hi = 1u << (p.cshift - 32), kmask_hi = (uint32_t)(k0mask >> 32);
    // pre-formatted records: bits [0, rf) are the slot image (rf = R + F >= 32), bits [rf, rf + S) the first probe position
    const uint32_t pre_qsh = (uint32_t)(p.R + p.F) - 32u, pre_emask = (1u << ((p.R + p.F) - 32)) - 1u;

    // Sizes of the lists of segment `seg`: lane c holds the size of piece c (one load for the wave, not one
    // after the other), seg_total adds them up, my_list picks this wave group's.
    // list_sizes only LOADS (the low half of the counter; nothing looks at the value, so nothing waits for it, until
    // seg_total / my_list clamp it to the list capacity where it is used -- for the next segment: when the stream is dry)
    const uint32_t *cnt32 = reinterpret_cast<const uint32_t *>(list_cnt);
    const uint32_t cap32 = list_start ? 0xFFFFFFFFu : (uint32_t)min(list_cap, (uint64_t)0xFFFFFFFFu);
    auto list_sizes = [&](uint32_t seg) -> uint32_t {
        if (list_start) return lane == 0u ? cnt32[2ull * seg] : 0u;
        return lane < pieces ? cnt32[2ull * ((uint64_t)seg * pieces + lane)] : 0u;
    };
    auto seg_total = [&](uint32_t c) -> uint32_t {   // pieces <= 8
        asm volatile("" : "+v"(c));   // (keeps the clamp -- the first look at the loaded value -- down here)
        c = min(c, cap32);
        c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
        return __builtin_amdgcn_readfirstlane(c);
    };
    auto my_list = [&](uint32_t seg, uint32_t c, const uint64_t *&base) -> uint32_t {
        asm volatile("" : "+v"(c));
        c = min(c, cap32);
        if (list_start) { base = lists + (uint64_t)list_start[seg]; return __builtin_amdgcn_readfirstlane(c); }
        base = lists + ((uint64_t)seg * pieces + grp) * list_cap;
        return __builtin_amdgcn_readlane(c, grp);
    };


    uint64_t B[BK];
    // The first batches of the NEXT segment are loaded as soon as this wave's stream is dry (the batch registers
    // are free then) and arrive while the wave probes for its last keys and the segment is written out: a segment
    // costs ~23 us, of which the list sizes (scalar loads) and the keys (one HBM latency) used to be waited for
    // in the open at its start.
    uint32_t nx_seg = 0xFFFFFFFFu, nx_mine = 0;   // wave-uniform: the segment whose pass-0 batches B holds
    uint64_t nx_n = 0;
    const uint64_t *nx_base = nullptr;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const bool have = (nx_seg == seg);
        uint32_t sizes = 0;
        if (!have) sizes = list_sizes(seg);
        const uint64_t n = have ? nx_n : seg_total(sizes);
        uint64_t *slots = p.table + ((uint64_t)seg << p.S);
        if (n == 0) {
            if (fresh) {   // nothing to insert, but the stale slots must go
                for (uint32_t i = tid * 2; i < nslots; i += nt * 2)
                    *reinterpret_cast<uint4 *>(&slots[i]) = make_uint4(0, 0, 0, 0);
                if (tid == 0) p.seg_dirty[seg] = 0;
            }
            continue;
        }
        const bool dirty = !fresh && p.seg_dirty[seg] != 0;
        const uint64_t *base = nx_base;
        uint32_t mine = nx_mine;
        if (!have) {
            mine = my_list(seg, sizes, base);
            load_batches(base, mine, wi, nw, 0, lane, B);
        }
        // the next segment's list sizes: on their way now, looked at when the stream is dry
        const uint32_t seg2 = seg + gridDim.x;
        uint32_t sizes2 = 0;
        if (seg2 < nseg) sizes2 = list_sizes(seg2);
        lds_barrier();  // previous segment fully written out
        if (dirty) {
            for (uint32_t i = tid * 2; i < nslots; i += nt * 2)
                *reinterpret_cast<uint4 *>(&s_seg[i]) = *reinterpret_cast<const uint4 *>(&slots[i]);
        } else {
            for (uint32_t i = tid * 2; i < nslots; i += nt * 2)
                *reinterpret_cast<uint4 *>(&s_seg[i]) = make_uint4(0, 0, 0, 0);
        }
        lds_barrier();
        const uint32_t npass = ((mine + 63u) / 64u + nw * BK - 1u) / (nw * BK);   // wave-group uniform
        for (uint32_t pass = 0; pass < npass && !(dbg & 2); ++pass) {
            if (pass > 0) load_batches(base, mine, wi, nw, pass, lane, B);   // long lists only: not prefetched
            const uint32_t total = __builtin_amdgcn_readfirstlane(pass_total(mine, wi, nw, pass));   // (a scalar, said so)
            uint32_t cb = 0, off = 0, taken = 0;    // wave-uniform: current batch, keys consumed of it, keys consumed in all
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's reads of the ring are done
            // the batches must have arrived: the one wait for loads of this segment, spelled out so that the
            // prefetch below is issued behind it (and stays in flight), not in front of it
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
            ring[lane] = B[0];
            ring[64u + lane] = B[1];
            ring[128u + lane] = B[2];
            ring[192u + lane] = B[3];
            // The loop is bound by instruction issue (16 waves share 4 SIMDs: every instruction of the round costs
            // the round ~8 cycles; the LDS pipe would allow ~430 cycles per round, scripts/lds_cas_chain.hip), so it
            // is kept lean: a lane's state is (e0, q, i) with i == 0 meaning "holds no key"; nothing else lives
            // across rounds, the diagnostic switches are compiled out of the production instance.
            uint32_t e0_lo = 0, e0_hi = 0;
            uint32_t i = 0, q = 0;
            unsigned long long d_rounds = 0, d_t0 = 0;
            if (DIAG && (dbg & 16)) d_t0 = __builtin_amdgcn_s_memtime();
            // one probe for every lane that holds a key
            auto probe = [&]() {
                if (i != 0u) {
                    const uint32_t key_lo = e0_lo | i;
                    const unsigned long long old =
                        atomicCAS(reinterpret_cast<unsigned long long *>(&s_seg[q]), 0ULL,
                                  ((unsigned long long)(e0_hi | one_hi) << 32) | key_lo);
                    // (no && below: a short-circuit is a branch, and this loop pays for every instruction)
                    const uint32_t old_lo = (uint32_t)old, old_hi = (uint32_t)(old >> 32);
                    const bool empty = (old == 0ULL);
                    const bool same = (old_lo == key_lo) & ((old_hi & kmask_hi) == e0_hi);   // key_lo != 0: same implies !empty
                    bool placed = empty | same;
                    if (same) {
                        const unsigned long long prev =
                            atomicAdd(reinterpret_cast<unsigned long long *>(&s_seg[q]), (unsigned long long)one);
                        const uint64_t carry = ((prev >> p.cshift) + 1) >> p.C;
                        if (carry) sec_add(p, ((uint64_t)seg << p.S) | q, carry);
                    }
                    if (!placed & (i >= maxr)) {
                        atomicAdd(&p.stats[ST_FAIL], 1ULL);
                        placed = true;
                    }
                    if (DIAG && (dbg & 8)) placed = true;   // ablation: every key 'placed' by its first probe
                    ++i;
                    q = (q + i) & smask;
                    if (placed) i = 0u;
                }
            };
            // ---- main phase: hand the next keys of the stream to the lanes that hold none, then probe.  After a
            // hand-out at least one lane holds a key: no exit test in here.
            while (taken < total) {
                const unsigned long long nm = __ballot(i == 0u);
                if (nm) {
                    const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
                    // stream position cb * 64 + off + pre, modulo the ring's 256 words (batch b sits in quarter b & 3):
                    // consecutive stream positions = consecutive words of the ring: no bank conflicts
                    const uint64_t kf = ring[(((cb << 6) + off) + pre) & 255u];
                    const uint32_t left = total - taken;   // scalar
                    if (i == 0u && pre < left) {
                        i = 1u;
                        if (PRE) {
                            e0_lo = (uint32_t)kf;
                            e0_hi = (uint32_t)(kf >> 32) & pre_emask;
                            q = (uint32_t)(kf >> 32) >> pre_qsh;                // q_1, as level 2 left it
                        } else {
                            q = ((uint32_t)kf + 1u) & smask;                    // q_1 = q_0 + 1
                            const uint64_t e0 = ((kf >> sh_lg) << sh_r) & k0mask;   // split_key for WK = 1
                            e0_lo = (uint32_t)e0;
                            e0_hi = (uint32_t)(e0 >> 32);
                        }
                    }
                    const uint32_t got = min((uint32_t)__builtin_popcountll(nm), left);
                    taken = __builtin_amdgcn_readfirstlane(taken + got);
                    off = __builtin_amdgcn_readfirstlane(off + got);
                    if (off >= 64u) {   // batch cb is used up: its quarter of the ring takes batch cb + 4
                        off -= 64u;
                        cb = __builtin_amdgcn_readfirstlane(cb + 1u);
                        ring[(((cb + 3u) & 3u) << 6) + lane] = pick_batch(B, cb + 3u);
                    }
                }
                if (DIAG) ++d_rounds;
                probe();
            }
            // ---- the stream is dry: the batch registers take the next segment's first batches
            if (pass + 1u == npass && look_ahead < 4) {
                nx_seg = 0xFFFFFFFFu;
                if (seg2 < nseg) {
                    nx_n = seg_total(sizes2);
                    nx_seg = seg2;
                    if (nx_n) {
                        nx_mine = my_list(seg2, sizes2, nx_base);
                        load_batches(nx_base, nx_mine, wi, nw, 0, lane, B);
                    }
                }
            }
            // ---- the tail.  The wave works on its last keys only, and they meet the segment at its final load: the
            // longest of the ~2000 probe chains in flight is ~25 probes, ~16 rounds of a wave's ~38 run with a handful
            // of lanes.  A lane that is still probing looks at its next three probe positions as well (plain reads,
            // issued together) and steps over those that hold ANOTHER key: slots never become empty again, so a slot
            // seen taken by someone else stays out of the question; a slot seen empty or holding this key is where the
            // next round's CAS goes.  (Seven positions instead of three: no better, 6.04 vs 5.97 ms; looking ahead in
            // every round, not only in the tail: worse, 6.73 ms -- the main phase is bound by instruction issue.)
            while (__ballot(i != 0u) != 0ULL) {
                if (DIAG) ++d_rounds;
                probe();
                if (look_ahead && i != 0u && i + 3u < maxr) {
                    const uint32_t q1 = (q + i + 1u) & smask, q2 = (q1 + i + 2u) & smask, q3 = (q2 + i + 3u) & smask;
                    const uint64_t v0 = s_seg[q], v1 = s_seg[q1], v2 = s_seg[q2];
                    const uint64_t e0 = ((uint64_t)e0_hi << 32) | e0_lo;
                    const bool t0 = v0 != 0 && (v0 & k0mask) != (e0 | i);
                    const bool t1 = t0 && v1 != 0 && (v1 & k0mask) != (e0 | (i + 1u));
                    const bool t2 = t1 && v2 != 0 && (v2 & k0mask) != (e0 | (i + 2u));
                    if (t2) { i += 3u; q = q3; }
                    else if (t1) { i += 2u; q = q2; }
                    else if (t0) { i += 1u; q = q1; }
                }
            }
            if (DIAG && (dbg & 16) && lane == 0) {
                atomicAdd(&p.stats[ST_DBG0], d_rounds);
                atomicAdd(&p.stats[ST_DBG1], (unsigned long long)(__builtin_amdgcn_s_memtime() - d_t0));
            }
        }
        lds_barrier();
        if (!(dbg & 4))
            for (uint32_t i = tid * 2; i < nslots; i += nt * 2)
                *reinterpret_cast<uint4 *>(&slots[i]) = *reinterpret_cast<const uint4 *>(&s_seg[i]);
        if (tid == 0) p.seg_dirty[seg] = 1;
    }
}

// ---- the wave-stream build for multi-limb keys and slots ------------------------------------------------
// The same form as build_segments_stream_kernel (the per-lane form of round 2 had a load in flight that the wave
// waited for in nearly every round): a wave owns a stream of
// batches of 64 records, loaded up front into registers (BKW batches of RW words per wave and pass); the batch
// being consumed sits in the wave's own slice of LDS behind the segment (64 records; a round hands out keys of
// that one batch only), a lane that has placed its key takes the next record of it.  The slot protocol is the
// wide kernel's: limb 0 claimed by CAS with LOCK, limbs 1..W-1 written, limb 0 stored without LOCK.
template <int RW> struct WideBatches { static constexpr int value = (RW == 1) ? 16 : (RW == 2) ? 12 : 6; };

template <int WK>
__global__ __launch_bounds__(1024) void build_segments_wide_stream_kernel(TableParams p, const uint64_t *lists,
                                                                          const unsigned long long *list_start,
                                                                          const unsigned long long *list_cnt,
                                                                          uint64_t list_cap, uint32_t pieces, uint32_t nseg,
                                                                          int fresh) {
    constexpr int RW = RecWords<WK>::value;
    constexpr int BKW = WideBatches<RW>::value;
    extern __shared__ uint64_t s_seg[];  // 2^S slots of W words, then 64 records per wave
    const uint32_t W = (uint32_t)p.W;
    const uint32_t nwords = (1u << p.S) * W;
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t *ring = s_seg + nwords + wave * (64u * RW);
    const uint32_t npieces = list_start ? 1u : pieces;
    const uint32_t nw = (nt / 64u) / npieces;          // waves per list (pieces is a power of two <= 8)
    const uint32_t grp = wave / nw, wi = wave % nw;
    const uint32_t smask = (uint32_t)p.seg_mask;
    const uint64_t k0mask = p.k0mask, lock = p.lock_bit;
    const uint32_t maxr = p.max_reprobes;
    const uint64_t one = 1ULL << p.cshift;

    // list sizes of a segment in one load per wave (lane c: piece c), as in build_segments_stream_kernel
    auto list_sizes = [&](uint32_t seg) -> uint32_t {
        if (list_start) return lane == 0u ? (uint32_t)list_cnt[seg] : 0u;
        return lane < pieces ? (uint32_t)min((uint64_t)list_cnt[(uint64_t)seg * pieces + lane], list_cap) : 0u;
    };
    auto seg_total = [&](uint32_t c) -> uint32_t {
        c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
        return __builtin_amdgcn_readfirstlane(c);
    };
    auto my_list = [&](uint32_t seg, uint32_t c, const uint64_t *&base) -> uint32_t {
        if (list_start) { base = lists + (uint64_t)list_start[seg] * RW; return __builtin_amdgcn_readfirstlane(c); }
        base = lists + ((uint64_t)seg * pieces + grp) * list_cap * RW;
        return __builtin_amdgcn_readlane(c, grp);
    };

    uint64_t B[BKW][RW];
    auto load_pass = [&](const uint64_t *base, uint32_t mine, uint32_t pass) {
#pragma unroll
        for (int j = 0; j < BKW; ++j) {
            const uint32_t idx = (wi + (pass * BKW + (uint32_t)j) * nw) * 64u + lane;
            if (idx < mine) load_rec<RW>(base + (uint64_t)idx * RW, B[j]);
            else {
#pragma unroll
                for (int t = 0; t < RW; ++t) B[j][t] = 0;
            }
        }
    };
    // the next segment's first batches are loaded when the last batch of this one has been staged (B is free then)
    uint32_t nx_seg = 0xFFFFFFFFu, nx_mine = 0, nx_n = 0;
    const uint64_t *nx_base = nullptr;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const bool have = (nx_seg == seg);
        uint32_t sizes = 0;
        if (!have) sizes = list_sizes(seg);
        const uint64_t n = have ? nx_n : seg_total(sizes);
        uint64_t *slots = p.table + ((uint64_t)seg << p.S) * W;
        if (n == 0) {
            if (fresh) {
                for (uint32_t i = tid * 2; i < nwords; i += nt * 2)
                    *reinterpret_cast<uint4 *>(&slots[i]) = make_uint4(0, 0, 0, 0);
                if (tid == 0) p.seg_dirty[seg] = 0;
            }
            continue;
        }
        const bool dirty = !fresh && p.seg_dirty[seg] != 0;
        const uint64_t *base = nx_base;
        uint32_t mine = nx_mine;
        if (!have) {
            mine = my_list(seg, sizes, base);
            load_pass(base, mine, 0);
        }
        const uint32_t seg2 = seg + gridDim.x;
        uint32_t sizes2 = 0;
        if (seg2 < nseg) sizes2 = list_sizes(seg2);
        lds_barrier();  // previous segment fully written out
        if (dirty) {
            for (uint32_t i = tid * 2; i < nwords; i += nt * 2)
                *reinterpret_cast<uint4 *>(&s_seg[i]) = *reinterpret_cast<const uint4 *>(&slots[i]);
        } else {
            for (uint32_t i = tid * 2; i < nwords; i += nt * 2)
                *reinterpret_cast<uint4 *>(&s_seg[i]) = make_uint4(0, 0, 0, 0);
        }
        lds_barrier();
        const uint32_t npass = ((mine + 63u) / 64u + nw * BKW - 1u) / (nw * BKW);
        for (uint32_t pass = 0; pass < npass; ++pass) {
            if (pass > 0) load_pass(base, mine, pass);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // the batches have arrived: the one wait for loads of this pass
            uint32_t cb = 0;        // batch of the pass that sits in the ring
            uint32_t blen = 0;      // its length
            uint32_t off = 0;       // records of it handed out
            auto stage = [&](uint32_t j) {   // batch j of the pass -> ring (j wave-uniform); returns its length
                const uint32_t first = (wi + (pass * BKW + j) * nw) * 64u;
                const uint32_t len = (j < (uint32_t)BKW && first < mine) ? min(64u, mine - first) : 0u;
                if (len) {
                    uint64_t r[RW];
#pragma unroll
                    for (int t = 0; t < RW; ++t) r[t] = 0;
#pragma unroll
                    for (int jj = 0; jj < BKW; ++jj)
                        if ((uint32_t)jj == j) {
#pragma unroll
                            for (int t = 0; t < RW; ++t) r[t] = B[jj][t];
                        }
                    store_rec<RW>(ring + (uint64_t)lane * RW, r);
                }
                return len;
            };
            auto take_next = [&]() {   // B takes the next segment's first batches
                nx_seg = 0xFFFFFFFFu;
                if (seg2 < nseg) {
                    nx_n = seg_total(sizes2);
                    nx_seg = seg2;
                    if (nx_n) {
                        nx_mine = my_list(seg2, sizes2, nx_base);
                        load_pass(nx_base, nx_mine, 0);
                    }
                }
            };
            blen = stage(0);
            if (blen == 0 && pass + 1u == npass) take_next();
            uint64_t e0 = 0, hi[4] = {0, 0, 0, 0};
            uint32_t i = 0, q = 0, spins = 0;   // i == 0: the lane holds no key
            for (;;) {
                // ---- hand records of the staged batch to the lanes that hold none
                if (off < blen) {
                    const unsigned long long nm = __ballot(i == 0u);
                    if (nm) {
                        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
                        if (i == 0u && off + pre < blen) {
                            uint64_t rec[RW], h[WK], pos0;
                            load_rec<RW>(ring + (uint64_t)(off + pre) * RW, rec);
#pragma unroll
                            for (int t = 0; t < WK; ++t) h[t] = rec[t];
                            split_key<WK>(p, h, pos0, e0, hi);
                            i = 1u;
                            spins = 0;
                            q = ((uint32_t)pos0 + 1u) & smask;
                        }
                        const uint32_t got = min((uint32_t)__builtin_popcountll(nm), blen - off);
                        off = __builtin_amdgcn_readfirstlane(off + got);
                    }
                }
                if (off >= blen && blen) {   // the staged batch is used up (its reads are ahead of this write in LDS order)
                    cb = __builtin_amdgcn_readfirstlane(cb + 1u);
                    off = 0;
                    blen = stage(cb);
                    if (blen == 0 && pass + 1u == npass) take_next();   // that was the last one: B is free
                }
                if (__ballot(i != 0u) == 0ULL) {
                    if (blen == 0) break;   // stream dry and every key placed
                    continue;
                }
                // ---- one probe for every lane that holds a key.  No && and no else-if chain: every short-circuit is a branch,
                // and the loop is bound by instruction issue (the same cure as in build_segments_stream_kernel).
                if (i != 0u) {
                    unsigned long long *slot = reinterpret_cast<unsigned long long *>(&s_seg[(size_t)q * W]);
                    const uint64_t key0 = e0 | i;
                    const unsigned long long old = atomicCAS(slot, 0ULL, (unsigned long long)(key0 | lock | one));
                    const bool claimed = (old == 0ULL);
                    const bool mine0 = ((old & k0mask) == key0);           // limb-0 key bits equal (never so when claimed: i >= 1)
                    const bool locked = (old & lock) != 0ULL;
                    if (claimed && W > 1) {   // publish the other limbs, then the unlocked limb 0
                        for (uint32_t t = 1; t < W; ++t) slot[t] = hi[t - 1];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __hip_atomic_store(slot, (unsigned long long)(key0 | one), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    bool same = false;
                    if (mine0 & !locked) {    // (duplicates are rare on most inputs: this block is skipped by most waves)
                        same = true;
                        if (W > 1) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            for (uint32_t t = 1; t < W; ++t) same &= (slot[t] == hi[t - 1]);
                        }
                        if (same) {
                            const unsigned long long prev = atomicAdd(slot, (unsigned long long)one);
                            const uint64_t carry = ((prev >> p.cshift) + 1) >> p.C;
                            if (carry) sec_add(p, ((uint64_t)seg << p.S) | q, carry);
                        }
                    }
                    bool next = !claimed & !same & !(mine0 & locked);
                    if (mine0 & locked) {     // claimed by another lane, limbs not published yet: the same slot again next round (bounded)
                        if (++spins > (1u << 20)) { atomicAdd(&p.stats[ST_LOCKTO], 1ULL); spins = 0; next = true; }
                    }
                    bool placed = claimed | same;
                    if (next & (i >= maxr)) { atomicAdd(&p.stats[ST_FAIL], 1ULL); placed = true; }
                    const uint32_t i1 = i + (next ? 1u : 0u);
                    q = next ? ((q + i1) & smask) : q;
                    i = placed ? 0u : i1;
                }
            }
        }
        lds_barrier();
        for (uint32_t i = tid * 2; i < nwords; i += nt * 2)
            *reinterpret_cast<uint4 *>(&slots[i]) = *reinterpret_cast<const uint4 *>(&s_seg[i]);
        if (tid == 0) p.seg_dirty[seg] = 1;
    }
}

}  // namespace tsx
