// tsx_partition.h -- the partitioned insert path (k <= 32, one-limb slots).
//
// Random 64-bit global atomics top out at ~17 G/s on MI355X whatever the
// footprint (profiles/round1_atomics_ubench.txt), so for large inputs the table
// is not updated in place.  Instead:
//   count_fastq_kernel   writes hashed keys to a per-workgroup log (no atomics)
//   partition_kernel     x1 or x2: radix-scatters keys by the high bits of their
//                        home slot into one list per table segment, through
//                        LDS staging and 64/128-B bursts
//   build_segments_kernel one workgroup per 2^S-slot segment: segment in LDS,
//                        inserts with LDS atomics, streamed back once
// All HBM traffic is sequential.  Anything that does not fit a list (skewed
// data) falls back to insert_key(), i.e. the atomic path: slower, same result.
#pragma once
#include "tsx_kernels.h"

namespace tsx {

constexpr int PART_NT = 256;
constexpr int PART_RPT = 4;      // keys per thread per batch
constexpr int PART_FLUSH = 8;    // records per burst (64 B)

// One radix level.  Source = `nregions` regions of `src_cap` records each
// (`src_cnt[r]` valid, clamped to src_cap).  Destination lists have `dst_cap`
// records; list index = (prefix ? r * nb : 0) + ((key >> shift) & (nb - 1)).
// gridDim.x = nregions * cpr: workgroup (r, c) takes batches c, c+cpr, ...
__global__ __launch_bounds__(PART_NT) void partition_kernel(TableParams p, const uint64_t *src,
                                                            const unsigned long long *src_cnt, uint64_t src_cap,
                                                            uint32_t nregions, uint32_t cpr, uint64_t *dst,
                                                            unsigned long long *dst_cnt, uint64_t dst_cap,
                                                            uint32_t nb, uint32_t shift, int prefix, uint32_t cap,
                                                            int dbg) {
    extern __shared__ uint64_t s_part[];       // nb * cap staged keys, then per list: count, flush size, offset
    uint64_t *s_stage = s_part;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_part + (size_t)nb * cap);
    const uint32_t tid = threadIdx.x;
    const uint32_t r = blockIdx.x / cpr, c = blockIdx.x % cpr;
    if (r >= nregions) return;
    for (uint32_t b = tid; b < nb; b += PART_NT) s_cnt[b] = 0;
    __syncthreads();
    const uint64_t n = min((uint64_t)src_cnt[r], src_cap);
    const uint64_t *in = src + (uint64_t)r * src_cap;
    const uint64_t list0 = prefix ? (uint64_t)r * nb : 0;
    constexpr uint64_t BATCH_REC = (uint64_t)PART_NT * PART_RPT;

    auto put_direct = [&](uint64_t key, uint32_t b) {
        const uint64_t li = list0 + b;
        const unsigned long long at = atomicAdd(&dst_cnt[li], 1ULL);
        if (at < dst_cap) dst[li * dst_cap + at] = key;
        else { const uint64_t h[1] = {key}; insert_key<1>(p, h, 1); }
    };
    // Flush, two steps.  (1) thread b reserves room for list b with one returning
    // atomic on its cursor: all lists in ONE instruction per wave, its latency is paid
    // once per batch.  (2) 8 consecutive lanes serve one list, lane q of the octet
    // moves staged key q (+8, +16): every store instruction writes whole 64-B sectors.
    uint32_t *s_out = s_cnt + nb;                                          // keys to flush per list
    unsigned long long *s_at = reinterpret_cast<unsigned long long *>(s_cnt + 2 * nb);  // reserved offsets
    auto flush = [&](bool all) {
        for (uint32_t b = tid; b < nb; b += PART_NT) {
            const uint32_t have = min(s_cnt[b], cap);
            const uint32_t nout = all ? have : (have & ~(uint32_t)(PART_FLUSH - 1));
            s_out[b] = nout;
            if (nout) s_at[b] = atomicAdd(&dst_cnt[list0 + b], (unsigned long long)nout);
        }
        __syncthreads();
        const uint32_t oct = tid >> 3, ol = tid & 7;
        for (uint32_t b = oct; b < nb; b += PART_NT / 8) {
            const uint32_t nout = s_out[b];
            if (nout == 0) continue;                       // uniform inside the octet
            const uint32_t have = min(s_cnt[b], cap);
            const unsigned long long at = s_at[b];
            uint64_t *out = dst + (list0 + b) * dst_cap;
            uint64_t *st = s_stage + (size_t)b * cap;
            uint64_t keep[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const uint32_t q = ol + 8 * u;
                if (q < nout) {
                    const uint64_t key = st[q];
                    if (at + q < dst_cap) out[at + q] = key;
                    else { const uint64_t h[1] = {key}; insert_key<1>(p, h, 1); }
                }
                // staged keys beyond nout move to the front (at most 7 of them)
                const uint32_t qs = nout + ol + 8 * u;
                keep[u] = (qs < have) ? st[qs] : 0;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const uint32_t qs = nout + ol + 8 * u;
                if (qs < have) st[ol + 8 * u] = keep[u];
            }
            if (ol == 0) s_cnt[b] = have - nout;
        }
    };

    // software pipeline: the next batch's keys are in flight while this one is staged
    uint64_t cur[PART_RPT], nxt[PART_RPT];
    const uint64_t stride = (uint64_t)cpr * BATCH_REC;
    uint64_t base = (uint64_t)c * BATCH_REC;
#pragma unroll
    for (int q = 0; q < PART_RPT; ++q) {
        const uint64_t i = base + (uint64_t)q * PART_NT + tid;
        cur[q] = (i < n) ? in[i] : 0;
    }
    for (; base < n; base += stride) {
        const uint64_t nbase = base + stride;
#pragma unroll
        for (int q = 0; q < PART_RPT; ++q) {
            const uint64_t i = nbase + (uint64_t)q * PART_NT + tid;
            nxt[q] = (i < n) ? in[i] : 0;
        }
#pragma unroll
        for (int q = 0; q < PART_RPT; ++q) {
            const uint64_t i = base + (uint64_t)q * PART_NT + tid;
            if (i < n) {
                const uint64_t key = cur[q];
                const uint32_t b = (uint32_t)(key >> shift) & (nb - 1);
                const uint32_t slot = atomicAdd(&s_cnt[b], 1u);
                if (slot < cap) s_stage[(size_t)b * cap + slot] = key;
                else put_direct(key, b);
            }
        }
        __syncthreads();
        flush(false);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PART_RPT; ++q) cur[q] = nxt[q];
    }
    flush(true);
}

// One workgroup builds one segment: slots [seg << S, (seg+1) << S) live in LDS
// while the segment's key list is inserted with LDS atomics (same slot format,
// same probe sequence as insert_key), then go back to HBM in one sweep.
// Segments that already hold data (seg_dirty) are loaded first; untouched
// segments with an empty list are skipped.
__global__ __launch_bounds__(1024) void build_segments_kernel(TableParams p, const uint64_t *lists,
                                                              const unsigned long long *list_cnt,
                                                              uint64_t list_cap, uint32_t nseg, int dbg) {
    extern __shared__ uint64_t s_seg[];  // 2^S slots
    const uint32_t nslots = 1u << p.S;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint64_t n = min((uint64_t)list_cnt[seg], list_cap);
        if (n == 0) continue;
        uint64_t *slots = p.table + ((uint64_t)seg << p.S);
        const bool dirty = p.seg_dirty[seg] != 0;
        __syncthreads();  // previous segment fully written out
        if (dirty) {
            for (uint32_t i = tid * 2; i < nslots; i += nt * 2)
                *reinterpret_cast<uint4 *>(&s_seg[i]) = *reinterpret_cast<const uint4 *>(&slots[i]);
        } else {
            for (uint32_t i = tid * 2; i < nslots; i += nt * 2)
                *reinterpret_cast<uint4 *>(&s_seg[i]) = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        const uint64_t *in = lists + (uint64_t)seg * list_cap;
        const uint64_t one = 1ULL << p.cshift;
        // keys are fetched BUILD_UNROLL at a time before any of them is inserted, so the
        // HBM latency is paid once per group instead of once per key
        constexpr int BUILD_UNROLL = 8;
        for (uint64_t r0 = 0; r0 < n; r0 += (uint64_t)nt * BUILD_UNROLL) {
            uint64_t keys[BUILD_UNROLL];
#pragma unroll
            for (int u = 0; u < BUILD_UNROLL; ++u) {
                const uint64_t r = r0 + (uint64_t)u * nt + tid;
                keys[u] = (r < n) ? ((dbg & 32) ? (r * 0x9E3779B97F4A7C15ULL) : in[r]) : 0;
            }
            // All BUILD_UNROLL probe chains of a thread advance together: one round issues
            // up to 8 independent LDS CAS, so a round costs one LDS round trip, and the
            // number of rounds is the longest chain in the wave, not the sum over keys.
            uint32_t q0v[BUILD_UNROLL], iv[BUILD_UNROLL];
            uint64_t e0v[BUILD_UNROLL];
            uint32_t live = 0;
#pragma unroll
            for (int u = 0; u < BUILD_UNROLL; ++u) {
                const uint64_t r = r0 + (uint64_t)u * nt + tid;
                q0v[u] = (uint32_t)(keys[u] & p.seg_mask);
                e0v[u] = ((keys[u] >> p.l) << p.R) & p.k0mask;  // split_key for WK = 1
                iv[u] = 1;
                if (r < n && !(dbg & 2)) live |= 1u << u;
            }
            while (live) {
                unsigned long long oldv[BUILD_UNROLL];
                uint32_t qv[BUILD_UNROLL];
#pragma unroll
                for (int u = 0; u < BUILD_UNROLL; ++u) {  // issue: up to 8 CAS in flight
                    const uint32_t i = iv[u];
                    qv[u] = (q0v[u] + ((i * (i + 1)) >> 1)) & (uint32_t)p.seg_mask;
                    oldv[u] = 1;
                    if (live & (1u << u))
                        oldv[u] = atomicCAS(reinterpret_cast<unsigned long long *>(&s_seg[qv[u]]), 0ULL,
                                            (unsigned long long)(e0v[u] | i | one));
                }
#pragma unroll
                for (int u = 0; u < BUILD_UNROLL; ++u) {  // resolve
                    if (!(live & (1u << u))) continue;
                    const uint32_t i = iv[u];
                    const uint64_t key0 = e0v[u] | i;
                    if (oldv[u] == 0ULL) {
                        live &= ~(1u << u);
                    } else if ((oldv[u] & p.k0mask) == key0) {
                        const unsigned long long prev =
                            atomicAdd(reinterpret_cast<unsigned long long *>(&s_seg[qv[u]]), (unsigned long long)one);
                        const uint64_t carry = ((prev >> p.cshift) + 1) >> p.C;
                        if (carry) sec_add(p, ((uint64_t)seg << p.S) | qv[u], carry);
                        live &= ~(1u << u);
                    } else if (i + 1 > p.max_reprobes) {
                        atomicAdd(&p.stats[ST_FAIL], 1ULL);
                        live &= ~(1u << u);
                    } else {
                        iv[u] = i + 1;
                    }
                }
            }
        }
        __syncthreads();
        if (!(dbg & 4))
            for (uint32_t i = tid * 2; i < nslots; i += nt * 2)
                *reinterpret_cast<uint4 *>(&slots[i]) = *reinterpret_cast<const uint4 *>(&s_seg[i]);
        if (tid == 0) p.seg_dirty[seg] = 1;
    }
}

}  // namespace tsx
