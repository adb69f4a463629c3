/*
 * tsxcount_hip.h -- C ABI of libtsxcount_hip.so, the MI355X (gfx950) k-mer
 * counting hash map that backs tsxCount's --mode=HIP.
 *
 * Every entry point is what a TSXHashMap subclass in the reference would bind
 * for this path; the reference interface each one replaces is cited as
 * file:line of mjoppich/tsxCount.  Plain pointers and sizes only.  All
 * functions return TSX_HIP_OK (0) or a negative TSX_HIP_E* code; none of them
 * falls back to a CPU path -- without a GPU they fail with TSX_HIP_ENODEVICE.
 *
 * k-mers cross the boundary 2-bit encoded exactly like UBigInt holds them in
 * the reference (src/utils/SequenceUtils.h:86-160): base i of the k-mer sits
 * in bits 2i,2i+1 (A=0 C=1 G=2 T=3), packed little-endian into
 * tsx_hip_key_limbs(k) = ceil(2k/64) uint64 limbs per k-mer.
 */
#ifndef TSXCOUNT_HIP_H
#define TSXCOUNT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsx_hip_map tsx_hip_map; /* opaque: one table on one GPU */

enum {
    TSX_HIP_OK = 0,
    TSX_HIP_EINVAL = -1,    /* bad argument; 2k <= l mirrors TSXException (TSXHashMap.h:91-94) */
    TSX_HIP_ENODEVICE = -2, /* no usable HIP device */
    TSX_HIP_ENOMEM = -3,    /* device allocation failed */
    TSX_HIP_EHIP = -4,      /* a HIP runtime call failed (see tsx_hip_last_error) */
    TSX_HIP_EFULL = -5,     /* a k-mer could not be placed: reference exit(42), TSXHashMap.h:340-343 */
    TSX_HIP_EOVERFLOW = -6, /* the secondary (count overflow) array is full */
    TSX_HIP_ERANGE = -7,    /* output buffer too small */
    TSX_HIP_ELOCK = -8      /* a multi-limb slot stayed locked past the spin bound: counts may be wrong */
};

/* Layout the library derived from (k, l, storagebits); see DESIGN.md. */
typedef struct tsx_hip_layout {
    int32_t k, l;           /* as given */
    int32_t key_limbs;      /* ceil(2k/64): limbs per k-mer at the boundary */
    int32_t entry_limbs;    /* uint64 limbs per table slot */
    int32_t func_bits;      /* 2k-l hashed-key bits stored in the slot (TSXTypes.h:39) */
    int32_t reprobe_bits;   /* width of the reprobe field stored next to them */
    int32_t count_bits;     /* in-slot counter width ("storage bits", TSXHashMap.h:83) */
    int32_t overflow_l;     /* log2 slots of the secondary overflow array */
    int32_t shard_bits;     /* the whole table has 2^(l+shard_bits) slots spread over 2^shard_bits GPUs */
    int32_t shard_index;    /* this GPU holds home slots [shard_index << l, (shard_index+1) << l) */
    uint32_t max_reprobes;  /* probes tried before TSX_HIP_EFULL */
    uint64_t slots;         /* 2^l (getMaxElements, TSXHashMap.h:162) */
    uint64_t table_bytes;   /* device bytes of the primary table */
} tsx_hip_layout;

/* Counters kept on the device; print_stats()/main.cpp:479-501 equivalents. */
typedef struct tsx_hip_stats {
    uint64_t kmers_added;      /* k-mer occurrences inserted ("add calls") */
    uint64_t insert_failures;  /* occurrences lost to TSX_HIP_EFULL */
    uint64_t overflow_carries; /* carries pushed to the secondary array */
    uint64_t overflow_failures;/* carries lost to TSX_HIP_EOVERFLOW */
    uint64_t distinct;         /* occupied slots = getKmerCount() (TSXHashMap.h:645) */
    uint64_t overflow_used;    /* occupied secondary slots */
    uint64_t lock_timeouts;    /* multi-limb claim spins that gave up (must be 0) */
    uint64_t fallback_inserts; /* keys the partitioned path inserted atomically because a list was full */
    uint64_t count_sum;        /* sum of getKmerCount(kmer) over every stored k-mer, read back from the slots
                                  and the secondary array: equals kmers_added when nothing was lost (the
                                  reference's --check walks every k-mer instead, main.cpp:224-396) */
} tsx_hip_stats;

int tsx_hip_key_limbs(int k);
const char *tsx_hip_strerror(int code);
const char *tsx_hip_last_error(void); /* text of the last HIP runtime failure in this thread */
int tsx_hip_device_count(void);

/* TSXSeqUtils::fromSequence (SequenceUtils.h:86-160).  Non-ACGT bytes get the
 * fixed code ((b>>1)^(b>>2))&3 where the reference draws rand()%2 bits.      */
int tsx_hip_encode(const char *seq, int k, uint64_t *limbs_out);
/* TSXSeqUtils::toSequence (SequenceUtils.h:47-84). out must hold k+1 bytes. */
int tsx_hip_decode(const uint64_t *limbs, int k, char *out);

/*
 * TSXHashMap(iL, iStorageBits, iK) (TSXHashMap.h:79-154) and
 * TSXHashMapCAS(iL, iStorageBits, iK, iThreads) (TSXHashMapCAS.h:239-245).
 *   storagebits  0 = widest counter that fits the slot; 1..32 = exactly that
 *                many in-slot bits, larger counts carry into the secondary
 *                array (the reference chains overflow slots instead,
 *                TSXHashMapPerf.h:699-881).
 *   overflow_l   log2 slots of the secondary array; 0 = max(10, l-4).
 *   hash_seed    seed of the bijective GF(2) mapping; the reference draws it
 *                from time(NULL) (BijectiveKMapping.h:84).  The matrix is dense:
 *                multiplication by a random element of GF(2^2k), for every k <= 127
 *                (the reference uses a unit triangular matrix, which makes the slot a
 *                function of the first l/2 bases only).
 *   device       HIP device ordinal.
 */
int tsx_hip_create(tsx_hip_map **out, int k, int l, int storagebits, int overflow_l,
                   uint64_t hash_seed, int device);
/*
 * One shard of a table that spans 2^shard_bits GPUs (one process per GPU): the whole
 * table has 2^(l + shard_bits) slots, this map holds the slot range of shard_index.
 * The owner of a k-mer is the top shard_bits bits of its home slot.  Inserts and
 * lookups of k-mers that another shard owns are ignored / answer 0, so the same
 * batch can be offered to every shard.  shard_bits = 0 is tsx_hip_create.
 */
int tsx_hip_create_shard(tsx_hip_map **out, int k, int l, int storagebits, int overflow_l,
                         uint64_t hash_seed, int device, int shard_bits, int shard_index);
void tsx_hip_destroy(tsx_hip_map *m);
int tsx_hip_get_layout(const tsx_hip_map *m, tsx_hip_layout *out);
/* Zero the table, the secondary array and the counters. */
int tsx_hip_clear(tsx_hip_map *m);
/* Wait for everything queued on the map's stream and report sticky errors
 * (TSX_HIP_EFULL / TSX_HIP_EOVERFLOW / TSX_HIP_ELOCK) raised by earlier inserts. */
int tsx_hip_sync(tsx_hip_map *m);

/*
 * countKMers (src/mains/main.cpp:104-218): FASTXreader<FASTQEntry>::getEntries
 * (FastXReader.h:221-280,307-385) + createKMers (testExecution.h:15-36) +
 * fromSequence + addKmer, for one whole FASTQ text.  Empty lines are skipped,
 * every 4 remaining lines are a record, line 2 is the sequence, every window
 * of k bytes of it is one k-mer.
 *
 * Both refuse a map created with shard_bits > 0 (TSX_HIP_EINVAL): a sharded table
 * is filled through tsx_hip_shard_scan_device / tsx_hip_shard_build_device.
 *
 * _host copies `n` bytes from host memory through pinned staging buffers.
 * _device takes a device pointer (16-byte aligned, text starts at a record
 * boundary), queues the work on `stream` (a hipStream_t, NULL = the map's own
 * stream) and returns without waiting; call tsx_hip_sync before reading.
 */
int tsx_hip_count_fastq_host(tsx_hip_map *m, const char *text, size_t n);
int tsx_hip_count_fastq_device(tsx_hip_map *m, const void *dev_text, size_t n, void *stream);

/*
 * Blocked gzip (BGZF, `bgzip`) input -- FastXReader.h:178-206 reads `.gz` through zlib (gzopen / gzgets), which
 * takes a BGZF file as ordinary multi-member gzip.  Here the members are found on the host (BC extra field) and
 * inflated ON THE DEVICE, one member per lane, CRC-32 and ISIZE of every member checked; the text never exists in
 * host memory.  tsx_hip_bgzf_index_host: TSX_HIP_EINVAL when the buffer is not BGZF (single-stream gzip: inflate it
 * with zlib and call tsx_hip_count_fastq_host).  tsx_hip_inflate_bgzf_host returns the inflated bytes (tests, tools).
 */
int tsx_hip_bgzf_index_host(const void *gz, size_t n, size_t *members, size_t *text_bytes);
int tsx_hip_inflate_bgzf_host(int device, const void *gz, size_t n, void *out_host, size_t out_cap, size_t *out_bytes);
int tsx_hip_count_fastq_bgzf_host(tsx_hip_map *m, const void *gz, size_t n);

/*
 * Lines per record of the texts handed to the count_fastq / shard_scan entry points: 4 = FASTQ
 * (FASTQEntry, FastXReader.h:62-95; the default), 2 = FASTA exactly as FASTXreader<FASTAEntry> reads it
 * (FastXReader.h:97-116: header line, ONE sequence line; sequences wrapped over several lines are not
 * joined there either).  Empty lines are dropped in both.
 */
int tsx_hip_set_record_lines(tsx_hip_map *m, int lines);

/*
 * Measurement hooks (no reference counterpart): with timing enabled every
 * FASTQ piece records HIP events on its launch stream around the line passes,
 * around count_fastq_kernel and around the partition + segment-build kernels.
 * get_timing waits for them, returns the summed milliseconds of the three
 * phases and the number of pieces (= count_fastq_kernel launches), and resets
 * the accumulation.
 */
int tsx_hip_set_timing(tsx_hip_map *m, int enable);
int tsx_hip_get_timing(tsx_hip_map *m, double *line_ms, double *count_ms, double *build_ms,
                       uint64_t *launches);
/*
 * The same accumulation split per stage: stage_ms[7] = line passes, scan kernel
 * (count_fastq_kernel, or strip_desc_kernel + the walk), radix level 1 (offsets + partition; in a sharded run also the
 * histogram of the received keys), radix level 2, the segment build kernel, the gap between the end of the
 * scan and the start of the partition phase (0 except in a sharded run, where the owner split and the key
 * exchange lie there), and the inserts that wait for the build (overflow queues, deferred list).
 * Levels, build and inserts are 0 on the atomic path.  Either get_* call resets the accumulation.
 */
int tsx_hip_get_stage_timing(tsx_hip_map *m, double *stage_ms, uint64_t *launches);
/*
 * Insert path of the FASTQ entry points: 0 = choose per call (partitioned when
 * the text is at least 1/32 of the table bytes and k <= 32), 1 = always the
 * atomic path (one 64-bit CAS per distinct key), 2 = always the partitioned
 * path (keys radix-scattered by table segment, segments built in LDS).  Both
 * paths give identical tables up to slot order inside a segment.
 */
int tsx_hip_set_path(tsx_hip_map *m, int path);

/*
 * TSXHashMap::addKmer (TSXHashMap.h:182; CAS variant TSXHashMapCAS.h:268) for
 * a batch of n encoded k-mers; counts == NULL adds 1 per k-mer, otherwise
 * counts[i] occurrences (used by the multi-GPU merge).
 */
int tsx_hip_add_kmers_host(tsx_hip_map *m, const uint64_t *kmers, const uint64_t *counts, size_t n);
int tsx_hip_add_kmers_device(tsx_hip_map *m, const void *dev_kmers, const void *dev_counts, size_t n,
                             void *stream);

/* TSXHashMap::getKmerCount(kmer) (TSXHashMap.h:548-638) for n k-mers. */
int tsx_hip_get_counts_host(tsx_hip_map *m, const uint64_t *kmers, size_t n, uint64_t *counts_out);
int tsx_hip_get_counts_device(tsx_hip_map *m, const void *dev_kmers, size_t n, void *dev_counts_out,
                              void *stream);

/* getKmerCountDebug(kmer) (TSXHashMap.h:477-545): the count AND the slot the k-mer sits in
 * (KmerCountDebug::iFirstPos; UINT64_MAX when absent) -- main.cpp's --check marks these slots. */
int tsx_hip_lookup_host(tsx_hip_map *m, const uint64_t *kmers, size_t n, uint64_t *counts_out,
                        uint64_t *slots_out);
/* getKmerStarts / getKmerStartsRef (TSXHashMap.h:650-658) as a bitmap of 2^l bits: bit (i & 7) of
 * byte (i >> 3) is set iff slot i holds a k-mer.  nbytes >= 2^l / 8. */
int tsx_hip_kmer_starts_host(tsx_hip_map *m, uint8_t *bits_out, size_t nbytes);

/* getKmerCount() / print_stats() / iAddKmerCount (TSXHashMap.h:645,390; main.cpp:486-500). */
int tsx_hip_get_stats(tsx_hip_map *m, tsx_hip_stats *out);

/*
 * TSXHashMap::getAllKmers (TSXHashMap.h:660-722) plus each k-mer's count.
 * _device writes up to cap entries into device buffers (kmers: cap*key_limbs
 * uint64, counts: cap uint64) and the number written to *dev_n (uint64);
 * order is unspecified.  _host sizes with tsx_hip_get_stats().distinct.
 */
int tsx_hip_dump_host(tsx_hip_map *m, uint64_t *kmers_out, uint64_t *counts_out, size_t cap,
                      size_t *n_out);
int tsx_hip_dump_device(tsx_hip_map *m, void *dev_kmers_out, void *dev_counts_out, size_t cap,
                        void *dev_n, void *stream);
/*
 * Multi-GPU merge, sender side: like dump_device, but entries are grouped by
 * owner rank = tsx_hip_owner(kmer, nranks) into nranks contiguous segments;
 * dev_seg_counts (nranks uint64) receives the segment sizes.  The segments
 * travel through an RCCL all-to-all and are inserted on the owner with
 * tsx_hip_add_kmers_device.  cap must be >= distinct.
 */
int tsx_hip_partition_device(tsx_hip_map *m, int nranks, void *dev_kmers_out, void *dev_counts_out,
                             size_t cap, void *dev_seg_counts, void *stream);
/*
 * getAllKmers restricted to the table slots [slot_lo, slot_hi): a sample of the table for
 * cross-checks at sizes where the whole dump would not fit (bench.py's check).  cap must be
 * >= the number of occupied slots in the range (slot_hi - slot_lo always suffices).
 */
int tsx_hip_dump_range_device(tsx_hip_map *m, uint64_t slot_lo, uint64_t slot_hi, void *dev_kmers_out,
                              void *dev_counts_out, size_t cap, void *dev_n, void *stream);
int tsx_hip_owner_host(const tsx_hip_map *m, const uint64_t *kmer, int nranks);

/* IBijectiveFunction::apply / inv_apply (IBijectiveFunction.h:26-27) on the host,
 * and the matrix rows (row i <-> output bit 2k-1-i, BijectiveKMapping.h:202-256). */
int tsx_hip_hash_apply(const tsx_hip_map *m, const uint64_t *kmer, uint64_t *key_out);
int tsx_hip_hash_invert(const tsx_hip_map *m, const uint64_t *key, uint64_t *kmer_out);
int tsx_hip_hash_rows(const tsx_hip_map *m, uint64_t *rows_out /* 2k x key_limbs */);

/*
 * Multi-GPU counting with a sharded table (k <= 32).  Reads shard across the GPUs;
 * what travels between them is hashed keys BEFORE they are built into a table, not
 * table slots afterwards:
 *   shard_scan_window_device  scans the window [win_off, win_off + win_len) of this GPU's
 *                       device text (n_total bytes; win_off a multiple of 16; windows of one
 *                       text are given in order, win_off == 0 restarts the line count, a
 *                       window may begin anywhere -- inside a line, inside a record) and writes
 *                       the hashed keys of all its k-mers grouped by owner GPU into dev_send
 *                       (capacity from tsx_hip_shard_send_capacity(win_len)):
 *                       dev_send_counts[o] = keys for owner o.  With dev_own != NULL the keys
 *                       this GPU owns go there instead (they never travel) and dev_send holds
 *                       the other owners' groups back to back.  Keys that carry a count (hot
 *                       k-mers merged on chip) go to the (dev_hot_keys, dev_hot_counts) list,
 *                       *dev_hot_n of them.  *dev_key_sum (optional) += sum of all keys
 *                       written (mod 2^64): the integrity check of the exchange.
 *   shard_scan_device   the same for a whole text as one window, own keys inside dev_send.
 *   -- all-to-all of the owner groups (RCCL), all-gather of the hot lists --
 *   shard_build_device  builds the received keys into this GPU's slot range (radix
 *                       partition + LDS segment build); *dev_key_sum (optional) += their sum.
 *                       shard_build_pieces_device: the same for several runs of keys at once.
 *   add_hashed_device   adds (hashed key, count) pairs, skipping other owners' keys.
 * tsxcount_amd/distributed.py: ShardedCounter.
 */
int tsx_hip_shard_send_capacity(tsx_hip_map *m, size_t text_bytes, size_t *keys_out);
int tsx_hip_shard_scan_window_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t win_off,
                                     size_t win_len, void *dev_send, size_t send_cap_keys, void *dev_own,
                                     size_t own_cap_keys, void *dev_send_counts, void *dev_hot_keys,
                                     void *dev_hot_counts, size_t hot_cap, void *dev_hot_n, void *dev_key_sum,
                                     void *stream);
int tsx_hip_shard_scan_device(tsx_hip_map *m, const void *dev_text, size_t n, void *dev_send,
                              size_t send_cap_keys, void *dev_send_counts, void *dev_hot_keys,
                              void *dev_hot_counts, size_t hot_cap, void *dev_hot_n, void *stream);
int tsx_hip_shard_build_device(tsx_hip_map *m, const void *dev_keys, size_t n_keys, void *dev_key_sum,
                               void *stream);
/* The same for keys that arrived in several runs (own keys and one run per exchange window): run i is
 * piece_cnt[i] keys at dev_keys + piece_off[i] (host arrays, in keys).  ONE partition + build for all of
 * them: a build costs a full pass over this GPU's slot range however few keys it brings. */
int tsx_hip_shard_build_pieces_device(tsx_hip_map *m, const void *dev_keys, const uint64_t *piece_off,
                                      const uint64_t *piece_cnt, size_t npieces, void *dev_key_sum, void *stream);
/* Level 1 window by window.  tsx_hip_shard_l1_window_device partitions the n_keys received keys of exchange
 * window `window` (of nwindows) by radix level 1 as soon as they have arrived -- the later windows are still being
 * exchanged -- into sub-lists of its own (window 0 plans for est_total_keys keys in all; more than that is not an
 * error, only slower).  tsx_hip_shard_build_l1_device then runs level 2 + the build over what the windows left:
 * only these two wait for the last window.  tsx_hip_shard_l1_supported: one-limb keys and a table split by two
 * radix levels (otherwise: tsx_hip_shard_build_pieces_device). */
int tsx_hip_shard_l1_supported(tsx_hip_map *m);
int tsx_hip_shard_l1_window_device(tsx_hip_map *m, const void *dev_keys, size_t n_keys, uint32_t window, uint32_t nwindows,
                                   size_t est_total_keys, void *dev_key_sum, void *stream);
int tsx_hip_shard_build_l1_device(tsx_hip_map *m, void *stream);

/* Description exchange (small world sizes).  A strip description (16 bytes: 48 bases as 2-bit codes + 16 validity
 * bits) stands for up to 16 k-mer occurrences, a key for one: instead of sending every key to its owner, every GPU
 * describes its text window (tsx_hip_shard_desc_window_device: dev_count[0] packed descriptions at dev_desc,
 * dev_kmer_sum += k-mer occurrences they stand for), the
 * descriptions are all-gathered, and every GPU walks all of them and keeps the keys it owns
 * (tsx_hip_shard_walk_device, one call per (window, source GPU) = slot of nslots; slot 0 plans for est_total_keys owned
 * keys; dev_emit_sum += k-mer occurrences kept -- over all GPUs that must equal the k-mers scanned).  N x the rolling
 * work for N/8 of the bytes on the wire.  long_desc: four neighbouring strips in one description of 32 bytes (96 bases +
 * 64 validity bits) -- half the bytes again; dev_desc / desc_cap / n_desc then count those.  Then
 * tsx_hip_shard_build_l1_device.  Same support as tsx_hip_shard_l1_supported. */
int tsx_hip_shard_desc_capacity(tsx_hip_map *m, size_t text_bytes, int long_desc, size_t *descs_out);
int tsx_hip_shard_desc_window_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t win_off, size_t win_len,
                                     int long_desc, void *dev_desc, size_t desc_cap, void *dev_count, void *dev_kmer_sum,
                                     void *stream);
int tsx_hip_shard_walk_device(tsx_hip_map *m, const void *dev_desc, size_t n_desc, int long_desc, uint32_t slot,
                              uint32_t nslots, size_t est_total_keys, void *dev_emit_sum, void *stream);
/* The same result in two kernels (owner-filtered walk into per-wave key logs, then radix level 1 over the logs): the
 * better form when this GPU keeps few of the keys it walks (world sizes >= 4). */
int tsx_hip_shard_filter_device(tsx_hip_map *m, const void *dev_desc, size_t n_desc, int long_desc, uint32_t slot,
                                uint32_t nslots, size_t est_total_keys, void *dev_emit_sum, void *stream);

int tsx_hip_add_hashed_device(tsx_hip_map *m, const void *dev_keys, const void *dev_counts, size_t n,
                              void *stream);

/* Minimizer exchange (any world size <= 16, 20 <= k <= 32; csrc/tsx_minimizer.h).  The owner of a k-mer is a function of
 * its minimizer (the m-mer with the smallest hash value, m = min(11, k - 15)), so runs of consecutive k-mers share an owner
 * and every GPU holds a WHOLE table (shard_bits = 0) of the k-mers it owns: no slot-range split, no merge, and nobody walks
 * a position it does not own.  tsx_hip_mini_window_device describes a text window and splits the strip descriptions by
 * owner: list o (packed, dev_counts[o] descriptions, a multiple of 64 -- holes carry no valid start) at
 * dev_desc + o * cap_per_owner * 16 bytes, cap_per_owner >= tsx_hip_mini_capacity(win_len); dev_counts[nranks + b] =
 * occurrences of the homopolymer k-mer of base b (A, C, G, T) in the window, which are NOT in the descriptions: the caller
 * adds the totals on the owner of each (tsx_hip_mini_owner_host, tsx_hip_add_kmers_device); dev_kmer_sum += all k-mer
 * occurrences of the window.  The receiver walks what it was sent with tsx_hip_shard_walk_device (long_desc = 2: every key
 * stays, the calls of a step -- slot 0 .. nslots - 1 -- append to ONE set of level-1 lists) and
 * builds with tsx_hip_shard_build_l1_device.  tsx_hip_mini_owner_host: the owner of each one-limb k-mer (lookups). */
int tsx_hip_mini_supported(tsx_hip_map *m);
int tsx_hip_mini_capacity(tsx_hip_map *m, size_t text_bytes, int nranks, size_t *descs_per_owner_out);
int tsx_hip_mini_window_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t win_off, size_t win_len,
                               int nranks, void *dev_desc, size_t cap_per_owner, void *dev_counts, void *dev_kmer_sum,
                               void *stream);
/* The same in two steps, for a step of several exchange windows: the text is described ONCE (line pass + strip descriptions,
 * kept in the map's scratch until the next describe), then split share by share -- share `part` of `nparts` of the described
 * strips, lists and counts as above, cap_per_owner >= tsx_hip_mini_part_capacity(len, nparts) -- so that the exchange of one
 * share runs while the next is split and the one before is walked.  dev_kmer_sum += all k-mer occurrences of the text. */
int tsx_hip_mini_part_capacity(tsx_hip_map *m, size_t text_bytes, uint32_t nparts, size_t *descs_per_owner_out);
int tsx_hip_mini_describe_device(tsx_hip_map *m, const void *dev_text, size_t n_total, size_t off, size_t len, void *dev_kmer_sum,
                                 void *stream);
int tsx_hip_mini_split_device(tsx_hip_map *m, uint32_t part, uint32_t nparts, int nranks, void *dev_desc, size_t cap_per_owner,
                              void *dev_counts, void *stream);
int tsx_hip_mini_owner_host(int k, int nranks, const uint64_t *kmers, size_t n, uint32_t *owners_out);


/*
 * The multi-GPU run as ONE call from C++ (src/mains/main.cpp:404-507: one command runs the job): a group is one table per
 * GPU of this node, driven by one host thread per GPU inside the library (csrc/tsx_multi.cpp).
 *   group_create            ngpus tables (devices[r] = HIP ordinal of rank r, NULL = 0..ngpus-1).  comm 0 = RCCL
 *                           (ncclCommInitAll; librccl is loaded on demand; one GPU per rank), comm 1 = device-to-device
 *                           copies behind a barrier (ranks may share a GPU: tests of world sizes the box cannot give RCCL)
 *   group_count_fastq_host  countKMers for N GPUs: the text is cut into ngpus shards of whole records (empty lines
 *                           dropped, 4 or 2 lines per record: FastXReader.h:62-116,307-385), rank r counts shard r into
 *                           its own table (tsx_hip_count_fastq_host), then group_merge
 *   group_merge             the merge of the per-GPU tables: entries grouped by owner = tsx_hip_owner(kmer, ngpus)
 *                           (tsx_hip_partition_device), ONE all-to-all of k-mers and counts, the owner clears and
 *                           re-inserts (tsx_hip_add_kmers_device).  Afterwards every k-mer lives on the GPU that owns it
 *   group_get_counts_host   getKmerCount(kmer) (TSXHashMap.h:548-638): every k-mer is asked of its owner
 *   group_get_stats         sums over the GPUs (distinct = getKmerCount(), TSXHashMap.h:645)
 * tsx_hip_group_map gives the table of one rank for everything else in this header.
 */
typedef struct tsx_hip_group tsx_hip_group;
int tsx_hip_group_create(tsx_hip_group **out, int ngpus, const int *devices, int k, int l, int storagebits,
                         int overflow_l, uint64_t hash_seed, int comm);
void tsx_hip_group_destroy(tsx_hip_group *g);
int tsx_hip_group_size(const tsx_hip_group *g);
tsx_hip_map *tsx_hip_group_map(tsx_hip_group *g, int rank);
const char *tsx_hip_group_comm_name(const tsx_hip_group *g);   /* "rccl" or "copy" */
const char *tsx_hip_group_last_error(void);
int tsx_hip_group_set_record_lines(tsx_hip_group *g, int lines);
/* exchange 0 (default): every GPU counts its shard into its own table, the tables are merged afterwards (any k);
 * exchange 1: the minimizer exchange (20 <= k <= 32, at most 16 GPUs) -- strip descriptions travel to the GPU that owns
 * their k-mers' minimizer BEFORE anything is built, nothing is merged (tsx_hip_group_merge is then a no-op), lookups go to
 * tsx_hip_mini_owner_host(kmer). */
int tsx_hip_group_set_exchange(tsx_hip_group *g, int mode);
int tsx_hip_group_exchange(const tsx_hip_group *g);
int tsx_hip_group_clear(tsx_hip_group *g);
int tsx_hip_group_count_fastq_host(tsx_hip_group *g, const char *text, size_t n);
int tsx_hip_group_merge(tsx_hip_group *g);
int tsx_hip_group_get_counts_host(tsx_hip_group *g, const uint64_t *kmers, size_t n, uint64_t *counts_out);
int tsx_hip_group_get_stats(tsx_hip_group *g, tsx_hip_stats *out);
uint64_t tsx_hip_group_exchanged_entries(const tsx_hip_group *g);   /* entries that changed GPU in the last merge */
/* Where group_count_fastq_host cuts a text: cuts_out[0 .. parts], shard i = [cuts_out[i], cuts_out[i + 1]); every cut is
 * a record boundary of the reference's reader.  Host logic only (no GPU). */
int tsx_hip_cut_records_host(const char *text, size_t n, int parts, int lines_per_record, size_t *cuts_out);

/*
 * Synthetic reads shaped like generateFakeSequences.py (500-1000 random bases
 * + 100-300 'A', '@seq<i>' header, '&' qualities), written as FASTQ text
 * straight into device memory.  Sizing call: dev_out == NULL returns the byte
 * count in *bytes_out and the number of k-mers (for k) in *kmers_out.
 * tsxcount_amd/synth.py is the same generator on the host (numpy).
 */
int tsx_hip_synth_fastq_device(uint64_t seed, uint64_t first_read, uint64_t n_reads, int k,
                               void *dev_out, size_t cap, uint64_t *bytes_out, uint64_t *kmers_out,
                               uint64_t *polya_kmers_out, int device, void *stream);

/*
 * Zipf-skewed synthetic reads (BASELINE config 4: contention / reprobe stress): n_templates random template sequences of
 * 2 * read_len bases, read r = a window of read_len bases of template pick_r, pick_r Zipf distributed: thr (host,
 * n_templates ascending uint64) are the upper ends of the templates' shares of [0, 2^64).  '@z<r>' headers, 'I'
 * qualities.  Sizing call: dev_out == NULL.  tsxcount_amd/synth.py: zipf_thresholds(), the numpy twin and the analytic
 * count of every k-mer (what the bench's check compares the table with).
 */
int tsx_hip_synth_zipf_device(uint64_t seed, uint64_t n_reads, uint32_t read_len, uint32_t n_templates, const uint64_t *thr,
                              void *dev_out, size_t cap, uint64_t *bytes_out, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TSXCOUNT_HIP_H */
