// TSXHashMapHIP.h -- the reference-side binding of --mode=HIP: a TSXHashMap subclass
// (src/tsxcount/TSXHashMap.h:68) whose table lives on an MI355X behind the C ABI of
// libtsxcount_hip.so (include/tsxcount_hip.h).  This is the file a maintainer of
// mjoppich/tsxCount adds as src/tsxcount/TSXHashMapHIP.h; it
// compiles only inside the reference tree (it includes the reference's headers) and is
// built here, against /root/reference where it lies, by the test-only recipe in the
// parity checker's Makefile -- the reference's own countKMers loop and its own --check
// (src/mains/main.cpp:104-396) then run against the HIP table through these overrides.
//
//   addKmer(kmer)            TSXHashMap.h:182   buffered, flushed in batches through
//                                               tsx_hip_add_kmers_host (thread safe: the
//                                               reference calls it from OpenMP tasks)
//   getKmerCount(kmer)       TSXHashMap.h:548   tsx_hip_get_counts_host
//   getKmerCountDebug(kmer)  TSXHashMap.h:477   tsx_hip_lookup_host (count + slot)
//   finish()                 --                 flush + fill the base class's m_iKmerStarts
//                                               (TSXHashMap.h:645-658: getKmerCount() and
//                                               getKmerStartsRef() are not virtual) from
//                                               tsx_hip_kmer_starts_host
//   countFastq(text, n)      main.cpp:132-218   the whole parallel region in one call
#ifndef TSXCOUNT_TSXHASHMAPHIP_REFBINDING_H
#define TSXCOUNT_TSXHASHMAPHIP_REFBINDING_H

#include <tsxcount/TSXHashMap.h>

#include <cstring>
#include <ctime>
#include <mutex>
#include <vector>

extern "C" {
#include <tsxcount_hip.h>
}

class TSXHashMapHIP : public TSXHashMap {
public:
    // mirrors TSXHashMapCAS(iL, iStorageBits, iK, iThreads) (TSXHashMapCAS.h:239-245).  The base
    // class keeps its own (host) array and k-mer-start bitmap of 2^iL places; only the bitmap is used.
    TSXHashMapHIP(uint8_t iL, uint32_t iStorageBits, uint16_t iK, uint8_t iThreads = 0)
        : TSXHashMap(iL, iStorageBits, iK) {
        // one MemoryPool slice per OpenMP thread (TSXHashMap.h:741-746): countKMers' tasks allocate their
        // UBigInts from the pool of the thread they run on
        this->setThreads(iThreads ? iThreads : 1);
        int rc = tsx_hip_create(&m_pDev, iK, iL, (int)iStorageBits, /*overflow_l=*/0,
                                /*hash_seed=*/(uint64_t)time(NULL), /*device=*/0);
        if (rc != TSX_HIP_OK) throw TSXException(tsx_hip_strerror(rc));
        m_iLimbs = (size_t)tsx_hip_key_limbs(iK);
        m_vPending.reserve(FLUSH_KMERS * m_iLimbs);
    }
    ~TSXHashMapHIP() override { tsx_hip_destroy(m_pDev); }

    // UBigInt holds base i of the k-mer in bits 2i, 2i+1 (TSXSeqUtils::fromSequence,
    // SequenceUtils.h:86-160): the limbs of the C ABI are the same bits, little endian.
    void toLimbs(const TSX::tsx_kmer_t &kmer, uint64_t *limbs) const {
        memset(limbs, 0, m_iLimbs * 8);
        uint32_t bits = const_cast<TSX::tsx_kmer_t &>(kmer).getBitCount();
        if (bits > 2u * m_iK) bits = 2u * m_iK;
        for (uint32_t i = 0; i < bits; ++i)
            if (kmer.getBit(i)) limbs[i >> 6] |= 1ULL << (i & 63);
    }

    bool addKmer(TSX::tsx_kmer_t &kmer, bool verbose = false, bool noPrimaryAddition = false) override {
        uint64_t limbs[4];
        toLimbs(kmer, limbs);
        std::lock_guard<std::mutex> g(m_oLock);
        m_vPending.insert(m_vPending.end(), limbs, limbs + m_iLimbs);
        ++iAddKmerCount;
        if (m_vPending.size() >= FLUSH_KMERS * m_iLimbs) flushLocked();
        return true;
    }

    UBigInt getKmerCount(TSX::tsx_kmer_t &kmer, bool verbose = false, uint32_t addReprobes = 0) override {
        return lookup(kmer).oCount;
    }

    KmerCountDebug getKmerCountDebug(TSX::tsx_kmer_t &kmer, bool verbose = false, uint32_t addReprobes = 0) override {
        return lookup(kmer);
    }

    // Called once after the last addKmer (main.cpp prints getKmerCount() right after the counting loop).
    void finish() {
        std::lock_guard<std::mutex> g(m_oLock);
        flushLocked();
        check(tsx_hip_sync(m_pDev));
        std::vector<uint8_t> bits((size_t)((getMaxElements() + 7) / 8));
        check(tsx_hip_kmer_starts_host(m_pDev, bits.data(), bits.size()));
        for (uint64_t i = 0; i < getMaxElements(); ++i)
            if ((bits[i >> 3] >> (i & 7)) & 1) m_iKmerStarts.setBit(i, 1);
    }

    // The whole countKMers parallel region (main.cpp:132-218) in one call, for callers that hold the text.
    void countFastq(const char *text, size_t n) {
        int rc = tsx_hip_count_fastq_host(m_pDev, text, n);
        if (rc == TSX_HIP_EFULL) exit(42);  // TSXHashMap.h:340-343
        check(rc);
    }

    uint64_t iAddKmerCount = 0;   // "add calls" (TSXHashMapCAS::iAddKmerCount, main.cpp:498)
    tsx_hip_map *m_pDev = nullptr;

private:
    static const size_t FLUSH_KMERS = 1 << 20;

    static void check(int rc) {
        if (rc == TSX_HIP_EFULL) {
            std::cerr << "Could not insert kmer" << std::endl;
            exit(42);   // TSXHashMap.h:340-343
        }
        if (rc != TSX_HIP_OK) throw TSXException(tsx_hip_strerror(rc));
    }

    void flushLocked() {
        if (m_vPending.empty()) return;
        check(tsx_hip_add_kmers_host(m_pDev, m_vPending.data(), nullptr, m_vPending.size() / m_iLimbs));
        m_vPending.clear();
    }

    KmerCountDebug lookup(TSX::tsx_kmer_t &kmer) {
        uint64_t limbs[4], count = 0, slot = 0;
        toLimbs(kmer, limbs);
        std::lock_guard<std::mutex> g(m_oLock);
        flushLocked();
        // --check asks getKmerCountDebug and then getKmerCount for the same k-mer (main.cpp:285-290)
        if (!(m_bLast && memcmp(m_aLast, limbs, m_iLimbs * 8) == 0)) {
            check(tsx_hip_lookup_host(m_pDev, limbs, 1, &count, &slot));
            memcpy(m_aLast, limbs, m_iLimbs * 8);
            m_iLastCount = count; m_iLastSlot = slot; m_bLast = true;
        }
        KmerCountDebug oRet = KmerCountDebug();
        oRet.oCount = UBigInt(m_iLastCount, this->m_pPool);
        oRet.iFirstPos = (m_iLastSlot == ~0ULL) ? 0 : m_iLastSlot;
        return oRet;
    }

    std::mutex m_oLock;
    std::vector<uint64_t> m_vPending;
    size_t m_iLimbs = 1;
    uint64_t m_aLast[4] = {0, 0, 0, 0}, m_iLastCount = 0, m_iLastSlot = 0;
    bool m_bLast = false;
};

#endif  // TSXCOUNT_TSXHASHMAPHIP_REFBINDING_H
