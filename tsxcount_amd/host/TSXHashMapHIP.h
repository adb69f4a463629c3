// TSXHashMapHIP.h -- host-side C++ mirror of the reference's TSXHashMap surface
// (src/tsxcount/TSXHashMap.h) for --mode=HIP.  Everything below the class is
// the C ABI of libtsxcount_hip.so (include/tsxcount_hip.h); there is no CPU
// counting path in this class.
//
// A reference maintainer would make this `class TSXHashMapHIP : public
// TSXHashMap` and convert UBigInt <-> uint64 limbs at the boundary (see
// INTEGRATION.md); here k-mers are std::vector<uint64_t> limbs in the same
// bit layout UBigInt uses (base i -> bits 2i,2i+1).
#ifndef TSXCOUNT_TSXHASHMAPHIP_H
#define TSXCOUNT_TSXHASHMAPHIP_H

#include <cstdint>
#include <exception>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "tsxcount_hip.h"

// TSXException (TSXHashMap.h:28-47)
class TSXException : public std::exception {
public:
    TSXException(std::string sText, int iCode = TSX_HIP_EINVAL) : m_sText(std::move(sText)), m_iCode(iCode) {}
    const char *what() const throw() override { return m_sText.c_str(); }
    int code() const { return m_iCode; }

protected:
    const std::string m_sText;
    const int m_iCode;
};

typedef std::vector<uint64_t> tsx_kmer_t;  // TSX::tsx_kmer_t (TSXTypes.h:23)

class TSXHashMapHIP {
public:
    // TSXHashMapCAS(iL, iStorageBits, iK, iThreads) (TSXHashMapCAS.h:239-245);
    // iThreads is accepted for CLI compatibility, the GPU picks its own launch width.
    TSXHashMapHIP(uint8_t iL, uint32_t iStorageBits, uint16_t iK, uint8_t iThreads = 0, uint64_t iHashSeed = 1,
                  int iDevice = 0)
        : m_iL(iL), m_iK(iK), m_iThreads(iThreads) {
        int rc = tsx_hip_create(&m_pMap, iK, iL, (int)iStorageBits, 0, iHashSeed, iDevice);
        check(rc);
        check(tsx_hip_get_layout(m_pMap, &m_oLayout));
        std::cerr << "Creating array with " << m_oLayout.table_bytes << " bytes for " << m_oLayout.slots
                  << " places." << std::endl;
        std::cerr << "Maximum number of allowed reprobes per element: " << m_oLayout.max_reprobes << std::endl;
    }
    ~TSXHashMapHIP() { tsx_hip_destroy(m_pMap); }
    TSXHashMapHIP(const TSXHashMapHIP &) = delete;
    TSXHashMapHIP &operator=(const TSXHashMapHIP &) = delete;

    uint64_t getMaxElements() const { return m_oLayout.slots; }  // TSXHashMap.h:162
    uint32_t getK() const { return m_iK; }                       // TSXHashMap.h:172
    int getThreads() const { return m_iThreads; }                // TSXHashMap.h:737
    const tsx_hip_layout &getLayout() const { return m_oLayout; }

    // TSXSeqUtils::fromSequence (SequenceUtils.h:86-160)
    tsx_kmer_t fromSequence(const std::string &seq) const {
        tsx_kmer_t out(m_oLayout.key_limbs);
        check(tsx_hip_encode(seq.c_str(), m_iK, out.data()));
        return out;
    }
    // TSXSeqUtils::toSequence (SequenceUtils.h:47-84)
    std::string toSequence(const tsx_kmer_t &kmer) const {
        std::string s(m_iK + 1, '\0');
        check(tsx_hip_decode(kmer.data(), m_iK, &s[0]));
        s.resize(m_iK);
        return s;
    }

    // addKmer (TSXHashMap.h:182); batches go through addKmers.
    bool addKmer(const tsx_kmer_t &kmer) {
        check(tsx_hip_add_kmers_host(m_pMap, kmer.data(), nullptr, 1));
        return true;
    }
    void addKmers(const std::vector<uint64_t> &limbs, size_t n, const uint64_t *counts = nullptr) {
        check(tsx_hip_add_kmers_host(m_pMap, limbs.data(), counts, n));
    }

    // getKmerCount(kmer) (TSXHashMap.h:548)
    uint64_t getKmerCount(const tsx_kmer_t &kmer) {
        uint64_t c = 0;
        check(tsx_hip_get_counts_host(m_pMap, kmer.data(), 1, &c));
        return c;
    }
    void getKmerCounts(const std::vector<uint64_t> &limbs, size_t n, std::vector<uint64_t> &out) {
        out.resize(n);
        check(tsx_hip_get_counts_host(m_pMap, limbs.data(), n, out.data()));
    }
    // getKmerCount() (TSXHashMap.h:645)
    uint64_t getKmerCount() { return stats().distinct; }

    // getAllKmers (TSXHashMap.h:660), with counts
    std::vector<tsx_kmer_t> getAllKmers(std::vector<uint64_t> *pCounts = nullptr) {
        size_t n = (size_t)stats().distinct, got = 0;
        std::vector<uint64_t> limbs((n ? n : 1) * m_oLayout.key_limbs), counts(n ? n : 1);
        check(tsx_hip_dump_host(m_pMap, limbs.data(), counts.data(), n ? n : 1, &got));
        std::vector<tsx_kmer_t> out(got);
        for (size_t i = 0; i < got; ++i)
            out[i].assign(limbs.begin() + i * m_oLayout.key_limbs, limbs.begin() + (i + 1) * m_oLayout.key_limbs);
        if (pCounts) { counts.resize(got); *pCounts = counts; }
        return out;
    }

    // FASTXreader<FASTAEntry> (FastXReader.h:97-116) reads two lines per record, FASTQEntry (:62-95) four
    void setRecordLines(int iLines) { check(tsx_hip_set_record_lines(m_pMap, iLines)); }

    // countKMers body (main.cpp:104-218): whole FASTQ text -> table
    void countFastq(const char *pText, size_t iBytes) { check(tsx_hip_count_fastq_host(m_pMap, pText, iBytes)); }
    // the same for a blocked gzip (BGZF) file image: inflated on the device (FastXReader.h:178-206 uses zlib)
    void countFastqBgzf(const void *pGz, size_t iBytes) {
        int rc = tsx_hip_count_fastq_bgzf_host(m_pMap, pGz, iBytes);
        if (rc == TSX_HIP_EINVAL) throw TSXException(std::string("BGZF input: ") + tsx_hip_last_error(), rc);
        check(rc);
    }

    // empties the table (the reference has no counterpart: its maps are filled once)
    void clear() { check(tsx_hip_clear(m_pMap)); }

    tsx_hip_stats stats() {
        tsx_hip_stats s;
        check(tsx_hip_get_stats(m_pMap, &s));
        return s;
    }

    // print_stats (TSXHashMap.h:390-395)
    void print_stats() {
        tsx_hip_stats s = stats();
        std::cerr << "Used fields: " << s.distinct << std::endl;
        std::cerr << "Available fields: " << (double)m_oLayout.slots << std::endl;
        std::cerr << "k=" << m_iK << " l=" << (uint32_t)m_iL << " entry limbs=" << m_oLayout.entry_limbs
                  << " storage bits=" << m_oLayout.count_bits << std::endl;
    }

    tsx_hip_map *handle() { return m_pMap; }

private:
    static void check(int rc) {
        if (rc == TSX_HIP_OK) return;
        std::string msg = tsx_hip_strerror(rc);
        if (rc == TSX_HIP_EHIP || rc == TSX_HIP_ENODEVICE || rc == TSX_HIP_ENOMEM) {
            msg += " (";
            msg += tsx_hip_last_error();
            msg += ")";
        }
        throw TSXException(msg, rc);
    }

    tsx_hip_map *m_pMap = nullptr;
    tsx_hip_layout m_oLayout;
    const uint8_t m_iL;
    const uint32_t m_iK;
    const uint8_t m_iThreads;
};

// The same surface over the GPUs of one node (tsx_hip_group_*, csrc/tsx_multi.cpp): reads shard across the GPUs,
// the per-GPU tables are merged over RCCL, every k-mer then lives on the GPU that owns it.
class TSXHashMapHIPGroup {
public:
    // piDevices: HIP ordinal per rank (nullptr: 0 .. iGpus-1); iComm 0 = RCCL, 1 = device copies (ranks may share a GPU)
    TSXHashMapHIPGroup(int iGpus, const int *piDevices, uint8_t iL, uint32_t iStorageBits, uint16_t iK, uint64_t iHashSeed = 1,
                       int iComm = 0)
        : m_iK(iK) {
        check(tsx_hip_group_create(&m_pGroup, iGpus, piDevices, iK, iL, (int)iStorageBits, 0, iHashSeed, iComm));
        check(tsx_hip_get_layout(tsx_hip_group_map(m_pGroup, 0), &m_oLayout));
        std::cerr << "Creating " << iGpus << " arrays with " << m_oLayout.table_bytes << " bytes for " << m_oLayout.slots
                  << " places each (" << tsx_hip_group_comm_name(m_pGroup) << " merge)." << std::endl;
    }
    ~TSXHashMapHIPGroup() { tsx_hip_group_destroy(m_pGroup); }
    TSXHashMapHIPGroup(const TSXHashMapHIPGroup &) = delete;
    TSXHashMapHIPGroup &operator=(const TSXHashMapHIPGroup &) = delete;

    const tsx_hip_layout &getLayout() const { return m_oLayout; }
    int size() const { return tsx_hip_group_size(m_pGroup); }
    void setRecordLines(int iLines) { check(tsx_hip_group_set_record_lines(m_pGroup, iLines)); }
    // 0: per-GPU tables merged after the count (any k); 1: the minimizer exchange (20 <= k <= 32, at most 16 GPUs)
    void setExchange(int iMode) { check(tsx_hip_group_set_exchange(m_pGroup, iMode)); }
    int exchange() const { return tsx_hip_group_exchange(m_pGroup); }
    void clear() { check(tsx_hip_group_clear(m_pGroup)); }
    // countKMers (main.cpp:104-218) over all GPUs + the merge of the tables
    void countFastq(const char *pText, size_t iBytes) { check(tsx_hip_group_count_fastq_host(m_pGroup, pText, iBytes)); }
    void getKmerCounts(const std::vector<uint64_t> &limbs, size_t n, std::vector<uint64_t> &out) {
        out.resize(n);
        check(tsx_hip_group_get_counts_host(m_pGroup, limbs.data(), n, out.data()));
    }
    tsx_kmer_t fromSequence(const std::string &seq) const {
        tsx_kmer_t out(m_oLayout.key_limbs);
        if (tsx_hip_encode(seq.c_str(), m_iK, out.data()) != TSX_HIP_OK) throw TSXException("bad k-mer", TSX_HIP_EINVAL);
        return out;
    }
    tsx_hip_stats stats() {
        tsx_hip_stats s;
        check(tsx_hip_group_get_stats(m_pGroup, &s));
        return s;
    }
    uint64_t exchangedEntries() const { return tsx_hip_group_exchanged_entries(m_pGroup); }
    void print_stats() {
        tsx_hip_stats s = stats();
        std::cerr << "Used fields: " << s.distinct << std::endl;
        std::cerr << "Available fields: " << (double)m_oLayout.slots * size() << std::endl;
        std::cerr << "k=" << m_iK << " l=" << m_oLayout.l << " x " << size() << " GPUs, entry limbs=" << m_oLayout.entry_limbs
                  << " storage bits=" << m_oLayout.count_bits << std::endl;
    }

private:
    static void check(int rc) {
        if (rc == TSX_HIP_OK) return;
        std::string msg = tsx_hip_strerror(rc);
        const std::string why = tsx_hip_group_last_error();
        if (!why.empty()) msg += " (" + why + ")";
        throw TSXException(msg, rc);
    }
    tsx_hip_group *m_pGroup = nullptr;
    tsx_hip_layout m_oLayout;
    const uint32_t m_iK;
};

#endif  // TSXCOUNT_TSXHASHMAPHIP_H
