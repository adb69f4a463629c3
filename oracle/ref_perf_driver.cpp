// ref_perf_driver.cpp -- TEST INFRASTRUCTURE, nothing here ships.
//
// A small driver over the REFERENCE's own serial table: it is compiled against the sources where
// they lie under /root/reference (oracle/Makefile, target _ref/ref_perf_driver; nothing is copied)
// and calls, per k-mer of a FASTQ file,
//     FASTXreader<FASTQEntry>::getEntries   (src/fastxutils/FastXReader.h:221-280)
//     TSXSeqUtils::fromSequence             (src/utils/SequenceUtils.h:86-160)
//     TSXHashMapPerf::addKmer               (src/tsxcount/TSXHashMapPerf.h:56-205)
// and afterwards TSXHashMap::getKmerCount(kmer) (src/tsxcount/TSXHashMap.h:548-638) for every
// distinct k-mer, printing `kmer<TAB>count` in sorted order.
//
// Why it exists: the reference's CLI aborts for k >= 40 in --mode=CAS and in main.cpp's --check
// loader, but the serial body those modes repeat (TSXHashMapPerf) counts k = 40..63 correctly as
// long as no counter overflows at k = 63 (UBigInt::operator% is "not yet implemented",
// UBigInt.h:691-696).  tests/golden/make_ref_runs.py records the sha256 of this output and
// tests/test_oracle.py holds the C restatement to it.
//
// usage: ref_perf_driver FASTQ K L S
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include <fastxutils/FastXReader.h>
#include <tsxcount/TSXHashMapPerf.h>
#include <tsxcount/TSXTypes.h>
#include <utils/SequenceUtils.h>

int main(int argc, char **argv) {
    if (argc != 5) {
        std::cerr << "usage: ref_perf_driver FASTQ K L S" << std::endl;
        return 2;
    }
    std::string path = argv[1];
    const int k = atoi(argv[2]), l = atoi(argv[3]), s = atoi(argv[4]);
    TSXHashMapPerf table((uint8_t)l, (uint32_t)s, (uint16_t)k, 1);
    MemoryPool<FIELDTYPE> *pool = table.getMemoryPool();
    std::map<std::string, uint64_t> seen;   // k-mer -> occurrences fed (reported on stderr only)
    uint64_t fed = 0;
    {
        FASTXreader<FASTQEntry> reader(&path);
        while (reader.hasNext()) {
            std::vector<FASTQEntry> *entries = reader.getEntries(40);
            for (size_t e = 0; e < entries->size(); ++e) {
                const std::string seq = entries->at(e).getSequence();
                for (size_t i = 0; i + (size_t)k <= seq.size(); ++i) {   // createKMers, testExecution.h:15-36
                    std::string word = seq.substr(i, (size_t)k);
                    TSX::tsx_kmer_t kmer = TSXSeqUtils::fromSequence(word, pool);
                    table.addKmer(kmer);
                    ++seen[word];
                    ++fed;
                }
            }
            delete entries;
        }
    }
    std::cerr << "fed " << fed << " k-mers, " << seen.size() << " distinct; table reports " << table.getKmerCount()
              << " used positions" << std::endl;
    for (std::map<std::string, uint64_t>::iterator it = seen.begin(); it != seen.end(); ++it) {
        std::string word = it->first;
        TSX::tsx_kmer_t kmer = TSXSeqUtils::fromSequence(word, pool);
        // a k-mer whose counter has overflowed: for k >= 40 the reference throws from the overflow walk
        // (findOverflowCounts -> UBigInt arithmetic, "char const*"); such k-mers are printed as `!` and
        // stay unpinned, every other k-mer of the file is still answered
        try {
            UBigInt count = table.getKmerCount(kmer);
            const uint64_t value = count.toUInt();
            std::cout << word << "\t" << value << "\n";
        } catch (const char *what) {
            std::cout << word << "\t!\n";
        } catch (const std::exception &e) {
            std::cout << word << "\t!\n";
        }
    }
    return 0;
}
