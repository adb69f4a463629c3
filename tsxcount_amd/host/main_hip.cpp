// main_hip.cpp -- the tsxCount command line (src/mains/main.cpp) with the one
// new mode this repo adds: --mode=HIP.  Options, defaults, console lines and
// the --check procedure follow main.cpp:30-40,404-507 and :222-396; the CPU
// modes stay in the reference build.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "TSXHashMapHIP.h"

struct arguments {
    int k = 14, l = 26, storagebits = 4, threads = 0;  // main.cpp:410-413
    std::string input_path, mode = "HIP", format;   // format: "", "fastq" or "fasta" ("" = by file name)
    bool check = false, checkabort = false;
    unsigned long long seed = 1;
    int device = 0;
    bool group = false;           // --gpus given: the group path, even for one GPU (its merge then runs through RCCL with one rank)
    int gpus = 1;                 // --gpus N: reads shard across N GPUs, per-GPU tables merged over RCCL
    std::string comm = "rccl";    // --comm=rccl|copy (copy: device-to-device copies, ranks may share a GPU)
    std::string exchange = "auto"; // --exchange=merge|mini|auto: tables merged after the count / minimizer exchange (20 <= k <= 32;
                                  // auto: from 4 GPUs on where it applies)
    std::vector<int> devices;     // --devices=a,b,...: HIP ordinal per rank (default 0 .. N-1)
};

static bool opt(const char *arg, const char *name, std::string &val) {
    std::string a(arg), n = std::string("--") + name;
    if (a == n) { val = ""; return true; }
    if (a.compare(0, n.size() + 1, n + "=") == 0) { val = a.substr(n.size() + 1); return true; }
    return false;
}

static int usage() {
    std::cerr << "Usage: tsxCount --input=FASTQ|FASTA[.gz] [--k=K] [--l=L] [--s=STORAGE] [--mode=HIP] [--threads=T]\n"
                 "                [--check] [--checkabort] [--seed=S] [--device=D] [--format=fastq|fasta]\n"
                 "                [--gpus=N [--comm=rccl|copy] [--devices=a,b,...] [--exchange=merge|mini|auto]]\n"
                 "Count k-mers on an MI355X. --check compares with FASTQ.<k>.count (kmer<TAB>count per line)."
              << std::endl;
    return 1;
}

// FastXReader.h:178-206: a .gz input read through zlib (any gzip stream, BGZF included)
static bool read_gz(const std::string &path, std::vector<char> &owned) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    static char buf[1 << 16];
    int r;
    while ((r = gzread(f, buf, sizeof buf)) > 0) owned.insert(owned.end(), buf, buf + r);
    gzclose(f);
    return r == 0;
}

// whole input in memory: plain files are mmap'ed; .gz (FastXReader.h:178-206 picks zlib mode by the same suffix
// test): a blocked gzip file (BGZF) is mmap'ed as it is and inflated on the GPU (bgzf = true), any other gzip
// stream goes through zlib here
static bool load_input(const std::string &path, std::vector<char> &owned, const char *&text, size_t &n, void *&map,
                       bool &bgzf, bool allow_bgzf) {
    map = nullptr;
    bgzf = false;
    if (path.size() > 3 && path.rfind(".gz") == path.size() - 3) {
        int zfd = open(path.c_str(), O_RDONLY);
        struct stat zst;
        if (allow_bgzf && zfd >= 0 && fstat(zfd, &zst) == 0 && zst.st_size > 0) {
            void *zm = mmap(nullptr, (size_t)zst.st_size, PROT_READ, MAP_PRIVATE, zfd, 0);
            if (zm != MAP_FAILED) {
                size_t members = 0, tb = 0;
                if (tsx_hip_bgzf_index_host(zm, (size_t)zst.st_size, &members, &tb) == TSX_HIP_OK) {
                    close(zfd);
                    std::cerr << "Input is BGZF: " << members << " members, " << tb << " bytes of text, inflated on the device"
                              << std::endl;
                    map = zm; text = (const char *)zm; n = (size_t)zst.st_size; bgzf = true;
                    return true;
                }
                munmap(zm, (size_t)zst.st_size);
            }
        }
        if (zfd >= 0) close(zfd);
        const bool ok = read_gz(path, owned);
        text = owned.data(); n = owned.size();
        return ok;
    }
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return false; }
    n = (size_t)st.st_size;
    if (n == 0) { close(fd); text = ""; return true; }
    map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) { map = nullptr; return false; }
    text = (const char *)map;
    return true;
}

static bool is_fasta(const arguments &a) {   // FASTA (two lines per record, FASTXreader<FASTAEntry>) by option or by file name
    std::string stem = a.input_path;
    if (stem.size() > 3 && stem.rfind(".gz") == stem.size() - 3) stem.resize(stem.size() - 3);
    auto ends = [&](const char *suf) { const std::string x(suf); return stem.size() >= x.size() && stem.compare(stem.size() - x.size(), x.size(), x) == 0; };
    return a.format == "fasta" || (a.format.empty() && (ends(".fa") || ends(".fasta") || ends(".fna")));
}

// "Added a total of ..." and the --check of main.cpp:224-396, for one table or a group of them
template <typename Map>
static int report_and_check(Map &oMap, const arguments &a, double dt) {
    tsx_hip_stats st = oMap.stats();
    std::cout << "Added a total of " << st.distinct << " different kmers" << std::endl;
    std::cerr << "add calls: " << st.kmers_added << std::endl;
    std::cerr << "count time [s]: " << dt << " (" << (dt > 0 ? st.kmers_added / dt : 0) << " k-mers/s, host to table)"
              << std::endl;
    int rc = 0;
    if (a.check) {  // main.cpp:224-396
        std::string sRefFilename = a.input_path + "." + std::to_string(a.k) + ".count";
        std::cout << "Checking kmer counts against manual hashmap ..." << std::endl;
        std::cerr << "Loading reference file: " << sRefFilename << std::endl;
        std::ifstream file(sRefFilename);
        if (!file.is_open()) {
            std::cerr << "Could not open " << sRefFilename << std::endl;
            return 4;
        }
        std::vector<uint64_t> limbs, expect, got;
        std::vector<std::string> names;
        std::string line;
        uint64_t iRefCount = 0, totalerrors = 0;
        auto flush = [&]() {
            if (expect.empty()) return;
            oMap.getKmerCounts(limbs, expect.size(), got);
            for (size_t i = 0; i < expect.size(); ++i)
                if (got[i] != expect[i]) {
                    ++totalerrors;
                    if (totalerrors <= 20)
                        std::cout << "kmer: ( " << names[i] << " ): " << got[i] << " Should be " << expect[i] << std::endl;
                    if (a.checkabort) exit(200);  // main.cpp:285-291
                }
            std::cout << "Checked " << expect.size() << " kmers" << std::endl;
            iRefCount += expect.size();
            limbs.clear(); expect.clear(); names.clear();
        };
        while (std::getline(file, line)) {
            size_t tab = line.find('\t');
            if (tab == std::string::npos) continue;
            std::string kmer = line.substr(0, tab);
            if ((int)kmer.size() != a.k) continue;
            tsx_kmer_t enc = oMap.fromSequence(kmer);
            limbs.insert(limbs.end(), enc.begin(), enc.end());
            expect.push_back(strtoull(line.c_str() + tab + 1, nullptr, 10));
            names.push_back(kmer);
            if (expect.size() >= 100000) flush();  // main.cpp:263
        }
        flush();
        std::cout << "total errors" << totalerrors << std::endl;
        std::cout << "Kmer count check completed." << std::endl;
        std::cout << "Reference kmer count: " << iRefCount << std::endl;
        std::cout << "tsxCount kmer count: " << st.distinct << std::endl;
        if (totalerrors || iRefCount != st.distinct) rc = 5;
    }
    oMap.print_stats();
    return rc;
}

// --gpus N: one table per GPU, the text cut into N shards of whole records, the tables merged over RCCL
// (tsx_hip_group_*); --check asks every k-mer of the GPU that owns it
static int run_group(const arguments &a) {
    std::cerr << "Creating TSXHashMap HIP on " << a.gpus << " GPUs" << std::endl;
    TSXHashMapHIPGroup oGroup(a.gpus, a.devices.empty() ? nullptr : a.devices.data(), (uint8_t)a.l, (uint32_t)a.storagebits,
                              (uint16_t)a.k, a.seed, a.comm == "copy" ? 1 : 0);
    if (is_fasta(a)) { oGroup.setRecordLines(2); std::cerr << "Format=FASTA (2 lines per record)" << std::endl; }
    // the minimizer exchange where it applies (auto: from 4 GPUs on, as bench.py); --exchange=mini insists on it
    if (a.exchange == "mini" || (a.exchange == "auto" && a.gpus >= 4 && a.gpus <= 16 && a.k >= 20 && a.k <= 32)) {
        try { oGroup.setExchange(1); }
        catch (const TSXException &e) { if (a.exchange == "mini") throw; }
    }
    std::cerr << "exchange: " << (oGroup.exchange() == 1 ? "minimizer owners (strip descriptions travel, nothing is merged)" : "per-GPU tables merged") << std::endl;
    std::vector<char> owned;
    const char *text = nullptr;
    size_t n = 0;
    void *map = nullptr;
    bool bgzf = false;
    // (.gz: inflated by zlib on the host, as the reference's reader does -- the record cuts need the text)
    if (!load_input(a.input_path, owned, text, n, map, bgzf, false)) {
        std::cerr << "Could not read " << a.input_path << std::endl;
        return 3;
    }
    auto t0 = std::chrono::steady_clock::now();
    oGroup.countFastq(text, n);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (map) munmap(map, n);
    std::cerr << (oGroup.exchange() == 1 ? "descriptions moved between GPUs by the last share: " : "entries moved between GPUs by the merge: ") << oGroup.exchangedEntries() << std::endl;
    return report_and_check(oGroup, a, dt);
}

int main(int argc, char *argv[]) {
    arguments a;
    for (int i = 1; i < argc; ++i) {
        std::string v;
        if (opt(argv[i], "k", v)) a.k = atoi(v.c_str());
        else if (opt(argv[i], "l", v)) a.l = atoi(v.c_str());
        else if (opt(argv[i], "s", v)) a.storagebits = atoi(v.c_str());
        else if (opt(argv[i], "threads", v)) a.threads = atoi(v.c_str());
        else if (opt(argv[i], "input", v)) a.input_path = v;
        else if (opt(argv[i], "mode", v)) a.mode = v;
        else if (opt(argv[i], "check", v)) a.check = true;
        else if (opt(argv[i], "checkabort", v)) a.checkabort = true;
        else if (opt(argv[i], "seed", v)) a.seed = strtoull(v.c_str(), nullptr, 10);
        else if (opt(argv[i], "format", v)) a.format = v;
        else if (opt(argv[i], "device", v)) a.device = atoi(v.c_str());
        else if (opt(argv[i], "gpus", v)) { a.gpus = atoi(v.c_str()); a.group = true; }
        else if (opt(argv[i], "comm", v)) a.comm = v;
        else if (opt(argv[i], "exchange", v)) a.exchange = v;
        else if (opt(argv[i], "devices", v)) {
            for (size_t at = 0; at < v.size();) {
                const size_t c = v.find(',', at);
                a.devices.push_back(atoi(v.substr(at, c == std::string::npos ? c : c - at).c_str()));
                if (c == std::string::npos) break;
                at = c + 1;
            }
        }
        else if (opt(argv[i], "help", v)) return usage();
        else if (argv[i][0] == '-') { std::cerr << "unknown option " << argv[i] << std::endl; return usage(); }
    }
    std::transform(a.mode.begin(), a.mode.end(), a.mode.begin(), ::toupper);
    if (a.input_path.empty()) return usage();

    std::cout << "Running with parameters " << std::endl;
    std::cerr << "K=" << a.k << std::endl;
    std::cerr << "L=" << a.l << std::endl;
    std::cerr << "StorageBits=" << a.storagebits << std::endl;
    std::cerr << "Check=" << (a.check ? "Yes" : "No") << std::endl;
    std::cerr << "Input=" << a.input_path << std::endl;
    std::cerr << "Threads=" << a.threads << std::endl;
    std::cerr << "Mode=" << a.mode << std::endl;
    if (a.mode != "HIP") {
        std::cerr << "This binary implements --mode=HIP only; SERIAL/PTHREAD/OMP/CAS/TSX are the reference's CPU modes."
                  << std::endl;
        return 2;
    }

    if (a.exchange != "merge" && a.exchange != "mini" && a.exchange != "auto") return usage();
    if (a.gpus < 1 || (a.comm != "rccl" && a.comm != "copy") || (!a.devices.empty() && (int)a.devices.size() != a.gpus)) return usage();
    try {
        if (a.group) return run_group(a);
        std::cerr << "Creating TSXHashMap HIP" << std::endl;
        TSXHashMapHIP oMap((uint8_t)a.l, (uint32_t)a.storagebits, (uint16_t)a.k, (uint8_t)a.threads, a.seed, a.device);
        if (is_fasta(a)) { oMap.setRecordLines(2); std::cerr << "Format=FASTA (2 lines per record)" << std::endl; }
        std::vector<char> owned;
        const char *text = nullptr;
        size_t n = 0;
        void *map = nullptr;
        bool bgzf = false;
        if (!load_input(a.input_path, owned, text, n, map, bgzf, true)) {
            std::cerr << "Could not read " << a.input_path << std::endl;
            return 3;
        }
        auto t0 = std::chrono::steady_clock::now();
        if (bgzf) {
            try {
                oMap.countFastqBgzf(text, n);
            } catch (const TSXException &e) {
                // the device path needs two batch-sized text buffers next to the table: without them the input is
                // read the way the reference reads it (zlib on the host) and counted through the staged host path
                if (e.code() != TSX_HIP_ENOMEM) throw;
                std::cerr << "BGZF on the device: " << e.what() << " -- reading through zlib instead" << std::endl;
                oMap.clear();
                if (!read_gz(a.input_path, owned)) { std::cerr << "Could not read " << a.input_path << std::endl; return 3; }
                oMap.countFastq(owned.data(), owned.size());
            }
        } else {
            oMap.countFastq(text, n);
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (map) munmap(map, n);
        return report_and_check(oMap, a, dt);
    } catch (const TSXException &e) {
        std::cerr << "TSXException: " << e.what() << std::endl;
        return e.code() == TSX_HIP_EFULL ? 42 : 10;  // exit(42): TSXHashMap.h:340-343
    }
}
