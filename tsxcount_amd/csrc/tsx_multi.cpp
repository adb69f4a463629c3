// tsx_multi.cpp -- the multi-GPU run of --mode=HIP as C++ host code: one process, one host thread per GPU.
//
// north_star: "Reads shard across the 8 GPUs of one node with a final RCCL merge of per-GPU hash tables
// over xGMI."  The reference's entry point is one command (src/mains/main.cpp:404-507); so is this:
// tsx_hip_group_count_fastq_host() cuts the text into one shard of whole records per GPU (record rules of
// FastXReader.h:307-385: empty lines dropped, 4 -- FASTA: 2 -- non-empty lines per record), every GPU counts
// its shard into a table of its own (tsx_hip_count_fastq_host), and the tables are merged: entries grouped
// by owner = tsx_hip_owner(kmer, N) on the device (tsx_hip_partition_device), ONE all-to-all of the groups,
// the owner clears and re-inserts what it received (tsx_hip_add_kmers_device with counts).  Afterwards GPU r
// holds exactly the k-mers it owns, with their totals; lookups go to the owner.
//
// Everything here sits ABOVE the C ABI of include/tsxcount_hip.h (plain pointers, device buffers from the HIP
// runtime).  The collective is behind a two-function interface with two implementations:
//   RcclExchange  ncclCommInitAll + grouped ncclSend/ncclRecv on every GPU's stream (librccl is loaded with
//                 dlopen when a group asks for it: the library itself does not depend on it)
//   CopyExchange  device-to-device copies behind a barrier (several shards may then share one GPU): the test
//                 double for world sizes a one-GPU box cannot give RCCL
// The sizes of the groups travel through host memory (the ranks are threads of this process).
#include "../../include/tsxcount_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_multi_error;

// ---- a reusable barrier for the rank threads --------------------------------------------------------------
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        const unsigned long gen = gen_;
        if (++at_ == n_) { at_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen_ != gen; });
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, at_ = 0;
    unsigned long gen_ = 0;
};

// ---- RCCL through dlopen ----------------------------------------------------------------------------------
typedef void *ncclComm_t;
struct RcclApi {
    void *so = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (so) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (so) break;
        }
        if (!so) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(so, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(so, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(so, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(so, "ncclSend");
        Recv = (decltype(Recv))dlsym(so, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
const int NCCL_UINT64 = 5;   // ncclUint64 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5)

// What rank r offers to the exchange: `send` (device, words of 8 bytes) grouped by destination rank, send_off[p] ..
// send_off[p + 1] for rank p; it receives into `recv`, recv_off[p] .. from rank p.
struct Slot {
    const uint64_t *send = nullptr;
    uint64_t *recv = nullptr;
    std::vector<size_t> send_off, recv_off;   // n + 1 entries each
    std::vector<size_t> send_len;             // optional: words for peer p (then send_off[p] is only where they start)
    int device = 0;
    hipStream_t stream = nullptr;
};

class Exchange {
public:
    virtual ~Exchange() {}
    // called by every rank's thread with its own slot filled in; returns 0 or a TSX_HIP_E* code
    virtual int all_to_all(int rank, std::vector<Slot> &slots, Barrier &bar) = 0;
    virtual const char *name() const = 0;
};

class CopyExchange : public Exchange {
public:
    int all_to_all(int rank, std::vector<Slot> &slots, Barrier &bar) override {
        bar.wait();   // every rank's slot is complete and its send buffer written
        Slot &me = slots[rank];
        int rc = TSX_HIP_OK;
        for (size_t p = 0; p < slots.size() && rc == TSX_HIP_OK; ++p) {
            const Slot &src = slots[p];
            const size_t words = src.send_len.empty() ? src.send_off[rank + 1] - src.send_off[rank] : src.send_len[rank];
            if (words != me.recv_off[p + 1] - me.recv_off[p]) { g_multi_error = "exchange: split sizes disagree"; rc = TSX_HIP_EINVAL; break; }
            if (!words) continue;
            hipError_t e = (src.device == me.device)
                               ? hipMemcpyAsync(me.recv + me.recv_off[p], src.send + src.send_off[rank], words * 8, hipMemcpyDeviceToDevice, me.stream)
                               : hipMemcpyPeerAsync(me.recv + me.recv_off[p], me.device, src.send + src.send_off[rank], src.device, words * 8, me.stream);
            if (e != hipSuccess) { g_multi_error = std::string("exchange copy: ") + hipGetErrorString(e); rc = TSX_HIP_EHIP; }
        }
        if (rc == TSX_HIP_OK && hipStreamSynchronize(me.stream) != hipSuccess) rc = TSX_HIP_EHIP;
        bar.wait();   // nobody frees a send buffer a peer is still reading
        return rc;
    }
    const char *name() const override { return "copy"; }
};

class RcclExchange : public Exchange {
public:
    RcclExchange(RcclApi *api, std::vector<ncclComm_t> comms) : api_(api), comms_(std::move(comms)) {}
    ~RcclExchange() override { for (ncclComm_t c : comms_) if (c) api_->CommDestroy(c); }
    int all_to_all(int rank, std::vector<Slot> &slots, Barrier &bar) override {
        bar.wait();
        Slot &me = slots[rank];
        const int n = (int)slots.size();
        int rc = api_->GroupStart();
        for (int p = 0; p < n && rc == 0; ++p) {
            const size_t sw = me.send_len.empty() ? me.send_off[p + 1] - me.send_off[p] : me.send_len[p];
            const size_t rw = me.recv_off[p + 1] - me.recv_off[p];
            if (sw) rc = api_->Send(me.send + me.send_off[p], sw, NCCL_UINT64, p, comms_[rank], me.stream);
            if (rc == 0 && rw) rc = api_->Recv(me.recv + me.recv_off[p], rw, NCCL_UINT64, p, comms_[rank], me.stream);
        }
        const int rc2 = api_->GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) {
            g_multi_error = std::string("RCCL: ") + (api_->GetErrorString ? api_->GetErrorString(rc) : "error");
            bar.wait();
            return TSX_HIP_EHIP;
        }
        const hipError_t e = hipStreamSynchronize(me.stream);
        bar.wait();
        if (e != hipSuccess) { g_multi_error = std::string("RCCL exchange: ") + hipGetErrorString(e); return TSX_HIP_EHIP; }
        return TSX_HIP_OK;
    }
    const char *name() const override { return "rccl"; }
private:
    RcclApi *api_;
    std::vector<ncclComm_t> comms_;
};

RcclApi g_rccl;

// Shards of whole records: cut i is the first record boundary at or behind byte i * n / parts.  A line counts when
// it is not empty (FastXReader.h:365-370); its terminator at p counts iff p > 0 and text[p - 1] != '\n'.
std::vector<size_t> cut_records(const char *text, size_t n, int parts, int lines_per_record) {
    std::vector<size_t> target(parts + 1), cuts(parts + 1, n);
    for (int i = 0; i <= parts; ++i) target[i] = (size_t)((unsigned __int128)n * i / parts);
    std::vector<size_t> cnt(parts, 0);
    auto counted = [&](size_t p) { return text[p] == '\n' && p > 0 && text[p - 1] != '\n'; };
    {
        std::vector<std::thread> th;
        for (int i = 0; i < parts; ++i)
            th.emplace_back([&, i] {
                size_t c = 0;
                const char *q = text + target[i], *end = text + target[i + 1];
                while (q < end && (q = (const char *)memchr(q, '\n', end - q)) != nullptr) { c += counted(q - text) ? 1 : 0; ++q; }
                cnt[i] = c;
            });
        for (auto &t : th) t.join();
    }
    cuts[0] = 0;
    size_t before = 0;
    for (int i = 1; i < parts; ++i) {
        before += cnt[i - 1];
        size_t c = before, p = target[i];
        cuts[i] = n;
        // a target that sits exactly on a record boundary (right behind a counted terminator, count a multiple of L)
        if (c % lines_per_record == 0 && p > 0 && counted(p - 1)) { cuts[i] = p; continue; }
        for (; p < n; ++p)
            if (counted(p) && ++c % lines_per_record == 0) { cuts[i] = p + 1; break; }
    }
    for (int i = 1; i <= parts; ++i) cuts[i] = std::max(cuts[i], cuts[i - 1]);
    cuts[parts] = n;
    return cuts;
}

}  // namespace

struct tsx_hip_group {
    int n = 0;
    std::vector<tsx_hip_map *> maps;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    Exchange *xch = nullptr;
    Barrier *bar = nullptr;
    int lines_per_record = 4;
    int key_limbs = 1;
    int k = 0;
    int exchange = 0;                 // 0: per-GPU tables merged after the count; 1: minimizer exchange (tsx_minimizer.h)
    uint64_t exchanged_entries = 0;   // entries that changed GPU in the last merge
    double last_merge_ms = 0;
};

extern "C" const char *tsx_hip_group_last_error(void) { return g_multi_error.c_str(); }

extern "C" void tsx_hip_group_destroy(tsx_hip_group *g) {
    if (!g) return;
    for (int r = 0; r < g->n; ++r) {
        if (r < (int)g->streams.size() && g->streams[r]) { (void)hipSetDevice(g->devices[r]); (void)hipStreamDestroy(g->streams[r]); }
        if (r < (int)g->maps.size()) tsx_hip_destroy(g->maps[r]);
    }
    delete g->xch;
    delete g->bar;
    delete g;
}

extern "C" int tsx_hip_group_create(tsx_hip_group **out, int ngpus, const int *devices, int k, int l, int storagebits,
                                    int overflow_l, uint64_t hash_seed, int comm) {
    if (!out || ngpus < 1 || ngpus > 64 || (comm != 0 && comm != 1)) return TSX_HIP_EINVAL;
    *out = nullptr;
    const int ndev = tsx_hip_device_count();
    if (ndev <= 0) { g_multi_error = "no HIP device (the HIP path has no CPU fallback)"; return TSX_HIP_ENODEVICE; }
    tsx_hip_group *g = new tsx_hip_group();
    g->n = ngpus;
    for (int r = 0; r < ngpus; ++r) g->devices.push_back(devices ? devices[r] : r);
    for (int r = 0; r < ngpus; ++r)
        if (g->devices[r] < 0 || g->devices[r] >= ndev) {
            g_multi_error = "device " + std::to_string(g->devices[r]) + " of a " + std::to_string(ndev) + "-GPU node";
            delete g;
            return TSX_HIP_ENODEVICE;
        }
    if (comm == 0) {   // RCCL: one communicator per GPU, all made by this thread
        std::vector<int> sorted(g->devices);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
            g_multi_error = "RCCL needs one GPU per rank (the same device was given twice)";
            delete g;
            return TSX_HIP_EINVAL;
        }
        if (!g_rccl.load()) { g_multi_error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); delete g; return TSX_HIP_EHIP; }
        std::vector<ncclComm_t> comms(ngpus, nullptr);
        const int rc = g_rccl.CommInitAll(comms.data(), ngpus, g->devices.data());
        if (rc != 0) {
            g_multi_error = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
            delete g;
            return TSX_HIP_EHIP;
        }
        g->xch = new RcclExchange(&g_rccl, comms);
    } else {
        g->xch = new CopyExchange();
    }
    g->bar = new Barrier(ngpus);
    g->key_limbs = tsx_hip_key_limbs(k);
    g->k = k;
    for (int r = 0; r < ngpus; ++r) {
        tsx_hip_map *m = nullptr;
        const int rc = tsx_hip_create(&m, k, l, storagebits, overflow_l, hash_seed, g->devices[r]);
        if (rc != TSX_HIP_OK) { g_multi_error = tsx_hip_last_error(); tsx_hip_group_destroy(g); return rc; }
        g->maps.push_back(m);
        hipStream_t st = nullptr;
        if (hipSetDevice(g->devices[r]) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            g_multi_error = "hipStreamCreate failed";
            tsx_hip_group_destroy(g);
            return TSX_HIP_EHIP;
        }
        g->streams.push_back(st);
    }
    *out = g;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_group_size(const tsx_hip_group *g) { return g ? g->n : 0; }
extern "C" tsx_hip_map *tsx_hip_group_map(tsx_hip_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->maps[rank] : nullptr; }
extern "C" const char *tsx_hip_group_comm_name(const tsx_hip_group *g) { return (g && g->xch) ? g->xch->name() : ""; }

extern "C" int tsx_hip_group_set_record_lines(tsx_hip_group *g, int lines) {
    if (!g || (lines != 2 && lines != 4)) return TSX_HIP_EINVAL;
    for (tsx_hip_map *m : g->maps) { const int rc = tsx_hip_set_record_lines(m, lines); if (rc != TSX_HIP_OK) return rc; }
    g->lines_per_record = lines;
    return TSX_HIP_OK;
}

extern "C" int tsx_hip_group_clear(tsx_hip_group *g) {
    if (!g) return TSX_HIP_EINVAL;
    for (tsx_hip_map *m : g->maps) { const int rc = tsx_hip_clear(m); if (rc != TSX_HIP_OK) return rc; }
    return TSX_HIP_OK;
}

// Runs fn(rank) on one thread per rank; a rank that fails keeps taking part in the barriers of the exchange (its
// groups are then empty), all return together, the first failure is reported.
template <typename F>
static int run_ranks(tsx_hip_group *g, F fn) {
    std::vector<int> rcs(g->n, TSX_HIP_OK);
    std::vector<std::string> errs(g->n);
    std::vector<std::thread> th;
    for (int r = 0; r < g->n; ++r)
        th.emplace_back([&, r] {
            rcs[r] = fn(r);
            if (rcs[r] != TSX_HIP_OK) errs[r] = g_multi_error.empty() ? tsx_hip_last_error() : g_multi_error;
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < g->n; ++r)
        if (rcs[r] != TSX_HIP_OK) { g_multi_error = "rank " + std::to_string(r) + ": " + errs[r]; return rcs[r]; }
    return TSX_HIP_OK;
}

// The merge of the per-GPU tables ("table merge"); every rank's thread runs it.
static int merge_rank(tsx_hip_group *g, int r, std::vector<Slot> &slots, std::vector<Slot> &cslots,
                      std::vector<std::vector<unsigned long long>> &seg, std::vector<int> &ok, int rc_in) {
    const int n = g->n, wk = g->key_limbs;
    tsx_hip_map *m = g->maps[r];
    int rc = rc_in;
    (void)hipSetDevice(g->devices[r]);
    hipStream_t st = g->streams[r];
    tsx_hip_stats s;
    memset(&s, 0, sizeof s);
    if (rc == TSX_HIP_OK) rc = tsx_hip_get_stats(m, &s);
    const size_t mine = (rc == TSX_HIP_OK) ? (size_t)s.distinct : 0;
    uint64_t *d_k = nullptr, *d_c = nullptr, *d_rk = nullptr, *d_rc = nullptr;
    unsigned long long *d_seg = nullptr;
    auto dmalloc = [&](void **p, size_t bytes) {
        if (rc != TSX_HIP_OK) return;
        if (hipMalloc(p, std::max<size_t>(bytes, 64)) != hipSuccess) { g_multi_error = "hipMalloc of a merge buffer failed"; rc = TSX_HIP_ENOMEM; }
    };
    dmalloc((void **)&d_k, mine * wk * 8);
    dmalloc((void **)&d_c, mine * 8);
    dmalloc((void **)&d_seg, (size_t)n * 8);
    seg[r].assign(n, 0);
    if (rc == TSX_HIP_OK && mine) {
        rc = tsx_hip_partition_device(m, n, d_k, d_c, mine, d_seg, st);   // waits for its own stream
        if (rc == TSX_HIP_OK && hipMemcpy(seg[r].data(), d_seg, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = TSX_HIP_EHIP;
    }
    if (rc != TSX_HIP_OK) seg[r].assign(n, 0);   // a failed rank offers nothing and still takes part
    g->bar->wait();                              // every rank's group sizes are on the table
    Slot &ks = slots[r], &cs = cslots[r];
    ks.device = cs.device = g->devices[r];
    ks.stream = cs.stream = st;
    ks.send_off.assign(n + 1, 0); ks.recv_off.assign(n + 1, 0);
    cs.send_off.assign(n + 1, 0); cs.recv_off.assign(n + 1, 0);
    for (int p = 0; p < n; ++p) {
        cs.send_off[p + 1] = cs.send_off[p] + (size_t)seg[r][p];
        cs.recv_off[p + 1] = cs.recv_off[p] + (size_t)seg[p][r];
        ks.send_off[p + 1] = cs.send_off[p + 1] * wk;
        ks.recv_off[p + 1] = cs.recv_off[p + 1] * wk;
    }
    const size_t got = cs.recv_off[n];
    dmalloc((void **)&d_rk, got * wk * 8);
    dmalloc((void **)&d_rc, got * 8);
    ks.send = d_k; ks.recv = d_rk; cs.send = d_c; cs.recv = d_rc;
    // a rank that cannot send or receive must not leave its peers inside the collective: agree first
    ok[r] = (rc == TSX_HIP_OK) ? 1 : 0;
    g->bar->wait();
    bool all_ok = true;
    for (int p = 0; p < n; ++p) all_ok = all_ok && ok[p];
    if (all_ok) {
        rc = g->xch->all_to_all(r, slots, *g->bar);
        const int rc2 = g->xch->all_to_all(r, cslots, *g->bar);   // (taken even after a failure: the peers are in it)
        if (rc == TSX_HIP_OK) rc = rc2;
    } else if (rc == TSX_HIP_OK) {
        g_multi_error = "another rank failed before the exchange";
        rc = TSX_HIP_EHIP;
    }
    if (rc == TSX_HIP_OK) rc = tsx_hip_clear(m);
    if (rc == TSX_HIP_OK && got) rc = tsx_hip_add_kmers_device(m, d_rk, d_rc, got, st);
    if (rc == TSX_HIP_OK && hipStreamSynchronize(st) != hipSuccess) rc = TSX_HIP_EHIP;
    if (rc == TSX_HIP_OK) rc = tsx_hip_sync(m);
    (void)hipFree(d_k); (void)hipFree(d_c); (void)hipFree(d_seg); (void)hipFree(d_rk); (void)hipFree(d_rc);
    if (r == 0) {
        uint64_t moved = 0;
        for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) if (a != b) moved += seg[a][b];
        g->exchanged_entries = moved;
    }
    return rc;
}

// ---- the minimizer exchange from C++ (the Python form is distributed.MinimizerCounter) ----------
// Every GPU describes its record shard once, splits the descriptions owner by owner in `parts` shares, the lists of a share
// travel through the group's exchange, every GPU walks what it was sent into a table of its own; ONE level 2 + build at the
// end, the homopolymer totals of all GPUs on their owners.  Nothing is merged afterwards: the tables are disjoint.
extern "C" int tsx_hip_group_set_exchange(tsx_hip_group *g, int mode) {
    if (!g || (mode != 0 && mode != 1)) return TSX_HIP_EINVAL;
    if (mode == 1) {
        if (g->n > 16) { g_multi_error = "minimizer exchange: at most 16 GPUs"; return TSX_HIP_EINVAL; }
        for (tsx_hip_map *m : g->maps)
            if (!tsx_hip_mini_supported(m)) { g_multi_error = "minimizer exchange: 20 <= k <= 32 and a table split by two radix levels"; return TSX_HIP_EINVAL; }
    }
    g->exchange = mode;
    return TSX_HIP_OK;
}
extern "C" int tsx_hip_group_exchange(const tsx_hip_group *g) { return g ? g->exchange : 0; }

namespace {
struct MiniShared {
    std::vector<std::vector<unsigned long long>> cnt;   // [rank][owner | 4 homopolymer totals] of the current share
    std::vector<std::vector<unsigned long long>> hom;   // [rank][4] over all shares
    std::vector<unsigned long long> described, walked;
    std::vector<int> ok;
    std::vector<Slot> slots;
    explicit MiniShared(int n) : cnt(n, std::vector<unsigned long long>(n + 4, 0)), hom(n, std::vector<unsigned long long>(4, 0)),
                                 described(n, 0), walked(n, 0), ok(n, 0), slots(n) {}
};
constexpr size_t MINI_PIECE = (size_t)2 << 30;     // bytes of text described at once
constexpr size_t MINI_MIN_SHARE = (size_t)32 << 20;
}  // namespace

static int mini_rank(tsx_hip_group *g, int r, const char *text, size_t len, size_t max_len, MiniShared &sh) {
    const int n = g->n;
    tsx_hip_map *m = g->maps[r];
    (void)hipSetDevice(g->devices[r]);
    hipStream_t st = g->streams[r];
    int rc = TSX_HIP_OK;
    // the same rounds on every rank, whatever its own shard: pieces of the longest shard, shares of a piece
    const size_t piece_bytes = std::min(MINI_PIECE, std::max<size_t>(4096, (max_len + 4095) & ~(size_t)4095));
    const uint32_t pieces = (uint32_t)std::max<size_t>(1, (max_len + piece_bytes - 1) / piece_bytes);
    const uint32_t parts = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, piece_bytes / MINI_MIN_SHARE));
    const uint32_t rounds = pieces * parts;
    size_t cap = 0;
    (void)tsx_hip_mini_part_capacity(m, piece_bytes + 256, parts, &cap);
    uint8_t *d_text = nullptr;
    uint64_t *d_lists = nullptr, *d_recv = nullptr;
    unsigned long long *d_cnt = nullptr, *d_emit = nullptr;
    size_t recv_words = 0;
    auto dmalloc = [&](void **p, size_t bytes) {
        if (rc != TSX_HIP_OK) return;
        if (hipMalloc(p, std::max<size_t>(bytes, 64)) != hipSuccess) { g_multi_error = "hipMalloc of an exchange buffer failed"; rc = TSX_HIP_ENOMEM; }
    };
    dmalloc((void **)&d_text, len + 256);
    dmalloc((void **)&d_lists, (size_t)n * cap * 16);
    dmalloc((void **)&d_cnt, ((size_t)n + 4) * 8);
    dmalloc((void **)&d_emit, 16);
    if (rc == TSX_HIP_OK && (hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, st) != hipSuccess ||
                             hipMemsetAsync(d_text + len, '\n', 256, st) != hipSuccess || hipMemsetAsync(d_emit, 0, 16, st) != hipSuccess))
        rc = TSX_HIP_EHIP;
    size_t est_total = 0, got_total = 0;
    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t piece = round / parts, part = round % parts;
        std::vector<unsigned long long> &mine = sh.cnt[r];
        std::fill(mine.begin(), mine.end(), 0ULL);
        if (rc == TSX_HIP_OK && part == 0) {
            const size_t off = std::min((size_t)piece * piece_bytes, len & ~(size_t)15);
            const size_t ln = (size_t)piece * piece_bytes < len ? std::min(piece_bytes, len - off) : 0;
            rc = tsx_hip_mini_describe_device(m, d_text, len, off, ln, d_emit, st);
        }
        if (rc == TSX_HIP_OK) rc = tsx_hip_mini_split_device(m, part, parts, n, d_lists, cap, d_cnt, st);
        if (rc == TSX_HIP_OK && (hipMemcpyAsync(mine.data(), d_cnt, ((size_t)n + 4) * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                                 hipStreamSynchronize(st) != hipSuccess))
            rc = TSX_HIP_EHIP;
        if (rc != TSX_HIP_OK) std::fill(mine.begin(), mine.end(), 0ULL);   // a failed rank offers nothing and still takes part
        for (int o = 0; o < n; ++o) mine[o] = std::min<unsigned long long>(mine[o], cap);
        for (int b = 0; b < 4; ++b) sh.hom[r][b] += mine[n + b];
        g->bar->wait();                               // every rank's list sizes are on the table
        Slot &sl = sh.slots[r];
        sl.device = g->devices[r]; sl.stream = st;
        sl.send_off.assign(n + 1, 0); sl.recv_off.assign(n + 1, 0); sl.send_len.assign(n, 0);
        for (int p = 0; p < n; ++p) {
            sl.send_off[p] = (size_t)p * cap * 2;
            sl.send_len[p] = (size_t)sh.cnt[r][p] * 2;
            sl.recv_off[p + 1] = sl.recv_off[p] + (size_t)sh.cnt[p][r] * 2;
        }
        const size_t got = sl.recv_off[n];            // words: two per description
        if (rc == TSX_HIP_OK && got > recv_words) {
            if (d_recv) { (void)hipStreamSynchronize(st); (void)hipFree(d_recv); d_recv = nullptr; }
            recv_words = got + got / 4 + 4096;
            dmalloc((void **)&d_recv, recv_words * 8);
        }
        sl.send = d_lists; sl.recv = d_recv;
        sh.ok[r] = (rc == TSX_HIP_OK) ? 1 : 0;
        g->bar->wait();
        bool all_ok = true;
        for (int p = 0; p < n; ++p) all_ok = all_ok && sh.ok[p];
        if (!all_ok) {
            if (rc == TSX_HIP_OK) { g_multi_error = "another rank failed before the exchange"; rc = TSX_HIP_EHIP; }
            break;                                    // every rank sees the same flags: all leave together
        }
        rc = g->xch->all_to_all(r, sh.slots, *g->bar);
        if (rc == TSX_HIP_OK) {
            if (round == 0) est_total = (size_t)((double)std::max<size_t>(got / 2, 4096) * 10.0 * rounds * 1.2) + 65536;
            rc = tsx_hip_shard_walk_device(m, d_recv, got / 2, 2, round, rounds, est_total, d_emit + 1, st);
            got_total += got / 2;
        }
        // (a failure from here on is reported at the next round's agreement, or after the last one)
    }
    if (rc == TSX_HIP_OK && got_total) rc = tsx_hip_shard_build_l1_device(m, st);
    unsigned long long em[2] = {0, 0};
    if (rc == TSX_HIP_OK && (hipMemcpyAsync(em, d_emit, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
        rc = TSX_HIP_EHIP;
    sh.described[r] = em[0]; sh.walked[r] = em[1];
    sh.ok[r] = (rc == TSX_HIP_OK) ? 1 : 0;
    g->bar->wait();                                   // totals of all ranks
    bool all_ok = true;
    for (int p = 0; p < n; ++p) all_ok = all_ok && sh.ok[p];
    if (all_ok) {
        unsigned long long described = 0, walked = 0, hom[4] = {0, 0, 0, 0};
        for (int p = 0; p < n; ++p) { described += sh.described[p]; walked += sh.walked[p]; for (int b = 0; b < 4; ++b) hom[b] += sh.hom[p][b]; }
        if (described != walked + hom[0] + hom[1] + hom[2] + hom[3]) {
            g_multi_error = "minimizer exchange: k-mer occurrences described != walked + homopolymers";
            rc = TSX_HIP_EHIP;
        }
        for (int b = 0; b < 4 && rc == TSX_HIP_OK; ++b) {   // the homopolymer k-mers, on their owners, with the totals of all ranks
            if (!hom[b]) continue;
            uint64_t kmer = 0, cnt1 = hom[b];
            for (int i = 0; i < g->k; ++i) kmer |= (uint64_t)b << (2 * i);
            uint32_t owner = 0;
            rc = tsx_hip_mini_owner_host(g->k, n, &kmer, 1, &owner);
            if (rc == TSX_HIP_OK && (int)owner == r) rc = tsx_hip_add_kmers_host(m, &kmer, &cnt1, 1);
        }
    } else if (rc == TSX_HIP_OK) {
        g_multi_error = "another rank failed";
        rc = TSX_HIP_EHIP;
    }
    if (rc == TSX_HIP_OK) rc = tsx_hip_sync(m);
    (void)hipStreamSynchronize(st);
    (void)hipFree(d_text); (void)hipFree(d_lists); (void)hipFree(d_recv); (void)hipFree(d_cnt); (void)hipFree(d_emit);
    if (r == 0) {
        uint64_t moved = 0;   // (of the last share only: the sizes of earlier shares are gone)
        for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) if (a != b) moved += sh.cnt[a][b];
        g->exchanged_entries = moved;
    }
    return rc;
}

// The tables hold per-GPU counts of the reads each GPU saw: merge them (any number of counts before one merge).
extern "C" int tsx_hip_group_merge(tsx_hip_group *g) {
    if (!g) return TSX_HIP_EINVAL;
    if (g->exchange == 1) return TSX_HIP_OK;   // minimizer exchange: the tables are disjoint as they are
    // (a group of one takes the same route: its entries travel from rank 0 to rank 0 through the collective)
    std::vector<Slot> slots(g->n), cslots(g->n);
    std::vector<std::vector<unsigned long long>> seg(g->n);
    std::vector<int> ok(g->n, 0);
    return run_ranks(g, [&](int r) { return merge_rank(g, r, slots, cslots, seg, ok, TSX_HIP_OK); });
}

// countKMers for N GPUs (main.cpp:104-218 + the merge): text -> record shards -> per-GPU tables -> merged tables.
extern "C" int tsx_hip_group_count_fastq_host(tsx_hip_group *g, const char *text, size_t n) {
    if (!g || (!text && n)) return TSX_HIP_EINVAL;
    const std::vector<size_t> cuts = cut_records(text, n, g->n, g->lines_per_record);
    if (g->exchange == 1) {
        size_t max_len = 0;
        for (int r = 0; r < g->n; ++r) max_len = std::max(max_len, cuts[r + 1] - cuts[r]);
        MiniShared sh(g->n);
        return run_ranks(g, [&](int r) { return mini_rank(g, r, text + cuts[r], cuts[r + 1] - cuts[r], max_len, sh); });
    }
    std::vector<Slot> slots(g->n), cslots(g->n);
    std::vector<std::vector<unsigned long long>> seg(g->n);
    std::vector<int> ok(g->n, 0);
    return run_ranks(g, [&](int r) {
        (void)hipSetDevice(g->devices[r]);
        const int rc = tsx_hip_count_fastq_host(g->maps[r], text + cuts[r], cuts[r + 1] - cuts[r]);
        return merge_rank(g, r, slots, cslots, seg, ok, rc);
    });
}

// getKmerCount(kmer) for n k-mers: every k-mer is asked of the GPU that owns it.
extern "C" int tsx_hip_group_get_counts_host(tsx_hip_group *g, const uint64_t *kmers, size_t n, uint64_t *counts_out) {
    if (!g || ((!kmers || !counts_out) && n)) return TSX_HIP_EINVAL;
    if (g->n == 1) return tsx_hip_get_counts_host(g->maps[0], kmers, n, counts_out);
    const int wk = g->key_limbs;
    std::vector<std::vector<uint64_t>> q(g->n);
    std::vector<std::vector<size_t>> where(g->n);
    for (size_t i = 0; i < n; ++i) {
        int o;
        if (g->exchange == 1) {
            uint32_t ow = 0;
            if (tsx_hip_mini_owner_host(g->k, g->n, kmers + i * wk, 1, &ow) != TSX_HIP_OK) return TSX_HIP_EINVAL;
            o = (int)ow;
        } else {
            o = tsx_hip_owner_host(g->maps[0], kmers + i * wk, g->n);
        }
        if (o < 0 || o >= g->n) return TSX_HIP_EINVAL;
        q[o].insert(q[o].end(), kmers + i * wk, kmers + (i + 1) * wk);
        where[o].push_back(i);
    }
    return run_ranks(g, [&](int r) {
        if (where[r].empty()) return (int)TSX_HIP_OK;
        std::vector<uint64_t> got(where[r].size());
        const int rc = tsx_hip_get_counts_host(g->maps[r], q[r].data(), where[r].size(), got.data());
        if (rc == TSX_HIP_OK) for (size_t j = 0; j < got.size(); ++j) counts_out[where[r][j]] = got[j];
        return rc;
    });
}

// print_stats / getKmerCount() over the whole group: sums of the per-GPU counters (every k-mer lives on one GPU).
extern "C" int tsx_hip_group_get_stats(tsx_hip_group *g, tsx_hip_stats *out) {
    if (!g || !out) return TSX_HIP_EINVAL;
    memset(out, 0, sizeof *out);
    for (tsx_hip_map *m : g->maps) {
        tsx_hip_stats s;
        const int rc = tsx_hip_get_stats(m, &s);
        if (rc != TSX_HIP_OK) return rc;
        out->insert_failures += s.insert_failures; out->overflow_carries += s.overflow_carries;
        out->overflow_failures += s.overflow_failures; out->distinct += s.distinct; out->overflow_used += s.overflow_used;
        out->lock_timeouts += s.lock_timeouts; out->fallback_inserts += s.fallback_inserts; out->count_sum += s.count_sum;
    }
    // the merge re-inserts DISTINCT k-mers: occurrences = the sum of all counts (nothing was lost when no failure is set)
    out->kmers_added = out->count_sum;
    return TSX_HIP_OK;
}

extern "C" uint64_t tsx_hip_group_exchanged_entries(const tsx_hip_group *g) { return g ? g->exchanged_entries : 0; }

// The record cuts of a text (host logic, no GPU): cuts_out[0 .. parts] with cuts_out[0] = 0, cuts_out[parts] = n.
extern "C" int tsx_hip_cut_records_host(const char *text, size_t n, int parts, int lines_per_record, size_t *cuts_out) {
    if ((!text && n) || parts < 1 || parts > 4096 || (lines_per_record != 2 && lines_per_record != 4) || !cuts_out) return TSX_HIP_EINVAL;
    const std::vector<size_t> c = cut_records(text, n, parts, lines_per_record);
    for (int i = 0; i <= parts; ++i) cuts_out[i] = c[i];
    return TSX_HIP_OK;
}
