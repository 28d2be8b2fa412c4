// Several GPUs behind the C ABI (BASELINE north_star: "query batches shard embarrassingly across the 8 GPUs of one node
// with a single RCCL all-gather over xGMI only to collect per-query scores/intervals").  One process and one bg_ctx per
// GPU; bg_shard_range / bg_shard_balanced cut the batch, every rank runs its slice through the *_dev entry points, and
// bg_gather_records brings the FIXED-SIZE result records of all ranks to every rank:
//   * RCCL flavour (bg_comm_init): ncclAllGather on the caller's stream when every rank holds the same number of records,
//     one grouped ncclBroadcast per rank otherwise (ragged shards; the counts travel first, as an 8-byte all-gather).
//     librccl.so is opened with dlopen when the first communicator is made — the library has no link-time dependency
//     on it, and a process that already carries an RCCL (PyTorch's) keeps using that one;
//   * host-staged flavour (bg_comm_init_host): ranks of one node meet in POSIX shared memory (a control segment with a
//     sense-reversing barrier and the per-rank counts, a data segment per gather).  It moves the records through host
//     memory — device pointers with a ctx, plain host pointers without — and exists for what RCCL cannot do: several
//     ranks on ONE GPU (tests/test_gpu_comm.py on the one-GPU test box) and no GPU at all (tests/test_comm_host.py).
// rust-bio has no counterpart (it is a single-process library): this is the engine's side of north_star's sharding.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <random>
#include <string>
#include <vector>

#include "bg_common.h"

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;      // (optional: bg_comm_world's read-back)
    decltype(&ncclCommUserRank) comm_rank = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.so) break;
        }
        if (!x.so) return x;
        x.get_id = (decltype(x.get_id))dlsym(x.so, "ncclGetUniqueId");
        x.init_rank = (decltype(x.init_rank))dlsym(x.so, "ncclCommInitRank");
        x.all_gather = (decltype(x.all_gather))dlsym(x.so, "ncclAllGather");
        x.broadcast = (decltype(x.broadcast))dlsym(x.so, "ncclBroadcast");
        x.group_start = (decltype(x.group_start))dlsym(x.so, "ncclGroupStart");
        x.group_end = (decltype(x.group_end))dlsym(x.so, "ncclGroupEnd");
        x.destroy = (decltype(x.destroy))dlsym(x.so, "ncclCommDestroy");
        x.comm_count = (decltype(x.comm_count))dlsym(x.so, "ncclCommCount");
        x.comm_rank = (decltype(x.comm_rank))dlsym(x.so, "ncclCommUserRank");
        x.ok = x.get_id && x.init_rank && x.all_gather && x.broadcast && x.group_start && x.group_end && x.destroy;
        return x;
    }();
    return r;
}

constexpr int kMaxRanks = 64;
struct ShmCtrl {
    std::atomic<uint32_t> arrived;  // ranks attached (init)
    std::atomic<uint32_t> count;    // barrier: arrivals of the current phase
    std::atomic<uint32_t> sense;    // barrier: flips when a phase completes
    std::atomic<int32_t> err;       // first error of the current gather: every rank fails together (rank 0 clears it afterwards)
    uint64_t n_rec[kMaxRanks];      // records each rank brings to the current gather
    uint64_t cap[kMaxRanks];        // records each rank's `all` can hold (~0: unchecked)
    // attach handshake: rank k writes a fresh nonce into hello[k]; the rank 0 of THIS run copies it into ack[k].  A segment
    // left behind by a crashed or earlier run of the same name never answers a fresh nonce, so a rank that opened it before
    // rank 0 replaced it notices (the name's inode changes) and attaches again
    std::atomic<uint64_t> hello[kMaxRanks], ack[kMaxRanks];
};

}  // namespace

struct bg_comm {
    bg_ctx* ctx = nullptr;  // null: host-staged over host pointers
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;  // RCCL flavour
    uint64_t* d_counts = nullptr;
    // host-staged flavour
    std::string name;
    ShmCtrl* ctrl = nullptr;
    uint32_t my_sense = 0;
    uint64_t seq = 0;
    // what the last gather did (bg_comm_world): 1 one ncclAllGather, 2 grouped broadcasts (ragged shards), 3 host-staged
    int last_path = 0;
    uint64_t n_gathers = 0;
};

namespace {

int shm_barrier(bg_comm* c) {
    ShmCtrl* s = c->ctrl;
    c->my_sense ^= 1u;
    if (s->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        s->count.store(0, std::memory_order_relaxed);
        s->sense.store(c->my_sense, std::memory_order_release);
        return BG_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->sense.load(std::memory_order_acquire) != c->my_sense) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return BG_ERR_HIP;  // a rank is gone
    }
    return BG_OK;
}

int copy_any(bg_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (!bytes) return BG_OK;
    if (!ctx) {
        memcpy(dst, src, bytes);
        return BG_OK;
    }
    BG_HIP(hipMemcpy(dst, src, bytes, kind));
    return BG_OK;
}

}  // namespace

extern "C" int bg_shard_range(uint64_t n_units, int rank, int world, uint64_t* lo, uint64_t* hi) {
    if (world < 1 || rank < 0 || rank >= world || !lo || !hi) return BG_ERR_INVALID_ARG;
    // [rank * N / W, (rank + 1) * N / W): contiguous, sizes differ by at most one (128-bit products: N up to 2^64 - 1)
    *lo = (uint64_t)((unsigned __int128)n_units * (unsigned)rank / (unsigned)world);
    *hi = (uint64_t)((unsigned __int128)n_units * (unsigned)(rank + 1) / (unsigned)world);
    return BG_OK;
}

extern "C" int bg_shard_balanced(const uint64_t* costs, uint64_t n, int world, uint64_t* bounds) {
    if (world < 1 || !bounds || (n && !costs)) return BG_ERR_INVALID_ARG;
    // contiguous ranges of (nearly) equal total cost — the sum of DP cells of mixed-length pairs, of pattern lengths:
    // boundary r is the first unit at which the running cost reaches r / world of the total
    unsigned __int128 total = 0;
    for (uint64_t i = 0; i < n; i++) total += costs[i];
    bounds[0] = 0;
    unsigned __int128 run = 0;
    uint64_t i = 0;
    for (int r = 1; r < world; r++) {
        const unsigned __int128 want = total * (unsigned)r;  // compare run * world >= total * r
        while (i < n && (run + costs[i]) * (unsigned)world <= want) run += costs[i++];
        bounds[r] = i;
    }
    bounds[world] = n;
    return BG_OK;
}

extern "C" int bg_comm_unique_id(uint8_t* id) {
    if (!id) return BG_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) <= BG_COMM_ID_BYTES, "id buffer");
    Rccl& r = rccl();
    if (!r.ok) return BG_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (r.get_id(&u) != ncclSuccess) return BG_ERR_HIP;
    memset(id, 0, BG_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return BG_OK;
}

extern "C" int bg_comm_init(bg_ctx* ctx, int rank, int world, const uint8_t* id, bg_comm** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return BG_ERR_INVALID_ARG;
    Rccl& r = rccl();
    if (!r.ok) return BG_ERR_UNSUPPORTED;
    BG_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    bg_comm* c = new bg_comm;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (r.init_rank(&c->nccl, world, u, rank) != ncclSuccess) {
        delete c;
        return BG_ERR_HIP;
    }
    if (hipMalloc((void**)&c->d_counts, (size_t)world * 16 + 16) != hipSuccess) {
        r.destroy(c->nccl);
        delete c;
        return BG_ERR_OOM;
    }
    *out = c;
    return BG_OK;
}

extern "C" int bg_comm_init_host(bg_ctx* ctx, int rank, int world, const char* name, bg_comm** out) {
    if (!name || !out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return BG_ERR_INVALID_ARG;
    bg_comm* c = new bg_comm;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->name = std::string("/bg_") + name;
    const std::string ctl = c->name + "_ctl";
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120); };
    auto fail = [&](int rc) {
        if (c->ctrl) munmap(c->ctrl, sizeof(ShmCtrl));
        delete c;
        return rc;
    };
    if (rank == 0) {
        shm_unlink(ctl.c_str());  // whatever an earlier run of this name left behind
        int fd = shm_open(ctl.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(ShmCtrl)) != 0) {
            if (fd >= 0) close(fd);
            return fail(BG_ERR_HIP);
        }
        void* m = mmap(nullptr, sizeof(ShmCtrl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) {
            shm_unlink(ctl.c_str());
            return fail(BG_ERR_HIP);
        }
        c->ctrl = (ShmCtrl*)m;  // (a fresh segment is zero-filled: counters, sense, error flag and nonces start at 0)
        c->ctrl->arrived.fetch_add(1, std::memory_order_acq_rel);
        while (c->ctrl->arrived.load(std::memory_order_acquire) < (uint32_t)world) {
            sched_yield();
            if (late()) {
                shm_unlink(ctl.c_str());
                return fail(BG_ERR_HIP);
            }
        }
        for (int k = 1; k < world; k++) c->ctrl->ack[k].store(c->ctrl->hello[k].load(std::memory_order_acquire), std::memory_order_release);
        *out = c;
        return BG_OK;
    }
    std::random_device rd;
    for (;;) {  // attach; again if the segment turns out to be a stale one
        int fd = -1;
        struct stat sb;
        while ((fd = shm_open(ctl.c_str(), O_RDWR, 0600)) < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(ShmCtrl)) {
            if (fd >= 0) close(fd);
            fd = -1;
            if (late()) return fail(BG_ERR_HIP);
            usleep(1000);
        }
        void* m = mmap(nullptr, sizeof(ShmCtrl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return fail(BG_ERR_HIP);
        c->ctrl = (ShmCtrl*)m;
        uint64_t nonce = ((uint64_t)rd() << 32) ^ rd() ^ ((uint64_t)getpid() << 20);
        nonce |= 1;  // never 0
        c->ctrl->hello[rank].store(nonce, std::memory_order_release);
        c->ctrl->arrived.fetch_add(1, std::memory_order_acq_rel);
        bool stale = false;
        for (uint32_t spin = 0; c->ctrl->ack[rank].load(std::memory_order_acquire) != nonce; spin++) {
            sched_yield();
            if (late()) return fail(BG_ERR_HIP);
            if ((spin & 1023) == 1023) {  // is the name still the segment this rank mapped?
                struct stat now;
                const int f2 = shm_open(ctl.c_str(), O_RDWR, 0600);
                const bool same = f2 >= 0 && fstat(f2, &now) == 0 && now.st_ino == sb.st_ino && now.st_dev == sb.st_dev;
                if (f2 >= 0) close(f2);
                if (!same) {
                    stale = true;
                    break;
                }
            }
        }
        if (!stale) break;
        munmap(c->ctrl, sizeof(ShmCtrl));
        c->ctrl = nullptr;
        usleep(1000);
    }
    *out = c;
    return BG_OK;
}

// What the communicator IS, read back from the library that runs it (not from what the caller asked for): info[0] = world
// as given to bg_comm_init*, info[1] = ncclCommCount of the RCCL communicator (0: host-staged — no RCCL involved),
// info[2] = ncclCommUserRank (-1: host-staged), info[3] = path of the last gather (0 none yet, 1 one ncclAllGather,
// 2 grouped ncclBroadcasts for ragged shards, 3 host-staged through shared memory), info[4] = gathers done so far.
extern "C" int bg_comm_world(bg_comm* c, int64_t* info /* 5 entries */) {
    if (!c || !info) return BG_ERR_INVALID_ARG;
    info[0] = c->world;
    info[1] = 0;
    info[2] = -1;
    if (c->nccl) {
        Rccl& r = rccl();
        int n = -1, me = -1;
        if (!r.comm_count || !r.comm_rank) return BG_ERR_UNSUPPORTED;
        if (r.comm_count(c->nccl, &n) != ncclSuccess || r.comm_rank(c->nccl, &me) != ncclSuccess) return BG_ERR_HIP;
        info[1] = n;
        info[2] = me;
    }
    info[3] = c->last_path;
    info[4] = (int64_t)c->n_gathers;
    return BG_OK;
}

extern "C" int bg_comm_free(bg_comm* c) {
    if (!c) return BG_OK;
    if (c->nccl) rccl().destroy(c->nccl);
    hipFree(c->d_counts);
    if (c->ctrl) {
        munmap(c->ctrl, sizeof(ShmCtrl));
        if (c->rank == 0) shm_unlink((c->name + "_ctl").c_str());
    }
    delete c;
    return BG_OK;
}

namespace {

constexpr uint64_t kNoCap = ~0ull;

// The counts of every rank — and the smallest `all` any rank brought: whether the records fit is decided BEFORE anything
// moves, and identically on every rank (a rank that left the collective alone would hang the others).
int rccl_counts(bg_comm* c, uint64_t n_local, uint64_t cap, std::vector<uint64_t>& counts, uint64_t& min_cap, hipStream_t st) {
    Rccl& r = rccl();
    const int W = c->world;
    uint64_t mine[2] = {n_local, cap};
    std::vector<uint64_t> got((size_t)2 * W, 0);
    BG_HIP(hipMemcpyAsync(c->d_counts + 2 * W, mine, 16, hipMemcpyHostToDevice, st));
    if (r.all_gather(c->d_counts + 2 * W, c->d_counts, 16, ncclUint8, c->nccl, st) != ncclSuccess) return BG_ERR_HIP;
    BG_HIP(hipMemcpyAsync(got.data(), c->d_counts, (size_t)W * 16, hipMemcpyDeviceToHost, st));
    BG_HIP(hipStreamSynchronize(st));
    min_cap = kNoCap;
    for (int k = 0; k < W; k++) {
        counts[k] = got[2 * k];
        min_cap = std::min(min_cap, got[2 * k + 1]);
    }
    return BG_OK;
}

int rccl_records(bg_comm* c, const void* local, uint32_t rec_bytes, void* all, const std::vector<uint64_t>& counts, hipStream_t st) {
    Rccl& r = rccl();
    const int W = c->world;
    bool equal = true;
    for (int k = 1; k < W; k++) equal = equal && counts[k] == counts[0];
    c->last_path = equal ? 1 : 2;
    c->n_gathers++;
    if (equal) {  // one all-gather when the shards are equal
        if (counts[0] && r.all_gather(local, all, (size_t)counts[0] * rec_bytes, ncclUint8, c->nccl, st) != ncclSuccess) return BG_ERR_HIP;
        return BG_OK;
    }
    if (r.group_start() != ncclSuccess) return BG_ERR_HIP;  // ragged: one grouped broadcast per rank
    uint64_t off = 0;
    for (int k = 0; k < W; k++) {
        const size_t bytes = (size_t)counts[k] * rec_bytes;
        if (bytes && r.broadcast(k == c->rank ? local : nullptr, (uint8_t*)all + off, bytes, ncclUint8, k, c->nccl, st) != ncclSuccess) {
            r.group_end();
            return BG_ERR_HIP;
        }
        off += bytes;
    }
    return r.group_end() != ncclSuccess ? BG_ERR_HIP : BG_OK;
}

// host-staged flavour.  Every rank passes the same barriers whatever fails locally: a failure is published in the control
// block's error flag and every rank returns it together, after the last barrier — nobody is left spinning, and the barrier
// state stays usable for the next call.  (A barrier that times out means a rank is gone: the communicator is dead then;
// the data segment is unmapped, closed and unlinked on the way out all the same.)
int shm_gather(bg_comm* c, bg_ctx* ctx, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t cap,
               std::vector<uint64_t>& counts) {
    ShmCtrl* s = c->ctrl;
    const int W = c->world;
    s->n_rec[c->rank] = n_local;
    s->cap[c->rank] = cap;
    int rc = shm_barrier(c);
    if (rc) return rc;
    uint64_t total = 0, my_off = 0, min_cap = kNoCap;
    for (int k = 0; k < W; k++) {
        counts[k] = s->n_rec[k];
        min_cap = std::min(min_cap, s->cap[k]);
        if (k < c->rank) my_off += counts[k];
        total += counts[k];
    }
    if (total > min_cap) {  // every rank sees the same numbers and leaves here; nothing was moved
        rc = shm_barrier(c);  // (the counts are read: the next call may overwrite them)
        return rc ? rc : BG_ERR_OPS_CAP;
    }
    const size_t bytes = std::max<size_t>((size_t)total * rec_bytes, 16);
    const std::string dn = c->name + "_d" + std::to_string(c->seq++);
    auto publish = [&](int e) {
        int32_t none = 0;
        if (e) s->err.compare_exchange_strong(none, e, std::memory_order_acq_rel);
    };
    int fd = -1;
    uint8_t* data = nullptr;
    auto cleanup = [&] {
        if (data) munmap(data, bytes);
        if (fd >= 0) close(fd);
        if (c->rank == 0) shm_unlink(dn.c_str());
    };
    if (c->rank == 0) {
        shm_unlink(dn.c_str());
        fd = shm_open(dn.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) publish(BG_ERR_OOM);
    }
    if ((rc = shm_barrier(c))) {  // the data segment exists (or the error flag says why not)
        cleanup();
        return rc;
    }
    if (!s->err.load(std::memory_order_acquire)) {
        if (c->rank != 0 && (fd = shm_open(dn.c_str(), O_RDWR, 0600)) < 0) publish(BG_ERR_HIP);
        if (fd >= 0) {
            void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED)
                publish(BG_ERR_OOM);
            else
                data = (uint8_t*)m;
        }
        if (data) publish(copy_any(ctx, data + my_off * rec_bytes, local, (size_t)n_local * rec_bytes, hipMemcpyDeviceToHost));
    }
    if ((rc = shm_barrier(c))) {  // every slice is in place
        cleanup();
        return rc;
    }
    int e = s->err.load(std::memory_order_acquire);
    if (!e && data) e = copy_any(ctx, all, data, (size_t)total * rec_bytes, hipMemcpyHostToDevice);
    rc = shm_barrier(c);  // everybody has read the segment and the flag: both can go
    cleanup();
    if (c->rank == 0) s->err.store(0, std::memory_order_release);  // (the others touch it again only behind the next call's first barrier)
    return rc ? rc : e;
}

int gather_any(bg_comm* c, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t cap, uint64_t* counts_out,
               hipStream_t st) {
    const int W = c->world;
    std::vector<uint64_t> counts((size_t)W, 0);
    int rc;
    if (c->nccl) {
        BG_HIP(hipSetDevice(c->ctx->device));
        uint64_t min_cap = kNoCap, total = 0;
        if ((rc = rccl_counts(c, n_local, cap, counts, min_cap, st))) return rc;
        for (uint64_t k : counts) total += k;
        if (total > min_cap) return BG_ERR_OPS_CAP;
        if ((rc = rccl_records(c, local, rec_bytes, all, counts, st))) return rc;
    } else {
        if (!c->ctrl) return BG_ERR_INVALID_ARG;
        if (c->ctx) {
            BG_HIP(hipSetDevice(c->ctx->device));
            BG_HIP(hipStreamSynchronize(st));  // the records are results of work queued there
        }
        if ((rc = shm_gather(c, c->ctx, local, n_local, rec_bytes, all, cap, counts))) return rc;
        c->last_path = 3;
        c->n_gathers++;
    }
    if (counts_out)
        for (int k = 0; k < W; k++) counts_out[k] = counts[k];
    return BG_OK;
}

}  // namespace

extern "C" int bg_gather_records(bg_comm* c, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t* counts_out,
                                 void* stream) {
    if (!c || !rec_bytes || (n_local && !local) || !all) return BG_ERR_INVALID_ARG;
    return gather_any(c, local, n_local, rec_bytes, all, kNoCap, counts_out, (hipStream_t)stream);
}

// The same with the size of `all` stated (records): BG_ERR_OPS_CAP on EVERY rank, before any record moves, when the ranks'
// records together exceed the smallest `all_cap` any rank passed
extern "C" int bg_gather_records_cap(bg_comm* c, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t all_cap,
                                     uint64_t* counts_out, void* stream) {
    if (!c || !rec_bytes || (n_local && !local) || !all) return BG_ERR_INVALID_ARG;
    return gather_any(c, local, n_local, rec_bytes, all, all_cap, counts_out, (hipStream_t)stream);
}

// The same for records that sit in HOST memory (what the host-buffer entry points return: bg_align_batch's bg_alignment_t
// headers, bg_fm_backward_search_batch's arrays): staged through device scratch for an RCCL communicator (the collective
// itself runs over xGMI), straight through the shared segment for a host-staged one.  `all_cap`: records `all` can hold —
// checked against the gathered counts before a single record is written (BG_ERR_OPS_CAP on every rank).
extern "C" int bg_gather_records_host(bg_comm* c, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t all_cap,
                                      uint64_t* counts_out) {
    if (!c || !rec_bytes || (n_local && !local) || !all) return BG_ERR_INVALID_ARG;
    const int W = c->world;
    std::vector<uint64_t> counts((size_t)W, 0);
    int rc = BG_OK;
    if (!c->nccl) {
        if (!c->ctrl) return BG_ERR_INVALID_ARG;
        rc = shm_gather(c, nullptr, local, n_local, rec_bytes, all, all_cap, counts);  // host pointers: plain copies
        c->last_path = 3;
        c->n_gathers++;
    } else {
        BG_HIP(hipSetDevice(c->ctx->device));
        hipStream_t st = c->ctx->stream;
        uint64_t min_cap = kNoCap, total = 0;
        // a rank that fails locally (the staging allocations) must not leave its peers alone in the collective: d_loc is
        // allocated before the counts travel and its outcome rides with them (a count of ~0 marks a failed rank); d_all,
        // which is sized from the counts, is agreed on in a second 8-byte exchange before a single record moves
        void *d_loc = nullptr, *d_all = nullptr;
        const bool loc_ok = hipMalloc(&d_loc, std::max<size_t>((size_t)n_local * rec_bytes, 16)) == hipSuccess;
        if ((rc = rccl_counts(c, loc_ok ? n_local : kNoCap, all_cap, counts, min_cap, st))) {
            hipFree(d_loc);
            return rc;
        }
        bool peers_ok = true;
        for (uint64_t k : counts) peers_ok = peers_ok && k != kNoCap;
        if (!peers_ok) {
            hipFree(d_loc);
            return BG_ERR_OOM;  // on every rank
        }
        for (uint64_t k : counts) total += k;
        if (total > min_cap) {
            hipFree(d_loc);
            return BG_ERR_OPS_CAP;
        }
        const bool all_ok = hipMalloc(&d_all, std::max<size_t>((size_t)total * rec_bytes, 16)) == hipSuccess;
        {
            std::vector<uint64_t> oks((size_t)W, 0);
            uint64_t dummy = kNoCap;
            if ((rc = rccl_counts(c, all_ok ? 1 : 0, kNoCap, oks, dummy, st))) {
                hipFree(d_loc);
                hipFree(d_all);
                return rc;
            }
            bool every = true;
            for (uint64_t k : oks) every = every && k == 1;
            if (!every) {
                hipFree(d_loc);
                hipFree(d_all);
                return BG_ERR_OOM;  // on every rank
            }
        }
        if (n_local && hipMemcpyAsync(d_loc, local, (size_t)n_local * rec_bytes, hipMemcpyHostToDevice, st) != hipSuccess) rc = BG_ERR_HIP;
        const int rc2 = rccl_records(c, d_loc, rec_bytes, d_all, counts, st);  // entered even after a failed upload: the peers are in it
        if (!rc) rc = rc2;
        if (!rc && total && hipMemcpyAsync(all, d_all, (size_t)total * rec_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) rc = BG_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = BG_ERR_HIP;
        hipFree(d_loc);
        hipFree(d_all);
    }
    if (!rc && counts_out)
        for (int k = 0; k < W; k++) counts_out[k] = counts[k];
    return rc;
}
