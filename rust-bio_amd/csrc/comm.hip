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
    uint64_t n_rec[kMaxRanks];      // records each rank brings to the current gather
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
    if (hipMalloc((void**)&c->d_counts, (size_t)world * 8 + 8) != hipSuccess) {
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
    int fd = -1;
    if (rank == 0) {
        shm_unlink(ctl.c_str());
        fd = shm_open(ctl.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(ShmCtrl)) != 0) {
            if (fd >= 0) close(fd);
            delete c;
            return BG_ERR_HIP;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        struct stat sb;
        while ((fd = shm_open(ctl.c_str(), O_RDWR, 0600)) < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(ShmCtrl)) {
            if (fd >= 0) close(fd);
            fd = -1;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                delete c;
                return BG_ERR_HIP;
            }
            usleep(1000);
        }
    }
    c->ctrl = (ShmCtrl*)mmap(nullptr, sizeof(ShmCtrl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->ctrl == MAP_FAILED) {
        delete c;
        return BG_ERR_HIP;
    }
    // (a fresh segment is zero-filled: counters, sense and counts start at 0)
    c->ctrl->arrived.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctrl->arrived.load(std::memory_order_acquire) < (uint32_t)world) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            munmap(c->ctrl, sizeof(ShmCtrl));
            delete c;
            return BG_ERR_HIP;
        }
    }
    *out = c;
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

extern "C" int bg_gather_records(bg_comm* c, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t* counts_out,
                                 void* stream) {
    if (!c || !rec_bytes || (n_local && !local) || !all) return BG_ERR_INVALID_ARG;
    const int W = c->world;
    std::vector<uint64_t> counts((size_t)W, 0);
    if (c->nccl) {
        Rccl& r = rccl();
        hipStream_t st = (hipStream_t)stream;
        BG_HIP(hipSetDevice(c->ctx->device));
        // the counts first (8 bytes per rank), then the records: one all-gather when the shards are equal
        BG_HIP(hipMemcpyAsync(c->d_counts + W, &n_local, 8, hipMemcpyHostToDevice, st));
        if (r.all_gather(c->d_counts + W, c->d_counts, 8, ncclUint8, c->nccl, st) != ncclSuccess) return BG_ERR_HIP;
        BG_HIP(hipMemcpyAsync(counts.data(), c->d_counts, (size_t)W * 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        bool equal = true;
        for (int k = 1; k < W; k++) equal = equal && counts[k] == counts[0];
        if (equal) {
            if (counts[0] && r.all_gather(local, all, (size_t)counts[0] * rec_bytes, ncclUint8, c->nccl, st) != ncclSuccess) return BG_ERR_HIP;
        } else {
            if (r.group_start() != ncclSuccess) return BG_ERR_HIP;
            uint64_t off = 0;
            for (int k = 0; k < W; k++) {
                const size_t bytes = (size_t)counts[k] * rec_bytes;
                if (bytes && r.broadcast(k == c->rank ? local : nullptr, (uint8_t*)all + off, bytes, ncclUint8, k, c->nccl, st) != ncclSuccess) {
                    r.group_end();
                    return BG_ERR_HIP;
                }
                off += bytes;
            }
            if (r.group_end() != ncclSuccess) return BG_ERR_HIP;
        }
    } else {
        if (!c->ctrl) return BG_ERR_INVALID_ARG;
        if (c->ctx) {
            BG_HIP(hipSetDevice(c->ctx->device));
            BG_HIP(hipStreamSynchronize((hipStream_t)stream));  // the records are results of work queued there
        }
        c->ctrl->n_rec[c->rank] = n_local;
        int rc = shm_barrier(c);
        if (rc) return rc;
        uint64_t total = 0, my_off = 0;
        for (int k = 0; k < W; k++) {
            counts[k] = c->ctrl->n_rec[k];
            if (k < c->rank) my_off += counts[k];
            total += counts[k];
        }
        const size_t bytes = std::max<size_t>((size_t)total * rec_bytes, 16);
        const std::string dn = c->name + "_d" + std::to_string(c->seq++);
        int fd = -1;
        if (c->rank == 0) {
            shm_unlink(dn.c_str());
            fd = shm_open(dn.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return BG_ERR_OOM;
        }
        if ((rc = shm_barrier(c))) return rc;  // the data segment exists
        if (c->rank != 0 && (fd = shm_open(dn.c_str(), O_RDWR, 0600)) < 0) return BG_ERR_HIP;
        uint8_t* data = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (data == MAP_FAILED) return BG_ERR_OOM;
        rc = copy_any(c->ctx, data + my_off * rec_bytes, local, (size_t)n_local * rec_bytes, hipMemcpyDeviceToHost);
        const int rc2 = shm_barrier(c);  // every slice is in place
        if (!rc && !rc2) rc = copy_any(c->ctx, all, data, (size_t)total * rec_bytes, hipMemcpyHostToDevice);
        const int rc3 = shm_barrier(c);  // everybody has read: the segment can go
        munmap(data, bytes);
        if (c->rank == 0) shm_unlink(dn.c_str());
        if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    }
    if (counts_out)
        for (int k = 0; k < W; k++) counts_out[k] = counts[k];
    return BG_OK;
}

// The same for records that sit in HOST memory (what the host-buffer entry points return: bg_align_batch's bg_alignment_t
// headers, bg_fm_backward_search_batch's arrays): staged through device scratch for an RCCL communicator (the collective
// itself runs over xGMI), straight through the shared segment for a host-staged one.  `all_cap`: records `all` can hold.
extern "C" int bg_gather_records_host(bg_comm* c, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t all_cap,
                                      uint64_t* counts_out) {
    if (!c || !rec_bytes || (n_local && !local) || !all) return BG_ERR_INVALID_ARG;
    if (!c->nccl) {
        bg_ctx* const keep = c->ctx;
        c->ctx = nullptr;  // host pointers: plain copies
        const int rc = bg_gather_records(c, local, n_local, rec_bytes, all, counts_out, nullptr);
        c->ctx = keep;
        return rc;
    }
    BG_HIP(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    void *d_loc = nullptr, *d_all = nullptr;
    BG_HIP(hipMalloc(&d_loc, std::max<size_t>((size_t)n_local * rec_bytes, 16)));
    if (hipMalloc(&d_all, std::max<size_t>((size_t)all_cap * rec_bytes, 16)) != hipSuccess) {
        hipFree(d_loc);
        return BG_ERR_OOM;
    }
    std::vector<uint64_t> counts((size_t)c->world, 0);
    int rc = BG_OK;
    if (n_local && hipMemcpyAsync(d_loc, local, (size_t)n_local * rec_bytes, hipMemcpyHostToDevice, st) != hipSuccess) rc = BG_ERR_HIP;
    if (!rc) rc = bg_gather_records(c, d_loc, n_local, rec_bytes, d_all, counts.data(), st);
    uint64_t total = 0;
    for (uint64_t k : counts) total += k;
    if (!rc && total > all_cap) rc = BG_ERR_OPS_CAP;
    if (!rc && total && hipMemcpyAsync(all, d_all, (size_t)total * rec_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) rc = BG_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = BG_ERR_HIP;
    hipFree(d_loc);
    hipFree(d_all);
    if (!rc && counts_out)
        for (int k = 0; k < c->world; k++) counts_out[k] = counts[k];
    return rc;
}
