// Band construction on the device — `Band::create` (/root/reference/src/alignment/pairwise/banded.rs:1278-1367)
// for a sub-batch of pairs, so that the banded pipeline no longer waits for host threads:
//
//   B1 kmer_match_kernel   sparse::find_kmer_matches (sparse.rs:337-402): all exact k-mer matches of a pair,
//                          sorted by (x, y).  One block per pair: a chained hash table of y's k-mers in
//                          global scratch, every x position probes it, counts -> block scan -> fill.
//   B2 chain_kernel        sparse::sdpkpp (sparse.rs:188-295): the best chain under the affine gap
//                          penalty, with the reference's tie-breaking (derived Ord of PrevPtr and of the
//                          (score, index) dp tuples).  One wavefront per pair; the prefix-max Fenwick tree
//                          lives in LDS over rank-compressed y coordinates, its log(n) nodes are read /
//                          updated by different lanes at once, the event order is a two-pointer merge.
//   B3 band_kernel         Band::create_from_match_path (banded.rs:1330-1367) with set_boundaries /
//                          add_kmer / add_gap / add_entry.  Every one of them only lowers start[j] or
//                          raises end[j], so the order does not matter: lanes take path elements and
//                          apply them with atomicMin / atomicMax.
//   B4 band_rows_kernel    column ranges -> per-row column ranges, traceback offsets, cell count, flags
//                          (what banded_api.hip computes on the host for host-built bands).
//
// Pairs the device path does not cover (more matches than the LDS tree holds, overlong hash chains)
// are flagged and rebuilt by the host builder (band_host.cpp); results are identical either way —
// tests/test_gpu_banded.py compares the two builders band by band.
#include "banded_kernels.h"
#include "band_device.h"

namespace bgband_dev {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t kmer_hash(const uint8_t* s, uint32_t k) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (uint32_t t = 0; t < k; t++) h = (h ^ s[t]) * 0x100000001B3ull + 0x9E3779B97F4A7C15ull;
    return h;
}
__device__ __forceinline__ bool kmer_equal(const uint8_t* a, const uint8_t* b, uint32_t k) {
    for (uint32_t t = 0; t < k; t++)
        if (a[t] != b[t]) return false;
    return true;
}

// global -> LDS copy of a sequence by the whole block: 16-byte loads from the first 16-byte boundary of src on
__device__ __forceinline__ void stage_bytes(uint8_t* dst, const uint8_t* src, uint32_t len) {
    const uint32_t mis = (uint32_t)((uintptr_t)src & 15u);  // bytes up to the first 16-byte boundary of src
    const uint32_t headb = min(len, (16u - mis) & 15u);
    for (uint32_t i = threadIdx.x; i < headb; i += blockDim.x) dst[i] = src[i];
    const uint32_t nvec = (len - headb) / 16;
    const uint4* s4 = (const uint4*)(src + headb);
    for (uint32_t v = threadIdx.x; v < nvec; v += blockDim.x) {
        const uint4 q = s4[v];
        uint32_t* d = (uint32_t*)(dst + headb + 16 * v);  // dst + headb is only byte-aligned in general
        if ((headb & 3u) == 0) {
            d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        } else {
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            for (int t = 0; t < 16; t++) dst[headb + 16 * v + t] = (uint8_t)(w[t >> 2] >> (8 * (t & 3)));
        }
    }
    for (uint32_t i = headb + 16 * nvec + threadIdx.x; i < len; i += blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------- B1
// STAGED: both sequences are copied to LDS first (16-byte loads) and every k-mer is hashed / compared from there.
// With the sequences in global memory the hash loop (k is a run-time value) is a chain of k byte loads, each waited
// for before the next — 16 L1/L2 round trips per hash, twice per position plus once per candidate comparison: that,
// not the hash table, was where the kernel spent its time (18.6 ms per 16 384 x 10 kb pairs).
constexpr uint32_t kStageMaxBytes = 40960;  // x + y: two blocks per CU keep their sequences in LDS
template <bool STAGED>
__global__ __launch_bounds__(256) void kmer_match_kernel(const BandDevArgs a) {
    extern __shared__ __align__(16) uint8_t s_seq[];
    const uint32_t pair = blockIdx.x;
    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo), n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
    const uint8_t* x = a.x + xo;
    const uint8_t* y = a.y + yo;
    if (STAGED) {
        uint8_t* sx = s_seq;
        uint8_t* sy = s_seq + ((m + 15) & ~15u);
        stage_bytes(sx, x, m);
        stage_bytes(sy, y, n);
        __syncthreads();
        x = sx;
        y = sy;
    }
    const uint32_t k = a.k;
    BandDevPair* st = a.state + pair;
    uint32_t* head = a.head + (size_t)pair * a.table_size;
    uint32_t* next = a.next + (size_t)pair * a.max_n;
    uint64_t* hy = a.hy + (size_t)pair * a.max_n;
    uint32_t* mx = a.mx + (size_t)pair * a.cap_matches;
    uint32_t* my = a.my + (size_t)pair * a.cap_matches;
    __shared__ uint32_t s_flag;
    if (threadIdx.x == 0) s_flag = 0;
    const uint32_t nky = n >= k && k ? n - k + 1 : 0, nkx = m >= k && k ? m - k + 1 : 0;
    for (uint32_t i = threadIdx.x; i < a.table_size; i += blockDim.x) head[i] = kNone;
    __syncthreads();
    const uint32_t shift = 64 - a.table_bits;
    for (uint32_t i = threadIdx.x; i < nky; i += blockDim.x) {
        const uint64_t h = kmer_hash(y + i, k);
        hy[i] = h;
        next[i] = atomicExch(&head[(h * 0xD6E8FEB86659FD93ull) >> shift], i);
    }
    __syncthreads();
    // x positions in tiles of 256 (thread t probes position tile + t: coalesced, and the matches come out
    // in x order): count -> exclusive scan inside the tile -> fill at the running offset
    __shared__ uint32_t s_scan[256];
    uint32_t run = 0;
    bool over = (k == 0 || m + k >= (1u << 20));
    for (uint32_t p0 = 0; p0 < nkx && !over; p0 += blockDim.x) {
        const uint32_t p = p0 + threadIdx.x;
        uint64_t h = 0;
        uint32_t c = 0;
        if (p < nkx) {
            h = kmer_hash(x + p, k);
            for (uint32_t i = head[(h * 0xD6E8FEB86659FD93ull) >> shift]; i != kNone; i = next[i])
                if (hy[i] == h && kmer_equal(y + i, x + p, k)) c++;
        }
        // exclusive scan of the counts over the tile: inside a wavefront with lane shifts, one barrier for the four
        // wavefront totals (a 256-wide LDS scan costs sixteen barriers per tile, forty tiles per read)
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
            if ((threadIdx.x & 63) >= (uint32_t)o) incl += v;
        }
        // the in-place sort below is quadratic in c, and the chain kernel holds kMaxChainMatches
        if (c > kMaxMatchesPerKmer) s_flag = 1;
        if ((threadIdx.x & 63) == 63) s_scan[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t wbase = 0, tile_total = 0;
        for (uint32_t wv = 0; wv < (blockDim.x >> 6); wv++) {
            if (wv < (threadIdx.x >> 6)) wbase += s_scan[wv];
            tile_total += s_scan[wv];
        }
        const uint32_t off = run + wbase + incl - c;
        if (run + tile_total > a.cap_matches || run + tile_total > kMaxChainMatches || s_flag) {
            over = true;  // block-uniform: s_flag and the totals are read after the barrier
        } else if (c) {
            uint32_t w = 0;
            for (uint32_t i = head[(h * 0xD6E8FEB86659FD93ull) >> shift]; i != kNone; i = next[i])
                if (hy[i] == h && kmer_equal(y + i, x + p, k)) {  // insertion sort by y (chains are in arbitrary order)
                    uint32_t t = w++;
                    while (t > 0 && my[off + t - 1] > i) {
                        my[off + t] = my[off + t - 1];
                        t--;
                    }
                    my[off + t] = i;
                }
            for (uint32_t t = 0; t < c; t++) mx[off + t] = p;
        }
        run += tile_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st->n_matches = over ? 0 : run;
        st->flags = over ? BP_HOST_FALLBACK : BP_OK;
    }
}

// ------------------------------------------------------------------------------------------- B1 (LDS)
// The whole join of a pair in LDS: both sequences, a table of 8 192 first positions, the per-position chain links and
// a 16-bit tag per y k-mer — 20 + 32 + 20 + 20 KB for 10 kb reads, one block of 512 threads per CU.  kmer_match_kernel
// keeps table, links and hashes in global memory (a 128 KB table per pair: 28 GB of HBM traffic per 16 384 pairs for
// 164 MB of sequences, every probe a chain of dependent L2 / HBM round trips); it had to fit next to a running fill
// (20 KB of LDS), which the join no longer does (banded_api.hip).  Every thread owns a contiguous run of positions and
// rolls a polynomial hash along it (one byte in, one byte out per position); what is hashed how is ours to choose —
// candidates are compared byte for byte — and the matches come out exactly as kmer_match_kernel leaves them: in x
// order, the y positions of one x position ascending.
constexpr uint32_t kLdsJoinTable = 8192;  // (16 384 in round 3: 124 KB per 10 kb pair; with 8 192 slots the join is 91 KB and fits next to two K3i blocks)
constexpr uint32_t kLdsJoinThreads = 512;
constexpr uint32_t kRollBase = 0x9E3779B1u;
__host__ __device__ inline size_t lds_join_bytes(uint32_t m, uint32_t n) {
    return (size_t)((m + 15) & ~15u) + ((n + 15) & ~15u) + 4 * (size_t)kLdsJoinTable + 2 * (size_t)((n + 7) & ~7u) * 2 + 64;
}
__global__ __launch_bounds__(512) void kmer_match_lds_kernel(const BandDevArgs a) {
    extern __shared__ __align__(16) uint8_t s_seq[];
    // (no raised wave priority here: on its own stream the join has a whole fill to finish under, and at priority 3 it
    //  took the issue slots the fill needed)
    const uint32_t pair = blockIdx.x;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo), n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
    uint8_t* sx = s_seq;
    uint8_t* sy = sx + ((m + 15) & ~15u);
    uint32_t* head = (uint32_t*)(sy + ((n + 15) & ~15u));
    uint16_t* next = (uint16_t*)(head + kLdsJoinTable);
    uint16_t* tag = next + ((n + 7) & ~7u);
    __shared__ uint32_t s_wsum[kLdsJoinThreads / 64];
    __shared__ uint32_t s_flag;
    if (tid == 0) s_flag = 0;
    {
        stage_bytes(sx, a.x + xo, m);
        stage_bytes(sy, a.y + yo, n);
        for (uint32_t i = tid; i < kLdsJoinTable / 4; i += nth) ((uint4*)head)[i] = make_uint4(kNone, kNone, kNone, kNone);
    }
    const uint32_t k = a.k;
    BandDevPair* st = a.state + pair;
    uint32_t* mx = a.mx + (size_t)pair * a.cap_matches;
    uint32_t* my = a.my + (size_t)pair * a.cap_matches;
    const uint32_t nky = n >= k && k ? n - k + 1 : 0, nkx = m >= k && k ? m - k + 1 : 0;
    uint32_t bk1 = 1;  // kRollBase^(k-1)
    for (uint32_t t = 1; t < k; t++) bk1 *= kRollBase;
    auto first_hash = [&](const uint8_t* q) {
        uint32_t h = 0;
        for (uint32_t t = 0; t < k; t++) h = h * kRollBase + q[t];
        return h;
    };
    auto slot_of = [](uint32_t h) { return (h * 0x85EBCA6Bu) >> 19; };  // 13 bits
    static_assert(kLdsJoinTable == 1u << 13, "slot_of");
    auto tag_of = [](uint32_t h) { return (uint16_t)(h ^ (h >> 16)); };
    __syncthreads();
    {  // y positions into the table: a contiguous run per thread, the hash rolled along it
        const uint32_t per = (nky + nth - 1) / nth, i0 = min(nky, tid * per), i1 = min(nky, i0 + per);
        uint32_t h = i0 < i1 ? first_hash(sy + i0) : 0;
        for (uint32_t i = i0; i < i1; i++) {
            tag[i] = tag_of(h);
            next[i] = (uint16_t)atomicExch(&head[slot_of(h)], i);  // kNone truncates to 0xFFFF: no position is that large
            if (i + 1 < i1) h = (h - sy[i] * bk1) * kRollBase + sy[i + k];
        }
    }
    __syncthreads();
    const bool too_long = (k == 0 || m + k >= (1u << 20));
    const uint32_t per = (nkx + nth - 1) / nth, p0 = min(nkx, tid * per), p1 = min(nkx, p0 + per);
    // The first 16 bytes of the x k-mer ride along in four registers (one byte out, one byte in per position); a
    // candidate's first 16 bytes come as five aligned LDS words shifted into place; both are compared under the masks of
    // the bytes below k — no loop, no branch.  Bytes from the 17th on (k > 16) take the byte loop.
    uint32_t kmask[4];
#pragma unroll
    for (int j = 0; j < 4; j++) kmask[j] = k >= 4u * (j + 1) ? 0xFFFFFFFFu : (k > 4u * j ? (1u << (8 * (k - 4 * j))) - 1 : 0u);
    auto window_at = [&](uint32_t p, uint32_t* w) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            w[j] = (uint32_t)sx[p + 4 * j] | ((uint32_t)sx[p + 4 * j + 1] << 8) | ((uint32_t)sx[p + 4 * j + 2] << 16) | ((uint32_t)sx[p + 4 * j + 3] << 24);
    };
    auto window_roll = [&](uint32_t p, uint32_t* w) {  // p: the position being left
        const uint32_t in = sx[p + 16];
        w[0] = __builtin_amdgcn_alignbyte(w[1], w[0], 1);
        w[1] = __builtin_amdgcn_alignbyte(w[2], w[1], 1);
        w[2] = __builtin_amdgcn_alignbyte(w[3], w[2], 1);
        w[3] = __builtin_amdgcn_alignbyte(in, w[3], 1);
    };
    const uint32_t* sy32 = (const uint32_t*)sy;
    auto walk = [&](uint32_t p, uint32_t h, const uint32_t* w, auto&& hit) {
        const uint16_t tg = tag_of(h);
        for (uint32_t i = head[slot_of(h)] & 0xFFFFu; i != 0xFFFFu; i = next[i]) {
            if (tag[i] != tg) continue;
            const uint32_t q = i >> 2, sh = i & 3u;
            const uint32_t d0 = sy32[q], d1 = sy32[q + 1], d2 = sy32[q + 2], d3 = sy32[q + 3], d4 = sy32[q + 4];
            uint32_t diff = (__builtin_amdgcn_alignbyte(d1, d0, sh) ^ w[0]) & kmask[0];
            diff |= (__builtin_amdgcn_alignbyte(d2, d1, sh) ^ w[1]) & kmask[1];
            diff |= (__builtin_amdgcn_alignbyte(d3, d2, sh) ^ w[2]) & kmask[2];
            diff |= (__builtin_amdgcn_alignbyte(d4, d3, sh) ^ w[3]) & kmask[3];
            for (uint32_t t = 16; t < k; t++) diff |= (uint32_t)(sy[i + t] ^ sx[p + t]);
            if (diff == 0) hit(i);
        }
    };
    // pass 1: matches per thread (and the longest list of one x position)
    uint32_t mine = 0, longest = 0;
    uint32_t w[4] = {0, 0, 0, 0};
    if (!too_long && p0 < p1) {
        uint32_t h = first_hash(sx + p0);
        window_at(p0, w);
        for (uint32_t p = p0; p < p1; p++) {
            uint32_t c = 0;
            walk(p, h, w, [&](uint32_t) { c++; });
            mine += c;
            longest = max(longest, c);
            if (p + 1 < p1) {
                h = (h - sx[p] * bk1) * kRollBase + sx[p + k];
                window_roll(p, w);
            }
        }
    }
    // the in-place sort below is quadratic in the list length, and the chain kernel holds kMaxChainMatches
    if (longest > kMaxMatchesPerKmer) s_flag = 1;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
        if ((tid & 63) >= (uint32_t)o) incl += v;
    }
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (uint32_t wv = 0; wv < (nth >> 6); wv++) {
        if (wv < (tid >> 6)) wbase += s_wsum[wv];
        total += s_wsum[wv];
    }
    const bool over = too_long || total > a.cap_matches || total > kMaxChainMatches || s_flag;
    if (!over && mine) {  // pass 2: the same walk, now writing
        uint32_t off = wbase + incl - mine;
        uint32_t h = first_hash(sx + p0);
        window_at(p0, w);
        for (uint32_t p = p0; p < p1; p++) {
            uint32_t nw = 0;
            walk(p, h, w, [&](uint32_t i) {  // insertion sort by y (the lists are in arbitrary order)
                uint32_t t = nw++;
                while (t > 0 && my[off + t - 1] > i) {
                    my[off + t] = my[off + t - 1];
                    t--;
                }
                my[off + t] = i;
            });
            for (uint32_t t = 0; t < nw; t++) mx[off + t] = p;
            off += nw;
            if (p + 1 < p1) {
                h = (h - sx[p] * bk1) * kRollBase + sx[p + k];
                window_roll(p, w);
            }
        }
    }
    if (tid == 0) {
        st->n_matches = over ? 0 : total;
        st->flags = over ? BP_HOST_FALLBACK : BP_OK;
    }
}

// ------------------------------------------------------------------------------------------- B2
// Fenwick node: the reference's PrevPtr order (plane, score, d, id[, x, y]) as two 64-bit keys.
//   hi = plane << 32 | score        lo = d << 32 | id << 20 | x_end      (id decides before x is looked at)
struct Frag {
    uint64_t hi, lo;
};
__device__ __forceinline__ bool frag_less(const Frag& p, const Frag& q) { return p.hi < q.hi || (p.hi == q.hi && p.lo < q.lo); }

// LDS_TREE: tree / score / back live in LDS (lowest latency per event, but LDS limits a CU to one to three
// pairs); otherwise in global scratch (L2-resident; every event costs a memory round trip, but tens of
// wavefronts per CU hide it).  Only the sort buffer of the end-y values is always in LDS.
// PART: 0 the whole chain in one launch; 1 only the preparation (sort of the end coordinates in LDS, tree positions,
// continuation links, empty tree); 2 only the event loop + the path.  With the tree in global scratch the event loop
// needs no LDS at all, and launched on its own it is resident 32 wavefronts per CU instead of the 10 that 16 KB of
// sort buffer per wavefront allow — every event is a dependent L2 round trip that only occupancy hides
// (11.7 ms instead of 26 per 16 384 pairs).
template <bool LDS_TREE, int PART = 0>
__global__ __launch_bounds__(64) void chain_kernel(const BandDevArgs a) {
    extern __shared__ __align__(16) uint8_t s_raw[];
    // This kernel mostly waits for memory; next to a fill (two wavefronts per SIMD that always have a vector instruction
    // ready, and are older) it otherwise issues only when both of them stall
    __builtin_amdgcn_s_setprio(3);
    const uint32_t pair = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    BandDevPair* st = a.state + pair;
    if (st->flags != BP_OK) return;
    const uint32_t nm = st->n_matches;
    uint32_t* path = a.path + (size_t)pair * a.cap_matches;
    if (nm == 0) {
        if (lane == 0) st->n_path = 0;
        return;
    }
    const uint32_t* mx = a.mx + (size_t)pair * a.cap_matches;
    const uint32_t* my = a.my + (size_t)pair * a.cap_matches;
    uint32_t* qpos = a.qpos + (size_t)pair * a.cap_matches;  // tree position a start event queries
    uint32_t* upos = a.upos + (size_t)pair * a.cap_matches;  // tree position an end event raises
    int32_t* cont = a.cont + (size_t)pair * a.cap_matches;   // the match one diagonal step earlier, or -1
    const uint32_t k = a.k;
    uint32_t np2 = 64;
    while (np2 < nm) np2 <<= 1;
    if (nm < a.chain_min || nm > a.chain_cap) return;  // another launch (LDS size class) owns this pair
    // LDS carve-up for chain_cap matches: tree | score | back.  The sort buffer of the end-y values
    // aliases the tree, which is only zeroed once the tree positions have been computed.
    uint32_t* ye = (uint32_t*)s_raw;                                 // sorted end-y values, [np2 <= chain_cap + 1]
    Frag* tree;       // [nm + 1], 1-based
    uint32_t* score;  // [nm]
    int16_t* back;    // [nm], indices < 4096
    if (LDS_TREE) {
        tree = (Frag*)s_raw;
        score = (uint32_t*)(s_raw + 16 * (size_t)(a.chain_cap + 1));
        back = (int16_t*)(score + a.chain_cap);
    } else {
        tree = (Frag*)a.g_tree + (size_t)pair * (a.cap_matches + 1);
        score = a.g_score + (size_t)pair * a.cap_matches;
        back = a.g_back + (size_t)pair * a.cap_matches;
    }
    if (PART != 2) {
    for (uint32_t i = lane; i < np2; i += 64) ye[i] = i < nm ? my[i] + k : kNone;
    for (uint32_t i = lane; i < nm; i += 64) {
        int32_t c = -1;
        const uint32_t cx = mx[i], cy = my[i];
        if (cx > 0 && cy > 0) {
            // sparse.rs:265-267 looks (x - 1, y - 1) up by binary search.  The matches are sorted by (x, y), so if it
            // exists it sits among the few matches of x - 1 and x just before this one: walk back over them (one or
            // two steps on ordinary reads, at most 2 kMaxMatchesPerKmer) instead of eleven dependent probes
            const uint32_t kx = cx - 1, ky = cy - 1;
            for (uint32_t j = i; j-- > 0;) {
                const uint32_t vx = mx[j];
                if (vx < kx) break;
                if (vx == kx) {
                    const uint32_t vy = my[j];
                    if (vy == ky) c = (int32_t)j;
                    if (vy <= ky) break;
                }
            }
        }
        cont[i] = c;
    }
    __syncthreads();
    // bitonic sort of the end-y values: rank compression of the tree's coordinate
    for (uint32_t size = 2; size <= np2; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = lane; t < (np2 >> 1); t += 64) {
                const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = (i & size) == 0;
                const uint32_t vi = ye[i], vj = ye[j];
                if ((vi > vj) == up) {
                    ye[i] = vj;
                    ye[j] = vi;
                }
            }
            __syncthreads();
        }
    // tree position of a coordinate v = number of end-y values <= v (equal values share a node)
    auto pos_of = [&](uint32_t v) -> uint32_t {
        uint32_t lo = 0, hi = nm;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ye[mid] <= v)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    for (uint32_t i = lane; i < nm; i += 64) {
        qpos[i] = pos_of(my[i]);
        upos[i] = pos_of(my[i] + k);
    }
    __syncthreads();
    for (uint32_t i = lane; i <= nm; i += 64) tree[i] = Frag{0, 0};
    __syncthreads();
    }
    if (PART == 1) return;

    const uint32_t ms = a.match_score;
    const uint32_t go = (uint32_t)(-(int64_t)a.gap_open), ge = (uint32_t)(-(int64_t)a.gap_extend);
    uint32_t best_score = k;  // (k, 0): sparse.rs:234
    int32_t best_idx = 0;
    auto better = [](uint32_t s1, int32_t i1, uint32_t s2, int32_t i2) { return s1 > s2 || (s1 == s2 && i1 > i2); };
    // Events in (x, y, tag) order: ends (tag = index) before starts (tag = index + nm) at equal
    // coordinates.  Both lists are sorted already, so the order is a merge of two cursors.  Each lane
    // keeps one record of the current 64-record window of either list; the current record is read
    // with a wave-uniform lane index, a window is reloaded every 64 events of its kind.
    uint32_t s = 0, e = 0;
    uint32_t ws_x = 0, ws_y = 0, ws_q = 0, we_x = 0, we_y = 0, we_u = 0;
    int32_t we_c = -1;
    auto load_s = [&](uint32_t base) {
        const uint32_t i = base + lane;
        if (i < nm) {
            ws_x = mx[i];
            ws_y = my[i];
            ws_q = qpos[i];
        }
    };
    auto load_e = [&](uint32_t base) {
        const uint32_t i = base + lane;
        if (i < nm) {
            we_x = mx[i] + k;
            we_y = my[i] + k;
            we_u = upos[i];
            we_c = cont[i];
        }
    };
    auto pick = [](uint32_t v, uint32_t idx) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)idx); };
    // maximum of the 128-bit keys of lanes 0..15, in every lane: four row_shr steps, then lane 15 is read
    auto row_max = [&](Frag v) -> Frag {
#define BG_SHR(x, ctl) (uint32_t) __builtin_amdgcn_update_dpp(0, (int)(x), ctl, 0xf, 0xf, true)
        uint32_t h1 = (uint32_t)(v.hi >> 32), h0 = (uint32_t)v.hi, l1 = (uint32_t)(v.lo >> 32), l0 = (uint32_t)v.lo;
#define BG_STEP(ctl)                                                                                   \
        {                                                                                                  \
            const uint32_t q1 = BG_SHR(h1, ctl), q0 = BG_SHR(h0, ctl), r1 = BG_SHR(l1, ctl), r0 = BG_SHR(l0, ctl); \
            const uint64_t qh = (uint64_t)q1 << 32 | q0, ql = (uint64_t)r1 << 32 | r0;                         \
            const uint64_t vh = (uint64_t)h1 << 32 | h0, vl = (uint64_t)l1 << 32 | l0;                         \
            if (vh < qh || (vh == qh && vl < ql)) {                                                            \
                h1 = q1;                                                                                       \
                h0 = q0;                                                                                       \
                l1 = r1;                                                                                       \
                l0 = r0;                                                                                       \
            }                                                                                                  \
        }
        BG_STEP(0x111) BG_STEP(0x112) BG_STEP(0x114) BG_STEP(0x118)
#undef BG_STEP
#undef BG_SHR
        Frag r;
        r.hi = (uint64_t)pick(h1, 15) << 32 | pick(h0, 15);
        r.lo = (uint64_t)pick(l1, 15) << 32 | pick(l0, 15);
        return r;
    };
    load_s(0);
    load_e(0);
    while (e < nm) {
        const uint32_t ex = pick(we_x, e & 63), ey = pick(we_y, e & 63);
        bool take_start = false;
        uint32_t sx = 0, sy = 0;
        if (s < nm) {
            sx = pick(ws_x, s & 63);
            sy = pick(ws_y, s & 63);
            take_start = sx < ex || (sx == ex && sy < ey);
        }
        if (take_start) {
            const uint32_t p = s;
            uint32_t sc = k * ms;
            int32_t bk = -1;
            // prefix maximum over tree[1 .. pos]: the chain pos, pos - lowbit(pos), ... has one node per set
            // bit b of pos, pos with the bits below b cleared; lane b reads it (pos < 4096: lanes 0..11)
            const uint32_t i = pick(ws_q, s & 63);
            Frag bp{0, 0};
            if (lane < 13 && ((i >> lane) & 1u)) bp = tree[i & ~((1u << lane) - 1u)];
            bp = row_max(bp);
            const uint32_t bscore = (uint32_t)bp.hi;
            if (bscore > 0) {
                const uint32_t bid = ((uint32_t)bp.lo >> 20) & 0xFFFu, d = (uint32_t)(bp.lo >> 32);
                const uint32_t bx = (uint32_t)bp.lo & 0xFFFFFu, by = d - bx;
                const uint32_t gap = max(sx - bx, sy - by);
                const uint32_t pen = gap > 0 ? go + gap * ge : 0;
                const uint32_t sum = bscore + k * ms;
                const uint32_t ns = sum > pen ? sum - pen : 0;
                if (better(ns, (int32_t)bid, sc, bk)) {
                    sc = ns;
                    bk = (int32_t)bid;
                }
                if (better(sc, (int32_t)p, best_score, best_idx)) {
                    best_score = sc;
                    best_idx = (int32_t)p;
                }
            }
            if (lane == 0) {
                score[p] = sc;
                back[p] = (int16_t)bk;
            }
            s++;
            if ((s & 63) == 0) load_s(s);
        } else {
            const uint32_t p = e;
            uint32_t sc = score[p];
            int32_t bk = back[p];
            const int32_t c = (int32_t)pick((uint32_t)we_c, e & 63);
            if (c >= 0) {
                const uint32_t cs = score[c] + ms;
                if (better(cs, c, sc, bk)) {
                    sc = cs;
                    bk = c;
                }
                if (better(sc, (int32_t)p, best_score, best_idx)) {
                    best_score = sc;
                    best_idx = (int32_t)p;
                }
                if (lane == 0) {
                    score[p] = sc;
                    back[p] = (int16_t)bk;
                }
            }
            const uint32_t d = ex + ey;
            Frag f;
            f.hi = (uint64_t)(uint32_t)(sc + d * ge) << 32 | sc;
            f.lo = (uint64_t)d << 32 | (uint64_t)p << 20 | ex;
            // raise tree[pos], tree[pos + lowbit(pos)], ...: the chain is the set of distinct round-ups of pos
            // to multiples of 2^b; lane b takes the one for b if it differs from the one for b - 1
            const uint32_t i = pick(we_u, e & 63);
            if (lane < 13) {
                const uint32_t cb = ((i - 1) | ((1u << lane) - 1u)) + 1u;
                const uint32_t cp = lane ? ((i - 1) | ((1u << (lane - 1)) - 1u)) + 1u : 0u;
                if (cb != cp && cb <= nm && frag_less(tree[cb], f)) tree[cb] = f;
            }
            e++;
            if ((e & 63) == 0) load_e(e);
        }
        __builtin_amdgcn_wave_barrier();  // one wavefront, LDS is in order: nothing to wait for, only keep the order
    }
    __syncthreads();
    // the chain, last element first, then reversed in place
    if (lane == 0) {
        uint32_t len = 0;
        for (int32_t q = best_idx; q >= 0; q = back[q]) path[len++] = (uint32_t)q;
        st->n_path = len;
    }
    __syncthreads();
    const uint32_t len = st->n_path;
    for (uint32_t t = lane; t < len / 2; t += 64) {
        const uint32_t u = path[t];
        path[t] = path[len - 1 - t];
        path[len - 1 - t] = u;
    }
}

// ------------------------------------------------------------------------------------------- B2, four pairs per wavefront
// chain_rows_kernel: the event loop + path of chain_kernel<false, 2> with ONE PAIR PER DPP ROW OF 16 LANES (round 5).
// The loop above is one dependent memory round trip per event, ~3 600 events per 10 kb pair, and of its 64 lanes thirteen
// touch the tree; its throughput is the number of pairs in flight.  Next to a K3p fill (2 x 188 VGPRs per SIMD) a wavefront
// slot is what is scarce: four pairs per wavefront put the 16 384 pairs of a sub-batch in flight at once instead of a
// third of them (the chaining was the longest kernel of the builder's chain, and the builder's chain the cycle of the
// pipeline: profiles/r04_banded_timeline_k3p.txt).  Same events in the same order per pair, same tree operations, same
// tie-breaking — only where a pair's lanes sit changes:
//   * lanes 0..12 of a row read / raise the Fenwick nodes (row_max is a DPP row operation already), lane 15 — where the
//     row_shr maximum lands — is the row's scalar lane: it computes the start event's score, stores score / back and keeps
//     the running best;
//   * the current start / end records come out of 16-record windows (one record per lane of the row) with ds_bpermute
//     instead of a wave-uniform readlane; a window is reloaded every 16 events of its kind;
//   * the four rows take different branches at one step: both event bodies are predicated (exec-masked per row), each is
//     skipped when no row wants it.
// Per-pair arrays are addressed as 32-bit element offsets from the kernel arguments (scalar bases): 31 VGPRs.
__device__ __forceinline__ uint32_t row_pick(uint32_t v, uint32_t idx) {  // lane (row, idx & 15) of v, per row
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)threadIdx.x & 48u) | (idx & 15u)) << 2), (int)v);
}

__global__ __launch_bounds__(64) void chain_rows_kernel(const BandDevArgs a) {
    __builtin_amdgcn_s_setprio(3);  // (see chain_kernel)
    const uint32_t lane = threadIdx.x, rl = lane & 15u;
    const uint32_t pair = blockIdx.x * 4u + (lane >> 4);
    uint32_t nm = 0;
    if (pair < a.n_pairs) {
        const BandDevPair* st = a.state + pair;
        if (st->flags == BP_OK) {
            nm = st->n_matches;
            if (nm == 0 && rl == 0) a.state[pair].n_path = 0;
            if (nm < a.chain_min || nm > a.chain_cap) nm = 0;  // another launch (LDS size class) owns this pair
        }
    }
    if (__builtin_amdgcn_ballot_w64(nm != 0) == 0) return;
    const uint32_t eb = pair * a.cap_matches;         // element offset of this pair's slices (nm == 0: never dereferenced)
    const uint32_t tb = pair * (a.cap_matches + 1u);  // ... of its tree
    const uint32_t* const g_mx = a.mx;
    const uint32_t* const g_my = a.my;
    const uint32_t* const g_qpos = a.qpos;
    const uint32_t* const g_upos = a.upos;
    const int32_t* const g_cont = a.cont;
    Frag* const g_tree = (Frag*)a.g_tree;
    uint32_t* const g_score = a.g_score;
    int16_t* const g_back = a.g_back;
    const uint32_t k = a.k, ms = a.match_score;
    const uint32_t go = (uint32_t)(-(int64_t)a.gap_open), ge = (uint32_t)(-(int64_t)a.gap_extend);
    uint32_t best_score = k;  // (k, 0): sparse.rs:234 — kept by the row's lane 15
    int32_t best_idx = 0;
    auto better = [](uint32_t s1, int32_t i1, uint32_t s2, int32_t i2) { return s1 > s2 || (s1 == s2 && i1 > i2); };
    uint32_t s = 0, e = 0;
    uint32_t ws_x = 0, ws_y = 0, ws_q = 0, we_x = 0, we_y = 0, we_u = 0;
    int32_t we_c = -1;
    auto load_s = [&](uint32_t base) {  // (callers are predicated per row)
        const uint32_t i = base + rl;
        if (i < nm) {
            ws_x = g_mx[eb + i];
            ws_y = g_my[eb + i];
            ws_q = g_qpos[eb + i];
        }
    };
    auto load_e = [&](uint32_t base) {
        const uint32_t i = base + rl;
        if (i < nm) {
            we_x = g_mx[eb + i] + k;
            we_y = g_my[eb + i] + k;
            we_u = g_upos[eb + i];
            we_c = g_cont[eb + i];
        }
    };
    // maximum of the 128-bit keys of a row's lanes 0..15 in its lane 15 (four row_shr steps; the other lanes hold prefixes)
    auto row_max15 = [&](Frag v) -> Frag {
#define BG_SHR(x, ctl) (uint32_t) __builtin_amdgcn_update_dpp(0, (int)(x), ctl, 0xf, 0xf, true)
        uint32_t h1 = (uint32_t)(v.hi >> 32), h0 = (uint32_t)v.hi, l1 = (uint32_t)(v.lo >> 32), l0 = (uint32_t)v.lo;
#define BG_STEP(ctl)                                                                                       \
        {                                                                                                      \
            const uint32_t q1 = BG_SHR(h1, ctl), q0 = BG_SHR(h0, ctl), r1 = BG_SHR(l1, ctl), r0 = BG_SHR(l0, ctl); \
            const uint64_t qh = (uint64_t)q1 << 32 | q0, ql = (uint64_t)r1 << 32 | r0;                         \
            const uint64_t vh = (uint64_t)h1 << 32 | h0, vl = (uint64_t)l1 << 32 | l0;                         \
            if (vh < qh || (vh == qh && vl < ql)) {                                                            \
                h1 = q1;                                                                                       \
                h0 = q0;                                                                                       \
                l1 = r1;                                                                                       \
                l0 = r0;                                                                                       \
            }                                                                                                  \
        }
        BG_STEP(0x111) BG_STEP(0x112) BG_STEP(0x114) BG_STEP(0x118)
#undef BG_STEP
#undef BG_SHR
        return Frag{(uint64_t)h1 << 32 | h0, (uint64_t)l1 << 32 | l0};
    };
    load_s(0);
    load_e(0);
    while (__builtin_amdgcn_ballot_w64(e < nm) != 0) {
        const bool on = e < nm;
        // Events in (x, y, tag) order: ends before starts at equal coordinates; both lists are sorted: a merge of two cursors
        const uint32_t ex = row_pick(we_x, e), ey = row_pick(we_y, e);
        const uint32_t sx = row_pick(ws_x, s), sy = row_pick(ws_y, s);
        const bool take_start = on && s < nm && (sx < ex || (sx == ex && sy < ey));
        const bool take_end = on && !take_start;
        if (__builtin_amdgcn_ballot_w64(take_start) != 0) {
            // prefix maximum over tree[1 .. pos]: one node per set bit b of pos (pos with the bits below b cleared); lane b reads it
            const uint32_t i = row_pick(ws_q, s);
            Frag bp{0, 0};
            if (take_start && rl < 13 && ((i >> rl) & 1u)) bp = g_tree[tb + (i & ~((1u << rl) - 1u))];
            bp = row_max15(bp);  // (DPP moves run for every row; rows that are not at a start event carry zeros)
            if (take_start && rl == 15) {
                const uint32_t p = s;
                uint32_t sc = k * ms;
                int32_t bk = -1;
                const uint32_t bscore = (uint32_t)bp.hi;
                if (bscore > 0) {
                    const uint32_t bid = ((uint32_t)bp.lo >> 20) & 0xFFFu, d = (uint32_t)(bp.lo >> 32);
                    const uint32_t bx = (uint32_t)bp.lo & 0xFFFFFu, by = d - bx;
                    const uint32_t gap = max(sx - bx, sy - by);
                    const uint32_t pen = gap > 0 ? go + gap * ge : 0;
                    const uint32_t sum = bscore + k * ms;
                    const uint32_t ns = sum > pen ? sum - pen : 0;
                    if (better(ns, (int32_t)bid, sc, bk)) {
                        sc = ns;
                        bk = (int32_t)bid;
                    }
                    if (better(sc, (int32_t)p, best_score, best_idx)) {
                        best_score = sc;
                        best_idx = (int32_t)p;
                    }
                }
                g_score[eb + p] = sc;
                g_back[eb + p] = (int16_t)bk;
            }
            if (take_start) {
                s++;
                if ((s & 15u) == 0) load_s(s);
            }
        }
        if (__builtin_amdgcn_ballot_w64(take_end) != 0) {
            const int32_t c = (int32_t)row_pick((uint32_t)we_c, e);
            const uint32_t i = row_pick(we_u, e);
            if (take_end) {
                const uint32_t p = e;
                uint32_t sc = g_score[eb + p];  // (every lane of the row reads the same words: one request)
                int32_t bk = g_back[eb + p];
                if (c >= 0) {
                    const uint32_t cs = g_score[eb + (uint32_t)c] + ms;
                    if (better(cs, c, sc, bk)) {
                        sc = cs;
                        bk = c;
                    }
                    if (rl == 15) {
                        if (better(sc, (int32_t)p, best_score, best_idx)) {
                            best_score = sc;
                            best_idx = (int32_t)p;
                        }
                        g_score[eb + p] = sc;
                        g_back[eb + p] = (int16_t)bk;
                    }
                }
                const uint32_t d = ex + ey;
                Frag f;
                f.hi = (uint64_t)(uint32_t)(sc + d * ge) << 32 | sc;
                f.lo = (uint64_t)d << 32 | (uint64_t)p << 20 | ex;
                // raise tree[pos], tree[pos + lowbit(pos)], ...: lane b takes the round-up of pos to a multiple of 2^b if it
                // differs from the one for b - 1
                if (rl < 13) {
                    const uint32_t cb = ((i - 1) | ((1u << rl) - 1u)) + 1u;
                    const uint32_t cp = rl ? ((i - 1) | ((1u << (rl - 1)) - 1u)) + 1u : 0u;
                    if (cb != cp && cb <= nm && frag_less(g_tree[tb + cb], f)) g_tree[tb + cb] = f;
                }
                e++;
                if ((e & 15u) == 0) load_e(e);
            }
        }
        __builtin_amdgcn_wave_barrier();  // one wavefront, its memory operations stay in order: only keep the order
    }
    // the chain, last element first (the row's lane 15 walks it), then reversed in place by the row
    uint32_t len = 0;
    if (nm != 0 && rl == 15) {
        uint32_t* const path = a.path + eb;
        for (int32_t q = best_idx; q >= 0; q = g_back[eb + (uint32_t)q]) path[len++] = (uint32_t)q;
        a.state[pair].n_path = len;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    len = row_pick(len, 15);
    if (nm != 0) {
        uint32_t* const path = a.path + eb;
        for (uint32_t t = rl; t < len / 2; t += 16) {
            const uint32_t u = path[t];
            path[t] = path[len - 1 - t];
            path[len - 1 - t] = u;
        }
    }
}

// ------------------------------------------------------------------------------------------- B3
// One tile of the per-column ranges in LDS: columns [j0, j0 + kBandTile) — 2048 columns, 16 KB: measured per 16 384
// 10 kb pairs 12.2 ms with 8192 columns (two blocks per CU), 7.6 ms with 4096, 7.4 ms with 2048 (the path is walked once
// per tile; what costs is the LDS atomics of a block, and more blocks per CU hide them).  Every operation only
// lowers start[j] or raises end[j]; columns outside the tile are skipped (another pass owns them).
#ifndef BG_BAND_TILE
#define BG_BAND_TILE 2048
#endif
constexpr uint32_t kBandTile = BG_BAND_TILE;
struct BandCols {
    uint32_t* start;  // LDS, indexed by j - j0
    uint32_t* end;
    uint32_t rows, cols;  // m + 1, n + 1
    uint32_t j0, j1;      // tile: [j0, j1)
};
__device__ __forceinline__ uint32_t ssub(uint32_t p, uint32_t q) { return p > q ? p - q : 0; }

// banded.rs:1111-1120
// (sub, nsub): the columns of the box are shared by nsub cooperating lanes
__device__ void add_entry(const BandCols& b, uint32_t r, uint32_t c, uint32_t w, uint32_t sub = 0, uint32_t nsub = 1) {
    const uint32_t lo = ssub(r, w), hi = (uint32_t)min((uint64_t)r + w + 1, (uint64_t)b.rows);
    for (uint64_t j = max((uint64_t)ssub(c, w), (uint64_t)b.j0) + sub, je = min(min((uint64_t)c + w + 1, (uint64_t)b.cols), (uint64_t)b.j1); j < je; j += nsub) {
        atomicMin(&b.start[j - b.j0], lo);
        atomicMax(&b.end[j - b.j0], hi);
    }
}
// banded.rs:1071-1107.  (tid, nth): the loops are shared by nth cooperating threads — every step only
// lowers a start or raises an end, so any split works.
__device__ void add_kmer(const BandCols& b, uint32_t r_, uint32_t c_, uint32_t k, uint32_t w, uint32_t tid = 0, uint32_t nth = 1) {
    if (k == 0) return;
    const uint64_t r = r_, c = c_, rows = b.rows, cols = b.cols;
    const uint64_t t0 = b.j0, t1 = b.j1;
    auto lower = [&](uint64_t j, uint32_t v) {
        if (j >= t0 && j < t1) atomicMin(&b.start[j - t0], v);
    };
    auto raise = [&](uint64_t j, uint32_t v) {
        if (j >= t0 && j < t1) atomicMax(&b.end[j - t0], v);
    };
    const uint64_t i0 = ssub(r_, w);
    for (uint64_t j = (uint64_t)ssub(c_, w) + tid, je = min(c + w + 1, cols); j < je; j += nth) lower(j, (uint32_t)i0);
    const uint64_t ja = min(c + w, cols);
    for (uint64_t j = ja + tid, je = min(c + k + w, cols); j < je; j += nth) lower(j, (uint32_t)(i0 + (j - ja)));
    const uint64_t jl = c + k - 1 > w ? c + k - 1 - w : 0, jf = c > w ? c - w : 0;
    for (uint64_t s = 1 + tid; s <= jl - min(jl, jf); s += nth)  // j = jl - s, i = r + w + k - s
        raise(jl - s, (uint32_t)min(r + w + k - s, rows));
    const uint32_t e = (uint32_t)min(r + w + k, rows);
    for (uint64_t j = jl + tid, je = min(c + k + w, cols); j < je; j += nth) raise(j, e);
}
// banded.rs:1123-1137 (u32 arithmetic like the reference's release build)
__device__ void add_gap(const BandCols& b, uint32_t r0, uint32_t c0, uint32_t r1, uint32_t c1, uint32_t w, uint32_t tid = 0, uint32_t nth = 1) {
    const uint32_t nr = r1 - r0, nc = c1 - c0;
    if (nr > nc) {
        for (uint64_t r = (uint64_t)r0 + tid; r < r1; r += nth)
            add_entry(b, (uint32_t)r, c0 + (c1 - c0) * ((uint32_t)r - r0) / (r1 - r0), w);
    } else {
        // only the columns whose +-w box can touch the tile
        const uint32_t ca = max(c0, ssub(b.j0, w)), cb = (uint32_t)min((uint64_t)c1, (uint64_t)b.j1 + w);
        for (uint64_t c = (uint64_t)ca + tid; c < cb; c += nth)
            add_entry(b, r0 + (r1 - r0) * ((uint32_t)c - c0) / (c1 - c0), (uint32_t)c, w);
    }
}
// add_gap for a group of nsub lanes that share the columns of every entry (gaps between chained matches are a few
// points long: sharing the points would leave most of the group idle)
__device__ void add_gap_cols(const BandCols& b, uint32_t r0, uint32_t c0, uint32_t r1, uint32_t c1, uint32_t w, uint32_t sub, uint32_t nsub) {
    const uint32_t nr = r1 - r0, nc = c1 - c0;
    if (nr > nc) {
        for (uint64_t r = r0; r < r1; r++) add_entry(b, (uint32_t)r, c0 + (c1 - c0) * ((uint32_t)r - r0) / (r1 - r0), w, sub, nsub);
    } else {
        const uint32_t ca = max(c0, ssub(b.j0, w)), cb = (uint32_t)min((uint64_t)c1, (uint64_t)b.j1 + w);
        for (uint64_t c = ca; c < cb; c++) add_entry(b, r0 + (r1 - r0) * ((uint32_t)c - c0) / (c1 - c0), (uint32_t)c, w, sub, nsub);
    }
}
// banded.rs:1150-1276
// (all threads of the block call this with the same arguments and share the loops)
__device__ void set_boundaries(const BandCols& b, uint32_t fx, uint32_t fy, uint32_t lx, uint32_t ly, uint32_t k, uint32_t w,
                               const BandDevArgs& a, uint32_t tid, uint32_t nth) {
    const uint64_t lazy = 2ull * k, rows = b.rows, cols = b.cols;
    {
        const uint64_t r = fx, c = fy;
        if (r != 0 || c != 0) {
            const int32_t to_start = (r > 0 ? a.xclip_prefix : 0) + (c > 0 ? a.yclip_prefix : 0);
            if (to_start == 0) {
                const uint64_t d = min(lazy, min(r, c));
                add_kmer(b, (uint32_t)(r - d), (uint32_t)(c - d), (uint32_t)d, w, tid, nth);
                add_gap(b, (uint32_t)(r > lazy ? r - lazy : 0), (uint32_t)(c > lazy ? c - lazy : 0), (uint32_t)(r - d), (uint32_t)(c - d), w, tid, nth);
            } else {
                const int32_t diag = r > c ? a.xclip_prefix : (r < c ? a.yclip_prefix : 0);
                if (diag == 0) {
                    const uint64_t d = min(r, c);
                    add_kmer(b, (uint32_t)(r - d), (uint32_t)(c - d), (uint32_t)d, w, tid, nth);
                    const uint32_t sr = (uint32_t)(r > lazy ? r - lazy : 0), sc = (uint32_t)(c > lazy ? c - lazy : 0);
                    const uint32_t er = (uint32_t)(r - d), ec = (uint32_t)(c - d);
                    if (sr <= er && sc <= ec) add_gap(b, sr, sc, er, ec, w, tid, nth);
                } else {
                    add_gap(b, 0, 0, fx, fy, w, tid, nth);
                }
            }
        }
    }
    {
        const uint64_t r = (uint64_t)lx + k, c = (uint64_t)ly + k;
        if (!(r == rows && c == cols)) {
            const int32_t from_end = (r == rows ? 0 : a.xclip_suffix) + (c == cols ? 0 : a.yclip_suffix);
            const uint64_t dr = rows - r, dc = cols - c;
            bool diagonal = false;
            uint64_t d = 0;
            if (from_end == 0) {
                d = min(lazy, min(dr, dc));
                diagonal = true;
            } else {
                const int32_t diag = dr > dc ? a.xclip_suffix : (dr < dc ? a.yclip_suffix : 0);
                if (diag == 0) {
                    d = min(dr, dc);
                    diagonal = true;
                }
            }
            if (diagonal) {
                add_kmer(b, (uint32_t)r, (uint32_t)c, (uint32_t)d, w, tid, nth);
                const uint64_t r1 = min(rows, r + d) - 1, c1 = min(cols, c + d) - 1;
                const uint64_t r2 = min(rows, r + lazy), c2 = min(cols, c + lazy);
                if (r1 <= r2 && c1 <= c2) add_gap(b, (uint32_t)r1, (uint32_t)c1, (uint32_t)r2, (uint32_t)c2, w, tid, nth);
            } else {
                add_gap(b, (uint32_t)r, (uint32_t)c, (uint32_t)rows, (uint32_t)cols, w, tid, nth);
            }
        }
    }
}

// Entries (r0 + s, c0 + s), s < len — what a run of diagonal continuations of the match path adds (banded.rs:1352-1357:
// one add_entry per continuation) — as one lower / raise per column: column j sees the entries max(0, j - w - c0) ..
// min(len - 1, j + w - c0), its start comes from the first of them, its end from the last.
__device__ void add_diag_run(const BandCols& b, uint32_t r0, uint32_t c0, uint32_t len, uint32_t w, uint32_t sub = 0, uint32_t nsub = 1) {
    const uint64_t jlo = max((uint64_t)ssub(c0, w), (uint64_t)b.j0);
    const uint64_t jhi = min(min((uint64_t)c0 + len + w, (uint64_t)b.cols), (uint64_t)b.j1);
    for (uint64_t j = jlo + sub; j < jhi; j += nsub) {
        const uint64_t s_min = j > (uint64_t)c0 + w ? j - w - c0 : 0, s_max = min((uint64_t)len - 1, j + w - c0);
        atomicMin(&b.start[j - b.j0], ssub((uint32_t)(r0 + s_min), w));
        atomicMax(&b.end[j - b.j0], (uint32_t)min((uint64_t)r0 + s_max + w + 1, (uint64_t)b.rows));
    }
}

__global__ __launch_bounds__(256) void band_kernel(const BandDevArgs a) {
    __shared__ uint32_t s_start[kBandTile], s_end[kBandTile];
    // path elements that do not continue their predecessor one step down the diagonal ("anchors"), in path order: a
    // thread takes an anchor together with the continuations behind it.  Nine path elements in ten are continuations;
    // with anchors and continuations interleaved over the lanes every wavefront walked both code paths at a tenth of
    // its width (7.4 -> 2.x ms per 16 384 pairs).
    __shared__ uint16_t s_anchor[kMaxChainMatches + 2];
    __shared__ uint32_t s_wtot[4], s_nanchor, s_lo, s_hi;
    constexpr uint32_t G = 4;  // lanes per anchor: they share the columns of its gap, its k-mer and its run
    __builtin_amdgcn_s_setprio(3);  // (runs next to a fill: see chain_kernel)
    const uint32_t pair = blockIdx.x;
    const BandDevPair* st = a.state + pair;
    if (st->flags != BP_OK) return;
    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo), n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
    uint32_t* g_start = a.col_start + (size_t)pair * (a.max_n + 1);
    uint32_t* g_end = a.col_end + (size_t)pair * (a.max_n + 1);
    const bool full = st->n_matches == 0;  // banded.rs:1309-1313: no matches -> the full matrix
    const uint32_t* mx = a.mx + (size_t)pair * a.cap_matches;
    const uint32_t* my = a.my + (size_t)pair * a.cap_matches;
    const uint32_t* path = a.path + (size_t)pair * a.cap_matches;
    // anchor coordinates, read by every tile pass without going through path[] (the chaining's tree positions are dead)
    uint32_t* acx = a.qpos + (size_t)pair * a.cap_matches;
    uint32_t* acy = a.upos + (size_t)pair * a.cap_matches;
    const uint32_t len = st->n_path, k = a.k, w = a.w;
    if (!full) {
        // every thread looks at a contiguous stretch of the path: anchors counted, scanned over the block, written in order
        const uint32_t per = (len + blockDim.x - 1) / blockDim.x, t0 = min(len, threadIdx.x * per), t1 = min(len, t0 + per);
        auto is_anchor = [&](uint32_t t) {
            return !(t > 0 && mx[path[t]] == mx[path[t - 1]] + 1 && my[path[t]] == my[path[t - 1]] + 1);
        };
        uint32_t mine = 0;
        for (uint32_t t = t0; t < t1; t++) mine += is_anchor(t) ? 1u : 0u;
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
            if ((threadIdx.x & 63) >= (uint32_t)o) incl += v;
        }
        if ((threadIdx.x & 63) == 63) s_wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (uint32_t wv = 0; wv < (blockDim.x >> 6); wv++) {
            if (wv < (threadIdx.x >> 6)) base += s_wtot[wv];
            total += s_wtot[wv];
        }
        uint32_t o = base + incl - mine;
        for (uint32_t t = t0; t < t1; t++)
            if (is_anchor(t)) {
                acx[o] = mx[path[t]];
                acy[o] = my[path[t]];
                s_anchor[o++] = (uint16_t)t;
            }
        if (threadIdx.x == 0) {
            s_anchor[total] = (uint16_t)len;  // end of the last run
            s_nanchor = total;
        }
        __syncthreads();
    }
    const uint32_t n_anchor = full ? 0 : s_nanchor;
    for (uint32_t j0 = 0; j0 <= n; j0 += kBandTile) {
        BandCols b;
        b.start = s_start;
        b.end = s_end;
        b.rows = m + 1;
        b.cols = n + 1;
        b.j0 = j0;
        b.j1 = min(n + 1, j0 + kBandTile);
        for (uint32_t j = threadIdx.x; j < b.j1 - j0; j += blockDim.x) {
            s_start[j] = full ? 0u : m + 1;  // empty range m+1..0 (banded.rs:1061-1067)
            s_end[j] = full ? m + 1 : 0u;
        }
        if (threadIdx.x == 0) {
            s_lo = 0xFFFFFFFFu;
            s_hi = 0;
        }
        __syncthreads();
        if (!full) {
            set_boundaries(b, mx[path[0]], my[path[0]], mx[path[len - 1]], my[path[len - 1]], k, w, a, threadIdx.x, blockDim.x);
            // the anchors whose columns (gap, k-mer, run: [gap start - w, run end + w]) touch the tile are a contiguous
            // range of the list: every thread tests its share, the range comes out of two LDS atomics
            for (uint32_t ai = threadIdx.x; ai < n_anchor; ai += blockDim.x) {
                const uint32_t t = s_anchor[ai], run = (uint32_t)s_anchor[ai + 1] - t - 1;  // continuations behind the anchor
                const uint32_t cy = acy[ai];
                if ((uint64_t)cy + k + run + w < j0) continue;  // entirely left of the tile
                // the path element before the anchor: the last continuation of the previous anchor
                const uint32_t first_col = ai > 0 ? ssub(acy[ai - 1] + ((uint32_t)s_anchor[ai] - (uint32_t)s_anchor[ai - 1] - 1) + k - 1, w) : ssub(cy, w);
                if (first_col >= b.j1) continue;  // entirely right of it
                atomicMin(&s_lo, ai);
                atomicMax(&s_hi, ai + 1);
            }
            __syncthreads();
            const uint32_t a_lo = s_lo, a_hi = s_hi;
            // banded.rs:1352-1365, G lanes per anchor
            for (uint32_t ai = a_lo < a_hi ? a_lo + threadIdx.x / G : a_hi; ai < a_hi; ai += blockDim.x / G) {
                const uint32_t sub = threadIdx.x % G;
                const uint32_t t = s_anchor[ai], run = (uint32_t)s_anchor[ai + 1] - t - 1;
                const uint32_t cx = acx[ai], cy = acy[ai];
                if (ai > 0) {
                    const uint32_t prun = (uint32_t)s_anchor[ai] - (uint32_t)s_anchor[ai - 1] - 1;
                    const uint32_t px = acx[ai - 1] + prun, py = acy[ai - 1] + prun;
                    add_gap_cols(b, px + (k - 1), py + (k - 1), cx, cy, w, sub, G);
                }
                add_kmer(b, cx, cy, k, w, sub, G);
                // continuation s of the anchor is add_entry((cx + s) + k, (cy + s) + k)
                if (run) add_diag_run(b, cx + k, cy + k, run, w, sub, G);
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < b.j1 - j0; j += blockDim.x) {
            g_start[j0 + j] = s_start[j];
            g_end[j0 + j] = s_end[j];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------- B4
__device__ __forceinline__ uint64_t block_sum(uint64_t v, uint64_t* s_tmp) {
    s_tmp[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = blockDim.x >> 1; o; o >>= 1) {
        if (threadIdx.x < o) s_tmp[threadIdx.x] += s_tmp[threadIdx.x + o];
        __syncthreads();
    }
    const uint64_t r = s_tmp[0];
    __syncthreads();
    return r;
}

// Column ranges -> per-row column ranges.  The band Band::create rasterises has no empty column between
// its first and last one, and starts / ends never decrease: then the first column of row i is the first
// one whose end exceeds i, its last column the last one whose start does not — every column hands its index to
// the rows it is the first / the last column of.  Anything else goes back to the host builder.
__global__ __launch_bounds__(256) void band_rows_kernel(const BandDevArgs a) {
    __builtin_amdgcn_s_setprio(3);  // (runs next to a fill: see chain_kernel)
    const uint32_t pair = blockIdx.x;
    BandDevPair* st = a.state + pair;
    if (st->flags == BP_HOST_FALLBACK) return;
    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo), n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
    const uint32_t* start = a.col_start + (size_t)pair * (a.max_n + 1);
    const uint32_t* end = a.col_end + (size_t)pair * (a.max_n + 1);
    int2* rowc = a.rowc + a.row0[pair];
    uint32_t* roff = a.row_off + a.row0[pair];
    __shared__ uint64_t s_tmp[256];
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_first, s_last, s_bad;
    if (threadIdx.x == 0) {
        s_first = 0xFFFFFFFFu;
        s_last = 0;
        s_bad = 0;
    }
    __syncthreads();
    // Band::num_cells (banded.rs:1374-1380), first / last non-empty column
    uint64_t cells = 0;
    uint32_t jf = 0xFFFFFFFFu, jl = 0;
    bool any = false;
    for (uint32_t j = threadIdx.x; j <= n; j += blockDim.x)
        if (end[j] > start[j]) {
            cells += end[j] - start[j];
            jf = min(jf, j);
            jl = max(jl, j);
            any = true;
        }
    if (any) {
        atomicMin(&s_first, jf);
        atomicMax(&s_last, jl);
    }
    cells = block_sum(cells, s_tmp);
    const uint32_t j_first = s_first, j_last = s_last;
    uint32_t flags = BP_OK;
    if (cells > 5000000ull)
        flags = BP_TOO_MANY_CELLS;  // banded.rs:104, 407-420
    else if (n == 0 || j_first == 0xFFFFFFFFu)
        flags = BP_UNSUPPORTED;
    if (flags == BP_OK) {
        for (uint32_t j = j_first + threadIdx.x; j <= j_last; j += blockDim.x) {
            if (end[j] <= start[j]) s_bad = 1;  // a hole
            if (j > j_first && (start[j] < start[j - 1] || end[j] < end[j - 1])) s_bad = 2;  // not monotone
        }
        __syncthreads();
        if (s_bad == 1) flags = BP_HOST_FALLBACK;  // the host sweep copes with holes
        if (s_bad == 2) flags = BP_UNSUPPORTED;
    }
    uint64_t tb_bytes = 0;
    if (flags == BP_OK) {
        // Both column arrays are non-decreasing here, so the rows whose first column is j are those from
        // min(end[j-1], m+1) up to min(end[j], m+1), and the rows whose last column is j those from start[j] up to
        // start[j+1] (up to m for the last column): two scatters from the columns instead of two searches per row.
        for (uint32_t i = threadIdx.x; i <= m; i += blockDim.x) rowc[i] = make_int2((int)(j_last + 1), -1);
        __syncthreads();
        for (uint32_t j = j_first + threadIdx.x; j <= j_last; j += blockDim.x) {
            const uint32_t e0 = j == j_first ? 0u : min(end[j - 1], m + 1), e1 = min(end[j], m + 1);
            for (uint32_t i = e0; i < e1; i++) rowc[i].x = (int)j;
            const uint32_t s0 = start[j], s1 = j == j_last ? m + 1 : min(start[j + 1], m + 1);
            for (uint32_t i = s0; i < s1; i++) rowc[i].y = (int)j;
        }
        __syncthreads();
        uint64_t covered = 0, run = 0;
        if (threadIdx.x == 0) {  // row 0: a closed form, not stored
            const int2 got = rowc[0];
            int2 rc = make_int2(1, 0);
            if ((uint32_t)got.x <= j_last && got.y >= 0 && got.x <= got.y) {
                rc = got;
                covered += (uint64_t)(rc.y - rc.x + 1);
            }
            rowc[0] = rc;
            roff[0] = 0;
        }
        for (uint32_t q0 = 0; q0 < m; q0 += blockDim.x) {  // tiles of 256 rows 1 + q0 ..: eight-row groups are eight lanes
            const uint32_t q = q0 + threadIdx.x, i = q + 1;
            uint32_t groups = 0;
            int2 rc = make_int2(1, 0);
            if (i <= m) {
                const int2 got = rowc[i];  // {first column whose end exceeds i, last column whose start is at most i}
                if ((uint32_t)got.x <= j_last && got.y >= 0 && got.x <= got.y) {
                    rc = got;
                    covered += (uint64_t)(rc.y - rc.x + 1);
                    groups = ((uint32_t)(rc.y - rc.x + 1) + (kTbRowAlign - 1)) / kTbRowAlign;
                }
                rowc[i] = rc;
            }
            // the eight rows of a line have as many group slots as the longest of them (banded_kernels.h)
            groups = max(groups, (uint32_t)__shfl_xor((int)groups, 1));
            groups = max(groups, (uint32_t)__shfl_xor((int)groups, 2));
            groups = max(groups, (uint32_t)__shfl_xor((int)groups, 4));
            const uint32_t width = (threadIdx.x & 7u) == 0 ? groups * kTbGroupStride : 0u;  // the line group's bytes, once
            // exclusive scan of the widths inside the tile: lane shifts inside a wavefront, one barrier for the four
            // wavefront totals (a 256-wide LDS scan is sixteen barriers per tile, forty tiles per 10 kb read)
            uint32_t incl = width;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
                if ((threadIdx.x & 63) >= (uint32_t)o) incl += v;
            }
            if ((threadIdx.x & 63) == 63) s_scan[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0, tile_total = 0;
            for (uint32_t wv = 0; wv < (blockDim.x >> 6); wv++) {
                if (wv < (threadIdx.x >> 6)) wbase += s_scan[wv];
                tile_total += s_scan[wv];
            }
            // (incl is the same in the eight lanes of a line group: only its first lane contributes)
            if (i <= m) roff[i] = (uint32_t)(run + wbase + incl - groups * kTbGroupStride + (q & (kTbLineRows - 1)) * kTbRowAlign);
            run += tile_total;
            __syncthreads();
        }
        const uint64_t cov = block_sum(covered, s_tmp);
        tb_bytes = (run + 15) & ~15ull;
        if (cov != cells || run > 0xFFFFFFF0ull) flags = BP_UNSUPPORTED;  // not one interval per row
    }
    if (flags != BP_OK) {
        for (uint32_t i = threadIdx.x; i <= m; i += blockDim.x) {
            rowc[i] = make_int2(1, 0);
            roff[i] = 0;
        }
        tb_bytes = 0;
    }
    if (threadIdx.x == 0) {
        st->flags = flags;
        st->cells = cells;
        st->tb_bytes = tb_bytes;
        st->start_0 = start[0];
        st->end_0 = end[0];
        st->start_n = start[n];
        st->end_n = end[n];
    }
}

}  // namespace

static size_t chain_lds_bytes(uint32_t cap) { return 16 * (size_t)(cap + 1) + 6 * (size_t)cap + 16; }

int launch_band_match(const BandDevArgs& a, hipStream_t st) {
    // positions are 16-bit in the LDS flavour, and everything of a pair has to fit 160 KB
    const size_t lds_bytes = lds_join_bytes(a.max_m, a.max_n);
    if (a.max_n < 0xFFFFu && lds_bytes <= 150 * 1024 && !a.join_global) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute((const void*)kmer_match_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        kmer_match_lds_kernel<<<dim3(a.n_pairs), dim3(kLdsJoinThreads), lds_bytes, st>>>(a);
        return hipGetLastError() == hipSuccess ? BG_OK : BG_ERR_HIP;
    }
    const size_t stage_bytes = (size_t)((a.max_m + 15) & ~15u) + a.max_n + 16;
    if (stage_bytes <= kStageMaxBytes)
        kmer_match_kernel<true><<<dim3(a.n_pairs), dim3(256), stage_bytes, st>>>(a);
    else
        kmer_match_kernel<false><<<dim3(a.n_pairs), dim3(256), 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? BG_OK : BG_ERR_HIP;
}

// part 1: what of the chaining wants LDS (the event preparation of the global-tree flavour); part 2: the rest (its event
// loop: 24 VGPRs, no LDS — the one builder kernel that runs well next to a K3v2 fill); 0: both
int launch_band_chain(const BandDevArgs& a, hipStream_t st, int part) {
    // two LDS size classes: most pairs of a long-read batch sit just around 2k matches
    BandDevArgs c = a;
    c.chain_min = 0;
    c.chain_cap = kMaxChainMatches;
    if (a.chain_global > 0 || (a.chain_global < 0 && a.n_pairs >= kChainGlobalMinPairs)) {
        // enough pairs to hide memory latency with occupancy: tree in global scratch, 16 KB of LDS per pair
        if (part != 2) chain_kernel<false, 1><<<dim3(a.n_pairs), dim3(64), 4 * (size_t)(kMaxChainMatches + 1), st>>>(c);
        if (part != 1) {
            // rows: four pairs per wavefront (chain_rows_kernel); 0: one pair per wavefront (round 2 .. 4, kept for A/B and tests)
            // (chain_rows_kernel addresses the per-pair slices with 32-bit element offsets, pair * (cap_matches + 1): a
            //  sub-batch a caller's chunk_pairs made larger than that goes through the 64-bit pointers of chain_kernel)
            if (a.chain_rows && (uint64_t)a.n_pairs * ((uint64_t)a.cap_matches + 1) < (1ull << 32))
                chain_rows_kernel<<<dim3((a.n_pairs + 3) / 4), dim3(64), 0, st>>>(c);
            else
                chain_kernel<false, 2><<<dim3(a.n_pairs), dim3(64), 0, st>>>(c);
        }
    } else if (part != 1) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute((const void*)chain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)chain_lds_bytes(kMaxChainMatches));
            attr_set = true;
        }
        // two LDS size classes: most pairs of a long-read batch sit just around 2k matches
        c.chain_cap = kSmallChainMatches;
        chain_kernel<true><<<dim3(a.n_pairs), dim3(64), chain_lds_bytes(c.chain_cap), st>>>(c);
        c.chain_min = kSmallChainMatches + 1;
        c.chain_cap = kMaxChainMatches;
        chain_kernel<true><<<dim3(a.n_pairs), dim3(64), chain_lds_bytes(c.chain_cap), st>>>(c);
    }
    return hipGetLastError() == hipSuccess ? BG_OK : BG_ERR_HIP;
}

int launch_band_raster(const BandDevArgs& a, hipStream_t st) {
    band_kernel<<<dim3(a.n_pairs), dim3(256), 0, st>>>(a);
    band_rows_kernel<<<dim3(a.n_pairs), dim3(256), 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? BG_OK : BG_ERR_HIP;
}

}  // namespace bgband_dev
