// Shared device pieces of the FM-index kernels (K5 backward search, K6 suffix-array lookup):
// the 2-bit block layout, the quad rank helper and the index handle.  Layout notes: fm_index.hip.
#ifndef BG_FM_KERNELS_H
#define BG_FM_KERNELS_H
#include <mutex>

#include "bg_common.h"

namespace bgfm {

constexpr uint32_t kSymPerBlock = 192;
constexpr uint32_t kMaxExcLds = 1024;   // exception positions staged in LDS
constexpr uint32_t kMaxExcSyms = 32;    // distinct exception byte values supported
constexpr uint8_t kClsZero = 4;         // in alphabet, never occurs in the BWT
constexpr uint8_t kClsExc = 8;          // kClsExc + e : exception symbol e
constexpr uint32_t kJumpK = 12;  // symbols covered by the jump table of K5 (4^12 entries x 16 bytes = 256 MB)
constexpr uint8_t kClsPanic = 255;      // not in the alphabet: the reference panics

struct FmDev {
    const uint4* blocks;
    const uint32_t* exc_pos;      // all exception positions, sorted
    const uint32_t* exc_sym_pos;  // per exception symbol, sorted, concatenated
    const uint8_t* sym_class;     // [256]
    const uint32_t* less;         // [256]
    uint32_t exc_sym_off[kMaxExcSyms + 1];
    uint32_t n;
    uint32_t n_exc;
};

// number of entries <= r in a sorted array
template <typename P>
__device__ __forceinline__ uint32_t count_le(P arr, uint32_t lo, uint32_t hi, uint32_t r) {
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (arr[mid] <= r)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// this lane's share of rank(code c) inside one block: lane t==0 holds the counters,
// lanes 1..3 hold 64 symbols each
__device__ __forceinline__ uint32_t block_part(const uint4 v, uint32_t t, uint32_t o, uint32_t c) {
    if (t == 0) {
        uint32_t lo = (c & 1) ? v.y : v.x;
        uint32_t hi = (c & 1) ? v.w : v.z;
        return (c & 2) ? hi : lo;
    }
    const int have = (int)o + 1 - (int)(t - 1) * 64;  // symbols of this lane inside [0, o]
    if (have <= 0) return 0;
    const uint64_t pat = (uint64_t)c * 0x5555555555555555ull;
    uint64_t w0 = ((uint64_t)v.y << 32) | v.x;
    uint64_t w1 = ((uint64_t)v.w << 32) | v.z;
    uint64_t e0 = ~(w0 ^ pat), e1 = ~(w1 ^ pat);
    e0 = e0 & (e0 >> 1) & 0x5555555555555555ull;
    e1 = e1 & (e1 >> 1) & 0x5555555555555555ull;
    const int t0 = have >= 32 ? 32 : have;
    const int t1 = have >= 64 ? 32 : (have > 32 ? have - 32 : 0);
    const uint64_t m0 = t0 == 32 ? ~0ull : ((1ull << (2 * t0)) - 1);
    const uint64_t m1 = t1 == 32 ? ~0ull : ((1ull << (2 * t1)) - 1);
    return (uint32_t)(__popcll(e0 & m0) + __popcll(e1 & m1));
}

}  // namespace bgfm

struct bg_fm {
    bg_ctx* ctx = nullptr;
    bgfm::FmDev dev = {};
    void* d_blocks = nullptr;
    void* d_exc_pos = nullptr;
    void* d_exc_sym_pos = nullptr;
    void* d_class = nullptr;
    void* d_less = nullptr;
    void* d_exc_byte = nullptr;  // byte value of every exception, parallel to exc_pos
    uint64_t bytes = 0;
    // suffix array attached for Interval::occ / SuffixArray::get (K6, sa_locate.hip)
    int sa_kind = 0;               // 0 none, 1 raw, 2 sampled
    void* d_sa = nullptr;          // raw: uint32 SA[n]; sampled: uint32 sample[]
    void* d_extra_row = nullptr;   // sampled: rows kept because their BWT byte is the sentinel (sorted)
    void* d_extra_pos = nullptr;
    uint64_t n_sample = 0, n_extra = 0;
    uint32_t sa_rate = 0;
    uint8_t sa_sentinel = 0;
    uint8_t code_byte[4] = {0, 0, 0, 0};  // byte value of each 2-bit code
    // K5's optional jump table (opt-in through bg_fm_set_option "jump_min_queries"; +2.5 % on a cache-resident
    // index, nothing on an HBM-resident one): built once, under jump_mu, and published only after the build
    // stream has synchronised — concurrent searches either see the finished table or none
    void* d_jump = nullptr;
    std::mutex jump_mu;
    uint64_t jump_min_queries = ~0ull;
    bool no_jump = true;
    int n_codes = 0;         // distinct bytes with a 2-bit code (<= 4)
    uint32_t less_len = 0;
    bool fmd_ok = false;  // the BWT is a word over dna::n_alphabet() + '$' (FMDIndex::from, fmindex.rs:323-327)
};

#endif
