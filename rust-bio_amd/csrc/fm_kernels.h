// Shared device pieces of the FM-index kernels (K5 backward search, K6 suffix-array lookup):
// the 2-bit block layout, the quad rank helper and the index handle.  Layout notes: fm_index.hip.
#ifndef BG_FM_KERNELS_H
#define BG_FM_KERNELS_H
#include <mutex>
#include <vector>

#include "bg_common.h"

namespace bgfm {

constexpr uint32_t kSymPerBlock = 192;
constexpr uint32_t kBvBits = 480;       // bits per 64-byte block of a dense symbol's rank bit vector (+ a 32-bit counter)
constexpr uint32_t kMaxExcLds = 1024;   // sparse exception positions (all of them fit in LDS)
constexpr uint32_t kJumpK = 12;  // symbols covered by the jump table of K5 (4^12 entries x 16 bytes = 256 MB)
// symbol classes (uint16 per byte value): how Occ::get(r, a) is answered for byte a
constexpr uint16_t kClsZero = 4;        // in the alphabet, never occurs in the BWT: 0
constexpr uint16_t kClsSparse = 0x100;  // + e: sparse exception symbol e — sorted position list
constexpr uint16_t kClsDense = 0x200;   // + d: dense symbol d — one-hot rank bit vector, one 64-byte block per rank
constexpr uint16_t kClsPanic = 0xFFFF;  // not in the alphabet: the reference panics (fmindex.rs:229, bwt.rs:158)
// 0..3: the symbol has a 2-bit code — rank inside the packed block stream

struct FmDev {
    const uint4* blocks;          // 2-bit stream: cnt[4] + 192 symbols per 64-byte block
    const uint4* bitvecs;         // dense symbol d: blocks [d * nbv_blocks, (d + 1) * nbv_blocks), counter + 480 bits each
    const uint32_t* exc_pos;      // all sparse exception positions, sorted (they sit in the stream as code 0)
    const uint32_t* exc_sym_pos;  // per sparse symbol, sorted, concatenated
    const uint32_t* sparse_off;   // [n_sparse + 1] ranges of exc_sym_pos
    const uint16_t* sym_class;    // [256]
    const uint32_t* less;         // [256]
    const uint8_t* bwt_raw;       // the BWT bytes (only with dense symbols: K6 reads bwt[pos] here)
    uint32_t n;
    uint32_t n_exc;               // sparse exceptions that need the code-0 correction (0 when dense symbols exist:
                                  // code 0 then belongs to no symbol at all)
    uint32_t nbv_blocks;
    uint32_t n_dense;
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
    // matches among the first t0 / t1 symbols of the two halves: the others are shifted out at the top (e has even
    // bits only; t0 >= 1, t1 may be 0: its shift goes in two halves) instead of being masked off — no select, no
    // 64-bit subtract
    const int t0 = min(have, 32), t1 = min(max(have - 32, 0), 32);
    return (uint32_t)(__popcll(e0 << (64 - 2 * t0)) + __popcll((e1 << (32 - t1)) << (32 - t1)));
}

// All four codes' shares at once (K7: a bi-interval extension wants the ranks of every symbol at both ends of the interval —
// eight block_part calls of ~28 vector instructions each were most of the kernel, round 6): the counts of this lane's symbols
// inside [0, o] with code 0, 1, 2, 3 in the four bytes of the result (a lane holds 64 symbols, the three lanes of a quad 192:
// sums over the quad fit a byte too); lane 0 of the quad (the counters) contributes 0.  Per dword of 16 symbols: the symbols
// beyond the position leave at the top, then three population counts — all bits, the low bits of the 2-bit fields, the
// fields with both bits — give codes 1, 2, 3; code 0 is what is left of the symbols that count.
__device__ __forceinline__ uint32_t block_counts4(const uint4 v, uint32_t t, uint32_t o) {
    const int have = (int)o + 1 - ((int)t - 1) * 64;  // symbols of this lane inside [0, o] (lane 0: irrelevant)
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t n_all = 0, n_low = 0, n_both = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t sh = 16u - (uint32_t)min(max(have - 16 * d, 0), 16);
        const uint32_t xt = (x[d] << sh) << sh;  // (two shifts of at most 16: the whole dword may leave)
        n_all += __popc(xt);
        n_low += __popc(xt & 0x55555555u);
        n_both += __popc(xt & (xt >> 1) & 0x55555555u);
    }
    const uint32_t kept = (uint32_t)min(max(have, 0), 64);
    const uint32_t n1 = n_low - n_both, n2 = n_all - n_low - n_both, n0 = kept - (n_all - n_both);
    return t == 0 ? 0u : n0 | n1 << 8 | n2 << 16 | n_both << 24;
}
// the four counters of a block (lane 0 of the quad holds them) in every lane of the quad
__device__ __forceinline__ uint32_t quad_lane0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
}

// block_part without branches (K5's fast kernel: the quad's lane 0 / lanes 1-3 split and the "no symbol of mine" early
// exit cost a divergent region each — exec-mask bookkeeping the straight-line form does not have): every lane computes
// both the counter select and the bitmap count and keeps the one its position in the quad calls for
__device__ __forceinline__ uint32_t block_part_bf(const uint4 v, uint32_t t, uint32_t o, uint32_t c) {
    const uint32_t lo = (c & 1) ? v.y : v.x, hi = (c & 1) ? v.w : v.z, cnt = (c & 2) ? hi : lo;
    const int have = (int)o + 1 - ((int)t - 1) * 64;  // symbols of this lane inside [0, o] (lane 0: irrelevant)
    const int t0 = min(max(have, 0), 32), t1 = min(max(have - 32, 0), 32);
    const uint64_t pat = (uint64_t)c * 0x5555555555555555ull;
    uint64_t w0 = ((uint64_t)v.y << 32) | v.x;
    uint64_t w1 = ((uint64_t)v.w << 32) | v.z;
    uint64_t e0 = ~(w0 ^ pat), e1 = ~(w1 ^ pat);
    e0 = e0 & (e0 >> 1) & 0x5555555555555555ull;
    e1 = e1 & (e1 >> 1) & 0x5555555555555555ull;
    // the symbols beyond the first t0 / t1 leave at the top: 64 - 2 t bits, in two halves (t may be 0)
    const uint32_t n = (uint32_t)(__popcll((e0 << (32 - t0)) << (32 - t0)) + __popcll((e1 << (32 - t1)) << (32 - t1)));
    return t == 0 ? cnt : n;
}

// ---- 2-step rank blocks (round 4) ---------------------------------------------------------------------------------------
// The memory system delivers random 128-byte lines at the rate of random 64-byte lines (tools/microbench/ub_gather128,
// profiles/r04_ub_gather128.json: 64 against 64 G lines/s inside the Infinity Cache, 53 against 53 at 1 GB, 51 against 51
// at 3 GB), and K5 sits at 0.9 of that rate — so the way to more queries per second is fewer requests per query: a block
// that answers TWO pattern symbols at once.  Position i of the BWT carries the pair (L[i], L[LF(i)]) — the two symbols in
// front of suffix i — as a 4-bit code (first symbol << 2 | second); one 128-byte block per 128 positions: sixteen
// counters (pairs of each code before the block) + the 128 codes bit-sliced (per 32 positions: four dwords, one per bit).  Two LF steps with symbols a then b collapse to
//   l' = C2[a][b] + Occ2(ab, l - 1),  r' = C2[a][b] + Occ2(ab, r) - 1,   C2[a][b] = less[b] + #{j < less[a] : L[j] = b}
// (the rows below less[a] + Occ(a, .) that hold b are exactly the images of the rows that hold the pair).  The double
// step is valid iff Occ2(ab, r) > Occ2(ab, l - 1); otherwise — and for the last symbol of an odd-length pattern — the
// same block answers a single step (first component == a: four counters summed, the high two bits of every nibble), so
// Partial(pl, pr, matched_len) comes out exactly as fmindex.rs:160-182 produces it.  Positions whose first or second
// symbol has no 2-bit code (the sentinel) hold 0 in that component and are listed on the side (at most kMaxExc2).
constexpr uint32_t kSym2PerBlock = 128;
constexpr uint32_t kMaxExc2 = 8;
struct Fm2Dev {
    const uint4* blocks2;        // null: no 2-step blocks (index not DNA-like, too many exceptions, or switched off)
    uint32_t c2[16];             // C2[a << 2 | b]
    uint32_t exc_pos[kMaxExc2];  // sorted
    uint8_t exc_nib[kMaxExc2];   // the nibble stored there | 16 if the FIRST component is the one without a code
    uint32_t n_exc;
};
// this lane's share of Occ2 inside one block: lane t holds counters 4t .. 4t+3 (vc) and, bit-sliced, the pair codes of
// positions 32t .. 32t+31 (vs.x/y/z/w = bit 0/1/2/3 of each code: "code == c" is three ANDs of four XORs, not a nibble
// comparison — the search is bound by vector instructions once its requests are halved).
// c: the pair code; inv: per bit of c, 0 where it is set and ~0 where it is clear (Pair2Key, once per step);
// single: only the first component (c >> 2, bits 2-3) counts — dc = ~0 makes bits 0-1 match anything
struct Pair2Key {
    uint32_t inv0, inv1, inv2, inv3, dc;
};
__device__ __forceinline__ Pair2Key pair2_key(uint32_t c, bool single) {
    Pair2Key k;
    k.inv0 = (c & 1u) - 1u;
    k.inv1 = ((c >> 1) & 1u) - 1u;
    k.inv2 = ((c >> 2) & 1u) - 1u;
    k.inv3 = ((c >> 3) & 1u) - 1u;
    k.dc = single ? ~0u : 0u;
    return k;
}
__device__ __forceinline__ uint32_t block2_part(const uint4 vc, const uint4 vs, uint32_t t, uint32_t o, uint32_t c, bool single,
                                                const Pair2Key& k) {
    const uint32_t lo = (c & 1) ? vc.y : vc.x, hi = (c & 1) ? vc.w : vc.z, one = (c & 2) ? hi : lo;
    const uint32_t cnt = single ? vc.x + vc.y + vc.z + vc.w : one;
    const uint32_t m = ((vs.x ^ k.inv0) | k.dc) & ((vs.y ^ k.inv1) | k.dc) & (vs.z ^ k.inv2) & (vs.w ^ k.inv3);
    const int h = min(max((int)o + 1 - (int)t * 32, 0), 32);  // positions of this lane inside [0, o]
    // matches among the first h positions = all matches - those from position h on (a 64-bit shift: h may be 32)
    const uint32_t n = (uint32_t)__popc(m) - (uint32_t)__popc((uint32_t)((uint64_t)m >> h));
    return n + (t == (c >> 2) ? cnt : 0u);
}

// this lane's share of rank1(o) inside one bit-vector block: lane 0 holds the counter and bits 0..95, lane t >= 1
// bits 96 + 128 (t - 1) ... + 127
__device__ __forceinline__ uint32_t bv_part(const uint4 v, uint32_t t, uint32_t o) {
    uint64_t lo, hi;
    int have;
    uint32_t acc = 0;
    if (t == 0) {
        acc = v.x;
        lo = ((uint64_t)v.z << 32) | v.y;
        hi = v.w;
        have = min((int)o + 1, 96);
    } else {
        lo = ((uint64_t)v.y << 32) | v.x;
        hi = ((uint64_t)v.w << 32) | v.z;
        have = (int)o + 1 - (96 + 128 * ((int)t - 1));
        if (have <= 0) return 0;
        have = min(have, 128);
    }
    const int h0 = min(have, 64), h1 = have - h0;  // h0 >= 1, 0 <= h1 <= 64: unwanted bits leave at the top
    const int s1 = 64 - h1, s1a = min(s1, 32);
    return acc + (uint32_t)(__popcll(lo << (64 - h0)) + __popcll((hi << s1a) << (s1 - s1a)));
}
// Occ::get(r, dense symbol d) for the quad: one 64-byte block
__device__ __forceinline__ uint4 bv_load(const FmDev& fm, uint32_t d, uint32_t r, uint32_t t, uint32_t& o) {
    const uint32_t b = r / kBvBits;
    o = r - b * kBvBits;
    return fm.bitvecs[((uint64_t)d * fm.nbv_blocks + b) * 4 + t];
}

// ---- 64-bit positions (round 5, fm_wide.hip) ------------------------------------------------------------------------------
// The reference indexes texts with usize (fmindex.rs:70-71 Interval, bwt.rs:94 Occ, suffix_array.rs:264); the layout above
// counts in uint32.  A text of 2^32 - 1 symbols or more (T$R$ of a human genome for an FMD index: 6.2 G) gets the SAME 64-byte
// blocks with counters RELATIVE to a superblock of 2^sb_shift blocks, plus one absolute 64-bit count per code and
// superblock (2^17 blocks = 25 M symbols: 8 KB of bases for 6.2 G symbols, cache-resident) — Occ::get is still one 64-byte
// line, one more load and a 64-bit add; l, r, less[] and the exception positions are 64-bit.  DNA-like BWTs only (four
// 2-bit codes + at most kMaxExcLds other positions): a text that needs rank bit vectors (protein) and 64-bit positions is
// refused with BG_ERR_UNSUPPORTED.
struct FmWideDev {
    const uint4* blocks;           // cnt[4] relative to the block's superblock + 192 symbols
    const uint64_t* sb;            // [n_superblocks][4]: occurrences of code c before the superblock
    const uint64_t* exc_pos;       // sparse exception positions, sorted
    const uint64_t* exc_sym_pos;   // per sparse symbol, sorted, concatenated
    const uint32_t* sparse_off;    // [n_sparse + 1]
    const uint16_t* sym_class;     // [256]
    const uint64_t* less;          // [256]
    uint64_t n;
    uint32_t n_exc;
    uint32_t sb_shift;             // blocks per superblock = 1 << sb_shift
};
// 2-step rank blocks on 64-bit positions (round 6): the 128-byte blocks of Fm2Dev with their sixteen counters RELATIVE to a
// superblock of 2^sb_shift blocks, and per superblock twenty absolute 64-bit bases — sixteen pair codes, then the four
// "first component == a" sums a single step adds up (one extra load per rank, from an array that stays in cache: 160 bytes
// per 2^sb_shift * 128 positions).  C2 and the exception positions are 64-bit.
struct Fm2WideDev {
    const uint4* blocks2;        // null: no 2-step blocks
    const uint64_t* sb2;         // [n_superblocks][20]
    uint64_t c2[16];
    uint64_t exc_pos[kMaxExc2];  // sorted; unused entries ~0
    uint8_t exc_nib[kMaxExc2];
    uint32_t n_exc;
    uint32_t sb_shift;
};
// what the kernels that exist for both position widths are instantiated over
template <bool WIDE>
struct FmLayout;
template <>
struct FmLayout<false> {
    using Pos = uint32_t;
    using Dev = FmDev;
    using Dev2 = Fm2Dev;
};
template <>
struct FmLayout<true> {
    using Pos = uint64_t;
    using Dev = FmWideDev;
    using Dev2 = Fm2WideDev;
};

// SEEDS: the patterns are the seed windows of a batch of reads (seed-and-extend, seed_extend.hip) — query q is
// seed q % S of read q / S: pat[pat_off[r] + k * stride ..+ seed_len) while it fits in the read (else an empty
// pattern: Absent), so overlapping windows need no copy of the reads.
struct SeedSrc {
    uint32_t S, stride, seed_len;
    uint32_t code_bytes;            // PACKED: the byte value of each 2-bit code (code c in bits 8c..8c+7)
    unsigned long long* lines;      // COUNT: receives the number of 64-byte block loads the launch issued
};
constexpr uint8_t kTagDeferred = 0xFF;  // the fast kernels leave such a query to the generic kernel launched behind them
static_assert(kTagDeferred > BG_FM_PANIC, "the deferral mark must not collide with a BG_FM_* tag");
constexpr uint32_t kFastSyms = 256;     // symbols of a pattern slot of the fast kernels (LDS)

__device__ __forceinline__ uint32_t count_le64(const uint64_t* arr, uint32_t lo, uint32_t hi, uint64_t r) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (arr[mid] <= r)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

}  // namespace bgfm

struct bg_fm {
    bg_ctx* ctx = nullptr;
    bgfm::FmDev dev = {};
    bgfm::Fm2Dev dev2 = {};      // 2-step rank blocks (fm_kernels.h), built behind the index by fm_build_step2
    void* d_blocks2 = nullptr;
    bool no_step2 = false;       // option "no_step2": searches take single steps only (tests, A/B)
    int ilp = 2;                 // option "ilp": queries per quad of the search (1: fm_search_fast_kernel / fmw_search_kernel; 2: the 2x kernels)
    void* d_blocks = nullptr;
    void* d_exc_pos = nullptr;
    void* d_exc_sym_pos = nullptr;
    void* d_class = nullptr;
    void* d_less = nullptr;
    void* d_exc_byte = nullptr;  // byte value of every exception, parallel to exc_pos
    void* d_sparse_off = nullptr;
    void* d_bitvecs = nullptr;
    void* d_bwt_raw = nullptr;
    void* d_text = nullptr;      // the text the index was built from (bg_fm_set_text): seed-and-extend cuts its windows here
    bool text_owned = false;
    uint64_t n_text = 0;
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
    bool no_fast = false;    // tests: never the pack-at-fetch kernel (fm_search_fast_kernel)
    int n_codes = 0;         // distinct bytes with a 2-bit code (<= 4)
    uint32_t less_len = 0;
    bool fmd_ok = false;  // the BWT is a word over dna::n_alphabet() + '$' (FMDIndex::from, fmindex.rs:323-327)
    uint16_t h_class[256] = {};  // host copy of the symbol classes (K7 picks its plain-DNA instantiation from them)
    // 64-bit positions (fm_wide.hip): `wdev` instead of `dev` / `dev2`; the suffix array attached to it is uint64
    bool wide = false;
    bgfm::FmWideDev wdev = {};
    void* d_sb = nullptr;
    bgfm::Fm2WideDev wdev2 = {};  // 2-step blocks on 64-bit positions (d_blocks2 holds the blocks)
    void* d_sb2 = nullptr;
    // what the handle was built from, for bg_fm_save (fm_persist.hip): the reference's FMIndex holds the same
    std::vector<uint8_t> alphabet;
    std::vector<uint64_t> h_less;
    uint32_t occ_k = 0;
};
int fm_decode_bwt_dev(const bg_fm* fm, uint8_t* d_out, hipStream_t st);  // fm_persist.hip: the handle's BWT bytes
void fm_remember_inputs(bg_fm* fm, const uint8_t* alphabet, uint32_t n_sym, uint32_t occ_k, const uint64_t* less, uint32_t less_len);

// fm_wide.hip: the index with 64-bit positions (built from a BWT in HBM; `less` null: the BWT's own cumulative counts)
int fm_wide_build_dev(bg_ctx* ctx, const uint8_t* d_bwt, uint64_t n, const uint8_t* alphabet, uint32_t n_sym, const uint64_t* less,
                      uint32_t less_len, uint64_t* less_out, bg_fm** out, hipStream_t st);
// `seeds` non-null: the SEEDS flavour (n_q = reads * S; S, stride, seed_len set); `packed`: pat is a 2-bit stream, offsets in symbols
int fm_wide_search_dev(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat, const uint64_t* d_pat_off, uint8_t* d_tag, uint64_t* d_lower,
                       uint64_t* d_upper, uint32_t* d_matched_len, hipStream_t st, const bgfm::SeedSrc* seeds = nullptr, bool packed = false);
// fm_index.hip: the 2x fast kernel instantiated for 64-bit positions (needs fm->wdev2.blocks2); deferred queries are left tagged
int fm_wide_fast2x_launch(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat, const uint64_t* d_pat_off, uint8_t* d_tag, uint64_t* d_lower,
                          uint64_t* d_upper, uint32_t* d_matched_len, hipStream_t st, const bgfm::SeedSrc* seeds, bool packed);
// fm_step2.hip: the 2-step blocks behind a finished 64-bit index (best effort; synchronises the stream)
void fm_build_step2_wide(bg_fm* fm, hipStream_t st);
int fm_wide_sa_get(bg_fm* fm, uint64_t n, const uint64_t* d_index, uint64_t* d_pos, hipStream_t st);
// texts from this many symbols on take the 64-bit layout (tests lower it through the ctx option "fm_wide_from")
uint64_t fm_wide_threshold(const bg_ctx* ctx);

// fm_step2.hip: builds fm->dev2 behind a finished index (best effort; synchronises the stream)
void fm_build_step2(bg_fm* fm, hipStream_t st);
// internal entry points shared between the FM translation units
int bg_fm_search_seeds_dev(bg_fm* fm, uint64_t n_reads, const uint8_t* d_reads, const uint64_t* d_read_off, uint32_t S,
                           uint32_t stride, uint32_t seed_len, uint8_t* d_tag, uint64_t* d_lower, uint64_t* d_upper,
                           uint32_t* d_matched_len, hipStream_t st);

#endif
