// K7 — FMD-index bi-directional search on gfx950: batched `FMDIndex::smems` / `all_smems`
// (/root/reference/src/data_structures/fmindex.rs:363-501) with `backward_ext` / `forward_ext` /
// `init_interval_with` (504-564) on the same 2-bit block index as K5.
//
// One quad of 4 lanes per pattern (persistent, like K5).  An extension needs Occ for up to eleven
// symbols at two rows; with the block layout that is two 64-byte loads: the quad derives the counts
// of all four codes of each block at once (4 + 4 quad reductions), the remaining symbols ('$', N,
// lower case) come from the exception lists or are zero.  The curr/prev interval lists of `smems`
// live in a per-quad slice of global scratch (16 bytes per entry).
#include <algorithm>
#include <type_traits>

#include "fm_kernels.h"

using namespace bgfm;

namespace {

// Round 6: every kernel here is a template over the index layout (FmLayout<WIDE>: uint32 or uint64 positions) and over the
// width of the records it writes (OUT64) — the reference's BiInterval is usize throughout (fmindex.rs:254-259), and T$R$ of
// a human genome (6.2 G symbols) is what the 64-bit layout exists for.  <false, false> is the kernel of rounds 2-5.
template <typename P>
struct BiIvT {  // BiInterval (fmindex.rs:254-259) + the match length smems tracks next to it
    P lower, lower_rev, size;
    uint32_t msz;   // match_size
    uint32_t mlen;  // match_len of the (interval, match_len) pairs in curr / prev
};

template <bool WIDE>
struct FmdArgsT {
    typename FmLayout<WIDE>::Dev fm;
    uint32_t less_len;
    uint64_t n_p;
    const uint8_t* pat;
    const uint64_t* pat_off;
    const uint32_t* i_pos;  // smems: position every pattern's matches must overlap; nullptr with all != 0
    uint32_t min_len;
    int all;
    uint32_t cap;           // output records per pattern
    uint32_t* out_count;    // [n_p] records found (may exceed cap), 0xFFFFFFFF: the reference would panic
    void* out;              // [n_p * cap * 6] uint32 or uint64: lower, lower_rev, size, match_size, position, length
    uint4* lists;           // per quad slot: 2 * list_cap entries (of one uint4, two on 64-bit positions)
    uint32_t list_cap;
};


__device__ __forceinline__ uint64_t k7_base(const FmDev&, uint64_t, uint32_t) { return 0; }
__device__ __forceinline__ uint64_t k7_base(const FmWideDev& fm, uint64_t blk, uint32_t c) { return fm.sb[(blk >> fm.sb_shift) * 4 + c]; }
__device__ __forceinline__ uint32_t k7_count_le(const uint32_t* a, uint32_t lo, uint32_t hi, uint32_t r) { return count_le(a, lo, hi, r); }
__device__ __forceinline__ uint32_t k7_count_le(const uint64_t* a, uint32_t lo, uint32_t hi, uint64_t r) { return count_le64(a, lo, hi, r); }
__device__ __forceinline__ uint32_t k7_dense(const FmDev& fm, uint32_t d, uint32_t r, uint32_t t) {
    uint32_t o;
    const uint4 v = bv_load(fm, d, r, t, o);
    return quad_sum(bv_part(v, t, o));
}
__device__ __forceinline__ uint32_t k7_dense(const FmWideDev&, uint32_t, uint64_t, uint32_t) { return 0; }  // (no dense symbols there)

// PLAIN: 0 the general extension, 1 plain DNA (straight line), 2 plain DNA whose '$' occurs at most twice and is the only listed
// symbol (T$, T$R$: its rows in registers) — chosen by the host from the index's classes (smems_dev)
template <bool WIDE, int PLAIN = 0>
struct Ctx {
    using P = typename FmLayout<WIDE>::Pos;
    using BiIv = BiIvT<P>;
    const FmdArgsT<WIDE>& a;
    const uint16_t* s_class;
    const P* s_less;
    const uint8_t* s_comp;
    const P* s_exc;
    const P* s_exc_sym;          // the sparse symbols' own position lists (LDS copies of fm.exc_sym_pos / fm.sparse_off: every
    const uint32_t* s_sparse_off;  // extension ranks '$' first, fmindex.rs:536 — from global memory that was a chain of four to
    uint32_t t;                  // six dependent round trips per extension, round 6)
    bool panic;
    // "plain DNA" (wavefront-uniform, found once per kernel): T, G, C, A have 2-bit codes, '$' is a listed symbol (or absent),
    // N and the lower-case letters never occur in the BWT — every index over an ACGT text.  plain_codes: the codes of T, G, C, A
    // in bits 0-1, 2-3, 4-5, 6-7; plain_dollar: '$' 's list index, 0xFFFF if it never occurs; 0xFFFFFFFF in plain_codes: not plain
    uint32_t plain_codes, plain_dollar;
    const uint8_t* s_pos;        // position of a byte in the order "$TGCNAtgcna" (fmindex.rs:536); 10 for 'a' and any other byte
    // plain DNA whose only listed symbol is '$' with at most two occurrences (T$, T$R$): its rows, "no such row" = the largest P —
    // the rank of '$' (also the number of exceptions in front of a row) is two comparisons instead of four binary searches over
    // LDS lists per extension.  small_dollar false: the searches.
    P dollar_row[2] = {~(P)0, ~(P)0};

    __device__ uint32_t exc_le(P r) const { return k7_count_le(s_exc, 0u, a.fm.n_exc, r); }
    __device__ P less_of(uint32_t s) {
        if (s >= a.less_len) {  // index out of bounds in the reference
            panic = true;
            return 0;
        }
        return s_less[s];
    }
    __device__ P occ_cls(uint32_t cls, P r, const P c[4]) {
        if (cls < 4) return c[cls];
        if (cls == kClsPanic) {
            panic = true;
            return 0;
        }
        if (cls >= kClsDense) return k7_dense(a.fm, cls - kClsDense, r, t);  // a genome with many N: ranked in bit vectors
        if (cls >= kClsSparse) {
            const uint32_t e = cls - kClsSparse;
            const uint32_t lo = s_sparse_off[e], hi = s_sparse_off[e + 1];
            return k7_count_le(s_exc_sym, lo, hi, r) - lo;
        }
        return 0;  // in the alphabet, never in the BWT
    }
    // the straight-line end of a plain-DNA extension: occ of '$' is in occR[0] / occL[0]; T, G, C, (N: 0,) A from the codes' ranks.
    __device__ __forceinline__ void plain_tail(const P (&cR)[4], const P (&cL)[4], P (&occR)[6], P (&occL)[6], bool has_l, const BiIv& iv, uint32_t sym, BiIv& r) {
        auto pick = [&](const P (&c)[4], uint32_t code) -> P { return code == 0 ? c[0] : code == 1 ? c[1] : code == 2 ? c[2] : c[3]; };
#pragma unroll
        for (int k = 0; k < 3; k++) {  // T, G, C
            occR[1 + k] = pick(cR, (plain_codes >> (2 * k)) & 3u);
            occL[1 + k] = pick(cL, (plain_codes >> (2 * k)) & 3u);
        }
        occR[5] = pick(cR, (plain_codes >> 6) & 3u);  // A
        occL[5] = pick(cL, (plain_codes >> 6) & 3u);
        occR[4] = occL[4] = 0;  // N
        // sizes in the order of fmindex.rs:536, their running sum in front of every symbol, and the lane's own symbol's entries
        const uint32_t pos = s_pos[sym & 0xFFu];
        P sz[6], before[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (!has_l) occL[k] = 0;
            sz[k] = occR[k] - occL[k];
        }
        before[0] = 0;
#pragma unroll
        for (int k = 1; k < 6; k++) before[k] = before[k - 1] + sz[k - 1];
        P o2 = 0, s2 = 0, b2 = before[5] + sz[5];  // (a position behind the order's six: everything is in front, the size is 0)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const bool here = (uint32_t)k == pos;
            o2 = here ? occL[k] : o2;
            s2 = here ? sz[k] : s2;
            b2 = here ? before[k] : b2;
        }
        r.lower = less_of(sym) + o2;
        r.lower_rev = iv.lower_rev + b2;
        r.size = s2;
        r.msz = iv.msz + 1;
    }
    // fmindex.rs:527-558
    __device__ BiIv backward_ext(const BiIv& iv, uint32_t sym) {
        BiIv r = iv;
        if (iv.lower + iv.size == 0) {  // usize underflow of lower + size - 1
            panic = true;
            return r;
        }
        // both rows' blocks are requested before either is looked at (one round trip, not two: the second load used to sit
        // behind `if (iv.lower > 0)` and the first one's reductions); lower == 0 reads row 0's block and drops the counts
        const P posR = iv.lower + iv.size - 1, posL = iv.lower > 0 ? iv.lower - 1 : 0;
        const uint64_t bR = (uint64_t)(posR / kSymPerBlock), bL = (uint64_t)(posL / kSymPerBlock);
        const uint4 vR = a.fm.blocks[bR * 4 + t], vL = a.fm.blocks[bL * 4 + t];
        const uint32_t oR = (uint32_t)(posR - (P)bR * kSymPerBlock), oL = (uint32_t)(posL - (P)bL * kSymPerBlock);
        P cR[4], cL[4];
        {  // ranks of the four codes at both positions: block counters + the quad's symbols up to the position (block_counts4)
            const uint32_t pkR = quad_sum(block_counts4(vR, t, oR)), pkL = quad_sum(block_counts4(vL, t, oL));
            const uint32_t ctrR[4] = {quad_lane0(vR.x), quad_lane0(vR.y), quad_lane0(vR.z), quad_lane0(vR.w)};
            const uint32_t ctrL[4] = {quad_lane0(vL.x), quad_lane0(vL.y), quad_lane0(vL.z), quad_lane0(vL.w)};
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                cR[k] = (P)k7_base(a.fm, bR, k) + ctrR[k] + ((pkR >> (8 * k)) & 0xFFu);
                cL[k] = (P)k7_base(a.fm, bL, k) + ctrL[k] + ((pkL >> (8 * k)) & 0xFFu);
            }
        }
        const bool has_l = iv.lower > 0;
        if constexpr (PLAIN == 2) {
            const P dR = (P)(dollar_row[0] <= posR) + (P)(dollar_row[1] <= posR), dL = (P)(dollar_row[0] <= posL) + (P)(dollar_row[1] <= posL);
            cR[0] -= dR;  // exceptions sit in the stream as code 0
            cL[0] -= dL;
            P occR[6], occL[6];
            occR[0] = dR;
            occL[0] = dL;
            plain_tail(cR, cL, occR, occL, has_l, iv, sym, r);
            return r;
        }
        if (a.fm.n_exc) {  // exceptions sit in the stream as code 0
            cR[0] -= exc_le(posR);
            cL[0] -= exc_le(posL);
        }
        if constexpr (PLAIN == 1) {
            // Plain DNA, straight line: the sizes of '$', T, G, C, (N: 0,) A in the order of fmindex.rs:536, a prefix sum up to
            // the lane's symbol.  The general loop below spends ~800 vector + scalar instructions per extension on eleven
            // per-lane class dispatches (SQ counters, profiles/r06_sq_fmd.txt); this is ~100.
            P occR[6], occL[6];
            occR[0] = occL[0] = 0;
            if (plain_dollar != 0xFFFFu) {
                const uint32_t lo = s_sparse_off[plain_dollar], hi = s_sparse_off[plain_dollar + 1];
                occR[0] = k7_count_le(s_exc_sym, lo, hi, posR) - lo;
                occL[0] = k7_count_le(s_exc_sym, lo, hi, posL) - lo;
            }
            plain_tail(cR, cL, occR, occL, has_l, iv, sym, r);
            return r;
        }
        P s = 0, o = 0, l = iv.lower_rev;
        // (A variant with the eleven classes as wavefront-uniform scalars and a per-lane select by the symbol's position in the
        //  order was built and measured, round 6: 131 - 139 VGPRs, three wavefronts per SIMD instead of six, 2.0 M reads/s
        //  against 2.4 M for this form.)
        constexpr uint8_t kOrder[11] = {'$', 'T', 'G', 'C', 'N', 'A', 't', 'g', 'c', 'n', 'a'};  // fmindex.rs:536
#pragma unroll
        for (int idx = 0; idx < 11; idx++) {
            const uint32_t b = kOrder[idx];
            const uint32_t cls = s_class[b];
            l += s;
            o = has_l ? occ_cls(cls, posL, cL) : (P)0;
            s = occ_cls(cls, posR, cR) - o;
            if (b == sym) break;
        }
        r.lower = less_of(sym) + o;
        r.lower_rev = l;
        r.size = s;
        r.msz = iv.msz + 1;
        return r;
    }
    // fmindex.rs:560-564
    __device__ BiIv forward_ext(const BiIv& iv, uint32_t sym) {
        BiIv sw = iv;
        sw.lower = iv.lower_rev;
        sw.lower_rev = iv.lower;
        BiIv e = backward_ext(sw, s_comp[sym]);
        const P lo = e.lower;
        e.lower = e.lower_rev;
        e.lower_rev = lo;
        return e;
    }
    // fmindex.rs:504-514
    __device__ BiIv init_interval_with(uint32_t sym) {
        BiIv r;
        r.lower = less_of(sym);
        r.lower_rev = less_of(s_comp[sym]);
        r.size = less_of(sym + 1) - r.lower;
        r.msz = 1;
        r.mlen = 0;
        return r;
    }
};

// what backward_ext's straight-line form needs to know about the index (see Ctx::plain_codes), and the order table
__device__ __forceinline__ void k7_plain(const uint16_t* s_class, uint8_t* s_pos, uint32_t& plain_codes, uint32_t& plain_dollar) {
    const char* order = "$TGCNAtgcna";  // fmindex.rs:536
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        uint8_t at = 10;
        for (int k = 0; k < 11; k++)
            if (i == (uint32_t)(uint8_t)order[k]) at = (uint8_t)k;
        s_pos[i] = at;
    }
    const uint32_t cT = s_class['T'], cG = s_class['G'], cC = s_class['C'], cA = s_class['A'], cD = s_class['$'];
    bool plain = cT < 4 && cG < 4 && cC < 4 && cA < 4 && (cD == kClsZero || (cD >= kClsSparse && cD < kClsDense));
    for (const char* z = "Ntgcna"; *z; z++) plain = plain && s_class[(uint8_t)*z] == kClsZero;
    uint32_t codes = plain ? (cT | cG << 2 | cC << 4 | cA << 6) : 0xFFFFFFFFu;
    uint32_t dollar = cD == kClsZero ? 0xFFFFu : cD - kClsSparse;
    plain_codes = (uint32_t)__builtin_amdgcn_readfirstlane((int)codes);
    plain_dollar = (uint32_t)__builtin_amdgcn_readfirstlane((int)dollar);
    __syncthreads();
}

// list entries: one uint4 per interval on 32-bit positions, two on 64-bit
__device__ __forceinline__ void list_put(uint4* list, uint32_t i, const BiIvT<uint32_t>& v) {
    list[i] = make_uint4(v.lower, v.lower_rev, v.size, v.msz << 16 | v.mlen);
}
__device__ __forceinline__ void list_put(uint4* list, uint32_t i, const BiIvT<uint64_t>& v) {
    list[2 * i] = make_uint4((uint32_t)v.lower, (uint32_t)(v.lower >> 32), (uint32_t)v.lower_rev, (uint32_t)(v.lower_rev >> 32));
    list[2 * i + 1] = make_uint4((uint32_t)v.size, (uint32_t)(v.size >> 32), v.msz << 16 | v.mlen, 0u);
}
__device__ __forceinline__ void list_get(const uint4* list, uint32_t i, BiIvT<uint32_t>& v) {
    const uint4 u = list[i];
    v = BiIvT<uint32_t>{u.x, u.y, u.z, u.w >> 16, u.w & 0xFFFFu};
}
__device__ __forceinline__ void list_get(const uint4* list, uint32_t i, BiIvT<uint64_t>& v) {
    const uint4 u = list[2 * i], w = list[2 * i + 1];
    v = BiIvT<uint64_t>{(uint64_t)u.y << 32 | u.x, (uint64_t)u.w << 32 | u.z, (uint64_t)w.y << 32 | w.x, w.z >> 16, w.z & 0xFFFFu};
}

template <bool WIDE>
__device__ __forceinline__ void k7_tables(const FmdArgsT<WIDE>& a, uint16_t* s_class, typename FmLayout<WIDE>::Pos* s_less, uint8_t* s_comp,
                                          typename FmLayout<WIDE>::Pos* s_exc, typename FmLayout<WIDE>::Pos* s_exc_sym, uint32_t* s_sparse_off) {
    // (sparse symbols: at most 256 of them, their positions together are the n_exc <= kMaxExcLds exception positions; an index
    //  with dense symbols lists only the sparse ones here)
    for (uint32_t i = threadIdx.x; i < 257; i += blockDim.x) s_sparse_off[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        const uint32_t c = a.fm.sym_class[i];
        if (c >= kClsSparse && c < kClsDense) {
            const uint32_t e = c - kClsSparse;
            s_sparse_off[e] = a.fm.sparse_off[e];
            s_sparse_off[e + 1] = a.fm.sparse_off[e + 1];  // (neighbours write the same value)
        }
    }
    __syncthreads();
    {
        uint32_t n_sym_pos = 0;
        for (uint32_t i = 0; i < 257; i++) n_sym_pos = max(n_sym_pos, s_sparse_off[i]);
        for (uint32_t i = threadIdx.x; i < n_sym_pos && i < kMaxExcLds; i += blockDim.x) s_exc_sym[i] = a.fm.exc_sym_pos[i];
    }
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        s_class[i] = a.fm.sym_class[i];
        s_less[i] = a.fm.less[i];
        uint8_t c = (uint8_t)i;  // dna::complement (alphabets/dna.rs:37-69)
        const char* from = "AGCTYRWSKMDVHBN";
        const char* to = "TCGARYWSMKHBDVN";
        for (int k = 0; k < 15; k++) {
            if (i == (uint32_t)from[k]) c = (uint8_t)to[k];
            if (i == (uint32_t)from[k] + 32) c = (uint8_t)(to[k] + 32);
        }
        s_comp[i] = c;
    }
    for (uint32_t i = threadIdx.x; i < a.fm.n_exc; i += blockDim.x) s_exc[i] = a.fm.exc_pos[i];  // n_exc <= kMaxExcLds
    __syncthreads();
}

// Round 6: the walk as a STATE MACHINE with ONE extension site.  smems (fmindex.rs:363-434) is a forward pass of forward_ext
// calls, then for every position to the left a pass of backward_ext calls over a list of intervals; all_smems (479-501) calls
// it again and again.  Written as nested loops, the sixteen quads of a wavefront — each somewhere else in its own read — ran
// the forward loop's extension and the backward loop's one after the other, every quad waiting through both round trips per
// step.  Here a quad's position in those loops is a phase + a few counters, every trip of the wavefront's loop takes each
// quad to its next extension (transitions that need none are chained in front), all quads then make their extension at the
// same instruction — forward_ext is backward_ext on the swapped interval with the complement (560-564) — and finish their
// step.  The read's bytes sit in LDS (reads up to kPatLds symbols), the next list entry is fetched beside the extension.
constexpr uint32_t kPatLds = 248;  // symbols of a read kept in LDS (64 quads x 248 B = 15.5 KB); longer reads are read in place

#ifndef K7_WAVES  // wavefronts per SIMD the kernel is compiled for (0: the compiler's choice; tools/exp/ko_build.sh variants)
#define K7_WAVES 0
#endif
#if K7_WAVES
#define K7_OCC __attribute__((amdgpu_waves_per_eu(K7_WAVES, K7_WAVES)))
#else
#define K7_OCC
#endif
template <bool WIDE, bool OUT64, int PLAIN>
__global__ __launch_bounds__(256) K7_OCC void fmd_smems_kernel(const FmdArgsT<WIDE> a) {
    using P = typename FmLayout<WIDE>::Pos;
    using BiIv = BiIvT<P>;
    using O = std::conditional_t<OUT64, uint64_t, uint32_t>;
    static_assert(OUT64 || !WIDE, "64-bit positions need 64-bit records");
    constexpr uint32_t LW = WIDE ? 2 : 1;  // uint4 per list entry
    __shared__ uint16_t s_class[256];
    __shared__ P s_less[256];
    __shared__ uint8_t s_comp[256];
    __shared__ P s_exc[kMaxExcLds];
    __shared__ P s_exc_sym[kMaxExcLds];
    __shared__ uint32_t s_sparse_off[257];
    __shared__ uint8_t s_pat[64 * kPatLds];
    __shared__ uint8_t s_pos[256];
    k7_tables<WIDE>(a, s_class, s_less, s_comp, s_exc, s_exc_sym, s_sparse_off);

    const uint32_t t = threadIdx.x & 3;
    const uint64_t slot = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const uint64_t n_slots = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint4* const list0 = a.lists + slot * 2 * a.list_cap * LW;
    uint4* const list1 = list0 + (uint64_t)a.list_cap * LW;
    uint8_t* const my_pat = s_pat + (threadIdx.x >> 2) * kPatLds;
    Ctx<WIDE, PLAIN> cx{a, s_class, s_less, s_comp, s_exc, s_exc_sym, s_sparse_off, t, false, 0xFFFFFFFFu, 0xFFFFu, s_pos};
    k7_plain(s_class, s_pos, cx.plain_codes, cx.plain_dollar);  // (PLAIN: the host looked at the same classes, bg_fm::h_class)
    if (PLAIN == 2) {  // '$' the only listed symbol, at most twice (the host checked): its rows in registers (Ctx::dollar_row)
        const uint32_t lo = cx.plain_dollar != 0xFFFFu ? s_sparse_off[cx.plain_dollar] : 0u;
        const uint32_t nd = cx.plain_dollar != 0xFFFFu ? s_sparse_off[cx.plain_dollar + 1] - lo : 0u;
        if (nd > 0) cx.dollar_row[0] = s_exc_sym[lo];
        if (nd > 1) cx.dollar_row[1] = s_exc_sym[lo + 1];
    }

    enum : uint32_t { PH_LOAD, PH_START, PH_FWD, PH_K, PH_BWD, PH_AFTER, PH_FINISH, PH_DONE };
    uint32_t phase = PH_LOAD;
    uint64_t q = slot;
    const uint8_t* pattern = nullptr;
    bool pat_lds = false;
    uint32_t plen = 0, i0 = 0, reach = 0, n_out = 0;
    O* out = nullptr;
    // forward pass
    BiIv interval{};
    uint32_t match_len = 0, p = 0;
    // lists
    bool flip = false;  // false: curr = list0, prev = list1
    uint32_t n_curr = 0, n_prev = 0;
    // backward passes
    int32_t k = 0, j = 0;
    uint32_t e = 0, sym_k = 0;
    bool have_last = false, prev_reversed = false, have_next = false;
    P last_size = 0;
    BiIv pv_next{};

    auto sym_at = [&](uint32_t pos) -> uint32_t { return pat_lds ? (uint32_t)my_pat[pos] : (uint32_t)pattern[pos]; };
    auto curr_list = [&]() { return flip ? list1 : list0; };
    auto prev_list = [&]() { return flip ? list0 : list1; };
    auto emit = [&](const BiIv& iv, uint32_t pos, uint32_t len) {
        if (t == 0 && n_out < a.cap) {
            O* o = out + (uint64_t)n_out * 6;
            o[0] = (O)iv.lower;
            o[1] = (O)iv.lower_rev;
            o[2] = (O)iv.size;
            o[3] = iv.msz;
            o[4] = pos;
            o[5] = len;
        }
        n_out++;
    };
    auto end_forward = [&]() {  // the loop at fmindex.rs:381-394 is over: its last interval, then the lists change roles
        interval.mlen = match_len;
        if (t == 0) list_put(curr_list(), n_curr, interval);
        n_curr++;
        flip = !flip;  // "reverse intervals such that longest comes first": prev is read back to front instead
        n_prev = n_curr;
        prev_reversed = true;
        j = (int32_t)plen;
        k = (int32_t)i0 - 1;
        phase = PH_K;
    };

    for (;;) {
        // ---- transitions that need no extension, until the quad stands in front of one (or is done)
        for (;;) {
            if (phase == PH_FWD && !cx.panic && p < plen) break;
            if (phase == PH_BWD && !cx.panic && e < n_prev) break;
            if (phase == PH_DONE) break;
            if (phase == PH_LOAD) {
                if (q >= a.n_p) {
                    phase = PH_DONE;
                    continue;
                }
                const uint64_t off = a.pat_off[q];
                plen = (uint32_t)(a.pat_off[q + 1] - off);
                pattern = a.pat + off;
                pat_lds = plen <= kPatLds;
                if (pat_lds)
                    for (uint32_t c = t; c < plen; c += 4) my_pat[c] = pattern[c];  // (the quad's lanes: same wavefront, LDS in order)
                cx.panic = false;
                n_out = 0;
                out = (O*)a.out + q * (uint64_t)a.cap * 6;
                phase = PH_START;
                if (plen == 0) {
                    if (!a.all) cx.panic = true;  // pattern[i] on an empty pattern
                    phase = PH_FINISH;
                } else if (a.all) {  // fmindex.rs:479-501
                    i0 = 0;
                    reach = 1;
                } else {
                    i0 = a.i_pos[q];
                    reach = 0;
                    if (i0 >= plen) {
                        cx.panic = true;
                        phase = PH_FINISH;
                    }
                }
            } else if (phase == PH_START) {  // one smems(pattern, i0, l) call (fmindex.rs:363-434)
                flip = false;
                n_curr = 0;
                match_len = 0;
                interval = cx.init_interval_with(sym_at(i0));
                if (interval.size != 0) match_len += 1;
                p = i0 + 1;
                phase = PH_FWD;
            } else if (phase == PH_FWD) {  // (p == plen, or the reference panicked)
                if (cx.panic)
                    phase = PH_FINISH;
                else
                    end_forward();
            } else if (phase == PH_K) {  // the head of the loop at fmindex.rs:401
                if (k < -1 || cx.panic) {
                    phase = PH_AFTER;
                } else {
                    sym_k = k == -1 ? (uint32_t)'$' : sym_at((uint32_t)k);
                    n_curr = 0;
                    have_last = false;
                    e = 0;
                    have_next = false;
                    phase = PH_BWD;
                }
            } else if (phase == PH_BWD) {  // (e == n_prev: the end of the inner loop, fmindex.rs:425-431)
                if (cx.panic || n_curr == 0) {
                    phase = PH_AFTER;
                } else {
                    flip = !flip;
                    n_prev = n_curr;
                    prev_reversed = false;
                    k--;
                    phase = PH_K;
                }
            } else if (phase == PH_AFTER) {  // smems returned
                if (a.all && !cx.panic && reach < plen) {
                    i0 = reach;
                    reach = i0 + 1;
                    phase = PH_START;
                } else {
                    phase = PH_FINISH;
                }
            } else {  // PH_FINISH
                if (t == 0) a.out_count[q] = cx.panic ? 0xFFFFFFFFu : n_out;
                q += n_slots;
                phase = PH_LOAD;
            }
        }
        if (!__any(phase != PH_DONE)) break;
        if (phase == PH_DONE) continue;
        // ---- the operands of this quad's extension
        const bool fwd_step = phase == PH_FWD;
        BiIv in, pv{};
        uint32_t sym;
        if (fwd_step) {  // forward_ext(interval, a) = backward_ext(swapped, complement(a)).swapped (fmindex.rs:560-564)
            in = interval;
            in.lower = interval.lower_rev;
            in.lower_rev = interval.lower;
            sym = s_comp[sym_at(p)];
        } else {
            const uint4* pl = prev_list();
            if (have_next)
                pv = pv_next;
            else
                list_get(pl, prev_reversed ? n_prev - 1 - e : e, pv);
            have_next = e + 1 < n_prev;
            if (have_next) list_get(pl, prev_reversed ? n_prev - 2 - e : e + 1, pv_next);  // (in flight beside the extension's blocks)
            in = pv;
            sym = sym_k;
        }
        // ---- the one extension site
        BiIv r = cx.backward_ext(in, sym);
        // ---- the rest of the step
        if (fwd_step) {
            BiIv fwd = r;
            fwd.lower = r.lower_rev;
            fwd.lower_rev = r.lower;
            if (interval.size != fwd.size) {
                interval.mlen = match_len;
                if (t == 0) list_put(curr_list(), n_curr, interval);
                n_curr++;
            }
            if (fwd.size == 0) {
                if (cx.panic)
                    phase = PH_FINISH;
                else
                    end_forward();
            } else {
                interval = fwd;
                match_len += 1;
                p++;
            }
        } else {
            if ((r.size == 0 || k == -1) && n_curr == 0 && k < j && pv.mlen >= a.min_len) {
                j = k;
                emit(pv, (uint32_t)(k + 1), pv.mlen);
                reach = max(reach, (uint32_t)(k + 1) + pv.mlen);
            }
            if (r.size != 0 && !(have_last && r.size == last_size)) {
                have_last = true;
                last_size = r.size;
                r.mlen = pv.mlen + 1;
                if (t == 0) list_put(curr_list(), n_curr, r);
                n_curr++;
            }
            e++;
        }
    }
}

// One request per quad: op 0 init_interval (fmindex.rs:517-524), 1 init_interval_with(a) (504-514),
// 2 backward_ext(iv, a) (527-558), 3 forward_ext(iv, a) (560-564).  iv / out: lower, lower_rev, size, match_size.
template <bool WIDE, bool OUT64>
__global__ __launch_bounds__(256) void fmd_interval_kernel(FmdArgsT<WIDE> a, uint64_t n_req, const uint8_t* op, const void* iv_in_,
                                                           const uint8_t* sym, void* iv_out_, uint8_t* ok) {
    using P = typename FmLayout<WIDE>::Pos;
    using BiIv = BiIvT<P>;
    using O = std::conditional_t<OUT64, uint64_t, uint32_t>;
    __shared__ uint16_t s_class[256];
    __shared__ P s_less[256];
    __shared__ uint8_t s_comp[256];
    __shared__ P s_exc[kMaxExcLds];
    __shared__ P s_exc_sym[kMaxExcLds];
    __shared__ uint32_t s_sparse_off[257];
    __shared__ uint8_t s_pos[256];
    k7_tables<WIDE>(a, s_class, s_less, s_comp, s_exc, s_exc_sym, s_sparse_off);
    uint32_t pc = 0xFFFFFFFFu, pd = 0xFFFFu;
    k7_plain(s_class, s_pos, pc, pd);
    const O* iv_in = (const O*)iv_in_;
    O* iv_out = (O*)iv_out_;
    const uint32_t t = threadIdx.x & 3;
    const uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    if (q >= n_req) return;  // quad-uniform
    Ctx<WIDE, false> cx{a, s_class, s_less, s_comp, s_exc, s_exc_sym, s_sparse_off, t, false, pc, pd, s_pos};
    BiIv in{(P)iv_in[4 * q], (P)iv_in[4 * q + 1], (P)iv_in[4 * q + 2], (uint32_t)iv_in[4 * q + 3], 0};
    BiIv r = in;
    switch (op[q]) {
        case 0: r = BiIv{0, 0, a.fm.n, 0, 0}; break;
        case 1: r = cx.init_interval_with(sym[q]); break;
        case 2: r = cx.backward_ext(in, sym[q]); break;
        default: r = cx.forward_ext(in, sym[q]); break;
    }
    if (t == 0) {
        iv_out[4 * q] = (O)r.lower;
        iv_out[4 * q + 1] = (O)r.lower_rev;
        iv_out[4 * q + 2] = (O)r.size;
        iv_out[4 * q + 3] = r.msz;
        ok[q] = cx.panic ? 0 : 1;
    }
}

template <bool WIDE>
FmdArgsT<WIDE> k7_args(const bg_fm* fm);
template <>
FmdArgsT<false> k7_args<false>(const bg_fm* fm) {
    FmdArgsT<false> a = {};
    a.fm = fm->dev;
    a.less_len = fm->less_len;
    return a;
}
template <>
FmdArgsT<true> k7_args<true>(const bg_fm* fm) {
    FmdArgsT<true> a = {};
    a.fm = fm->wdev;
    a.less_len = fm->less_len;
    return a;
}

// the device call behind bg_fmd_smems_batch_dev (uint32 records) and bg_fmd_smems_batch64_dev (uint64 records)
int smems_dev(bg_fm* fm, bool out64, int all, uint64_t n_p, const uint8_t* d_pat, const uint64_t* d_pat_off, const uint32_t* d_i_pos,
              uint32_t min_len, uint32_t max_pattern_len, uint32_t cap, uint32_t* d_count, void* d_out, hipStream_t st) {
    if (!fm || (n_p && (!d_pat_off || !d_count || (cap && !d_out))) || (!all && n_p && !d_i_pos)) return BG_ERR_INVALID_ARG;
    if (!fm->fmd_ok) return BG_ERR_UNSUPPORTED;  // FMDIndex::from's assert (fmindex.rs:323-327)
    if (fm->wide && !out64) return BG_ERR_UNSUPPORTED;  // intervals of a 64-bit index do not fit uint32 records: the *64 entry points
    if (max_pattern_len >= 65535) return BG_ERR_TOO_LARGE;
    if (n_p == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    bg_scratch_guard guard(ctx, st);  // the interval lists live in the ctx's scratch
    // persistent quads, as many as are resident (the kernel's registers decide: six blocks per CU on the 32-bit layout, five
    // on the 64-bit one; round 2 .. 5 launched four): the walk is a chain of dependent block accesses per read, and reads in
    // flight are all the parallelism it has
    // plain DNA (every index over an ACGT text, with or without '$' in the BWT): the straight-line extension (Ctx::backward_ext)
    bool plain = true;
    {
        const uint16_t* hc = fm->h_class;
        for (const char* z = "TGCA"; *z; z++) plain = plain && hc[(uint8_t)*z] < 4;
        for (const char* z = "Ntgcna"; *z; z++) plain = plain && hc[(uint8_t)*z] == kClsZero;
        const uint16_t cd = hc[(uint8_t)'$'];
        plain = plain && (cd == kClsZero || (cd >= kClsSparse && cd < kClsDense));
        if (getenv("BG_K7_GENERAL")) plain = false;  // (tests, A/B)
    }
    // ... and '$' the only listed symbol, with at most two rows (every T$ / T$R$ index): the exceptions of the code stream are
    // exactly those rows
    bool dollar2 = plain && !getenv("BG_K7_PLAIN1");
    {
        uint32_t listed = 0;
        for (uint32_t b = 0; b < 256; b++) listed += fm->h_class[b] >= kClsSparse && fm->h_class[b] < kClsDense;
        const uint16_t cd = fm->h_class[(uint8_t)'$'];
        const uint32_t want = (cd >= kClsSparse && cd < kClsDense) ? 1u : 0u;
        dollar2 = dollar2 && listed == want;
    }
    auto fill = [&](auto& a) {
        a.n_p = n_p;
        a.pat = d_pat;
        a.pat_off = d_pat_off;
        a.i_pos = d_i_pos;
        a.min_len = min_len;
        a.all = all;
        a.cap = cap;
        a.out_count = d_count;
        a.out = d_out;
        a.list_cap = max_pattern_len + 2;
    };
    // persistent quads, as many as are resident (the kernel's registers decide): the walk is a chain of dependent block
    // accesses per read, and reads in flight are all the parallelism it has
    auto launch = [&](auto kernel, auto args) -> int {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        const uint64_t blocks = std::min<uint64_t>((n_p + 63) / 64, 256ull * (uint64_t)per_cu);
        int rc = bg_reserve(&ctx->bnd, &ctx->bnd_bytes, blocks * 64 * 2 * (size_t)args.list_cap * sizeof(uint4) * (fm->wide ? 2 : 1));
        if (rc) return rc;
        args.lists = (uint4*)ctx->bnd;
        kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(args);
        return BG_OK;
    };
    int rc;
    auto pick = [&](auto a, auto wide_tag, auto out_tag) -> int {
        constexpr bool W = decltype(wide_tag)::value, O64 = decltype(out_tag)::value;
        fill(a);
        if (plain && dollar2 && a.fm.n_exc <= 2) return launch(fmd_smems_kernel<W, O64, 2>, a);
        if (plain) return launch(fmd_smems_kernel<W, O64, 1>, a);
        return launch(fmd_smems_kernel<W, O64, 0>, a);
    };
    if (fm->wide)
        rc = pick(k7_args<true>(fm), std::true_type{}, std::true_type{});
    else if (out64)
        rc = pick(k7_args<false>(fm), std::false_type{}, std::true_type{});
    else
        rc = pick(k7_args<false>(fm), std::false_type{}, std::false_type{});
    if (rc) return rc;
    BG_HIP(hipGetLastError());
    return BG_OK;
}

// host buffers in and out; `elem` = bytes of a record field (4 or 8)
int smems_host(bg_fm* fm, bool out64, int all, uint64_t n_p, const uint8_t* pat, const uint64_t* pat_off, const uint32_t* i_pos,
               uint32_t min_len, uint32_t cap, uint32_t* count, void* out) {
    if (!fm || (n_p && (!pat_off || !count || (cap && !out))) || (!all && n_p && !i_pos)) return BG_ERR_INVALID_ARG;
    if (n_p == 0) return BG_OK;
    const size_t rec = out64 ? 48 : 24;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    uint64_t max_len = 0;
    for (uint64_t q = 0; q < n_p; q++) max_len = std::max(max_len, pat_off[q + 1] - pat_off[q]);
    const uint64_t pat_bytes = pat_off[n_p];
    uint8_t* d_pat = nullptr;
    uint64_t* d_off = nullptr;
    uint32_t *d_i = nullptr, *d_cnt = nullptr;
    void* d_out = nullptr;
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_pat, std::max<uint64_t>(pat_bytes, 16)));
        BG_HIP(hipMalloc((void**)&d_off, (n_p + 1) * 8));
        BG_HIP(hipMalloc((void**)&d_i, n_p * 4));
        BG_HIP(hipMalloc((void**)&d_cnt, n_p * 4));
        BG_HIP(hipMalloc(&d_out, std::max<uint64_t>(n_p * (uint64_t)cap * rec, 16)));
        hipStream_t st = ctx->stream;
        if (pat_bytes) BG_HIP(hipMemcpyAsync(d_pat, pat, pat_bytes, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_off, pat_off, (n_p + 1) * 8, hipMemcpyHostToDevice, st));
        if (i_pos) BG_HIP(hipMemcpyAsync(d_i, i_pos, n_p * 4, hipMemcpyHostToDevice, st));
        int rc = smems_dev(fm, out64, all, n_p, d_pat, d_off, i_pos ? d_i : nullptr, min_len, (uint32_t)max_len, cap, d_cnt, d_out, st);
        if (rc) return rc;
        BG_HIP(hipMemcpyAsync(count, d_cnt, n_p * 4, hipMemcpyDeviceToHost, st));
        if (cap) BG_HIP(hipMemcpyAsync(out, d_out, n_p * (uint64_t)cap * rec, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    };
    int rc = run();
    hipFree(d_pat);
    hipFree(d_off);
    hipFree(d_i);
    hipFree(d_cnt);
    hipFree(d_out);
    if (rc) return rc;
    int status = BG_OK;
    for (uint64_t q = 0; q < n_p; q++) {
        if (count[q] == 0xFFFFFFFFu)
            status = BG_ERR_OUT_OF_ALPHABET;
        else if (count[q] > cap && status == BG_OK)
            status = BG_ERR_OPS_CAP;
    }
    return status;
}

int interval_host(bg_fm* fm, bool out64, uint64_t n_req, const uint8_t* op, const void* iv_in, const uint8_t* sym, void* iv_out) {
    if (!fm || (n_req && (!op || !iv_in || !sym || !iv_out))) return BG_ERR_INVALID_ARG;
    if (!fm->fmd_ok) return BG_ERR_UNSUPPORTED;
    if (fm->wide && !out64) return BG_ERR_UNSUPPORTED;
    if (n_req == 0) return BG_OK;
    const size_t rec = out64 ? 32 : 16;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    uint8_t *d_op = nullptr, *d_sym = nullptr, *d_ok = nullptr;
    void *d_in = nullptr, *d_out = nullptr;
    std::vector<uint8_t> okv(n_req);
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_op, n_req));
        BG_HIP(hipMalloc((void**)&d_sym, n_req));
        BG_HIP(hipMalloc((void**)&d_ok, n_req));
        BG_HIP(hipMalloc(&d_in, n_req * rec));
        BG_HIP(hipMalloc(&d_out, n_req * rec));
        hipStream_t st = ctx->stream;
        BG_HIP(hipMemcpyAsync(d_op, op, n_req, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_sym, sym, n_req, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_in, iv_in, n_req * rec, hipMemcpyHostToDevice, st));
        const dim3 grid((unsigned)((n_req + 63) / 64));
        if (fm->wide)
            fmd_interval_kernel<true, true><<<grid, dim3(256), 0, st>>>(k7_args<true>(fm), n_req, d_op, d_in, d_sym, d_out, d_ok);
        else if (out64)
            fmd_interval_kernel<false, true><<<grid, dim3(256), 0, st>>>(k7_args<false>(fm), n_req, d_op, d_in, d_sym, d_out, d_ok);
        else
            fmd_interval_kernel<false, false><<<grid, dim3(256), 0, st>>>(k7_args<false>(fm), n_req, d_op, d_in, d_sym, d_out, d_ok);
        BG_HIP(hipGetLastError());
        BG_HIP(hipMemcpyAsync(iv_out, d_out, n_req * rec, hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(okv.data(), d_ok, n_req, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    };
    int rc = run();
    hipFree(d_op);
    hipFree(d_sym);
    hipFree(d_ok);
    hipFree(d_in);
    hipFree(d_out);
    if (rc) return rc;
    for (uint64_t q = 0; q < n_req; q++)
        if (!okv[q]) return BG_ERR_OUT_OF_ALPHABET;
    return BG_OK;
}

}  // namespace

extern "C" int bg_fmd_smems_batch_dev(bg_fm* fm, int all, uint64_t n_p, const uint8_t* d_pat, const uint64_t* d_pat_off,
                                      const uint32_t* d_i_pos, uint32_t min_len, uint32_t max_pattern_len, uint32_t cap,
                                      uint32_t* d_count, uint32_t* d_out, void* stream) {
    return smems_dev(fm, false, all, n_p, d_pat, d_pat_off, d_i_pos, min_len, max_pattern_len, cap, d_count, d_out, (hipStream_t)stream);
}
extern "C" int bg_fmd_smems_batch64_dev(bg_fm* fm, int all, uint64_t n_p, const uint8_t* d_pat, const uint64_t* d_pat_off,
                                        const uint32_t* d_i_pos, uint32_t min_len, uint32_t max_pattern_len, uint32_t cap,
                                        uint32_t* d_count, uint64_t* d_out, void* stream) {
    return smems_dev(fm, true, all, n_p, d_pat, d_pat_off, d_i_pos, min_len, max_pattern_len, cap, d_count, d_out, (hipStream_t)stream);
}
extern "C" int bg_fmd_smems_batch(bg_fm* fm, int all, uint64_t n_p, const uint8_t* pat, const uint64_t* pat_off,
                                  const uint32_t* i_pos, uint32_t min_len, uint32_t cap, uint32_t* count, uint32_t* out) {
    return smems_host(fm, false, all, n_p, pat, pat_off, i_pos, min_len, cap, count, out);
}
extern "C" int bg_fmd_smems_batch64(bg_fm* fm, int all, uint64_t n_p, const uint8_t* pat, const uint64_t* pat_off,
                                    const uint32_t* i_pos, uint32_t min_len, uint32_t cap, uint32_t* count, uint64_t* out) {
    return smems_host(fm, true, all, n_p, pat, pat_off, i_pos, min_len, cap, count, out);
}
extern "C" int bg_fmd_interval_batch(bg_fm* fm, uint64_t n_req, const uint8_t* op, const uint32_t* iv_in, const uint8_t* sym,
                                     uint32_t* iv_out) {
    return interval_host(fm, false, n_req, op, iv_in, sym, iv_out);
}
extern "C" int bg_fmd_interval_batch64(bg_fm* fm, uint64_t n_req, const uint8_t* op, const uint64_t* iv_in, const uint8_t* sym,
                                       uint64_t* iv_out) {
    return interval_host(fm, true, n_req, op, iv_in, sym, iv_out);
}
