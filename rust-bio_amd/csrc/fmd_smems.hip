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

__constant__ uint8_t kExtOrder[11] = {'$', 'T', 'G', 'C', 'N', 'A', 't', 'g', 'c', 'n', 'a'};  // fmindex.rs:536

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

template <bool WIDE>
struct Ctx {
    using P = typename FmLayout<WIDE>::Pos;
    using BiIv = BiIvT<P>;
    const FmdArgsT<WIDE>& a;
    const uint16_t* s_class;
    const P* s_less;
    const uint8_t* s_comp;
    const P* s_exc;
    uint32_t t;
    bool panic;

    __device__ uint32_t exc_le(P r) const { return k7_count_le(s_exc, 0u, a.fm.n_exc, r); }
    // counts of the four codes in bwt[0..=r]
    __device__ void counts(P r, P c[4]) const {
        const uint64_t b = (uint64_t)(r / kSymPerBlock);
        const uint32_t o = (uint32_t)(r - (P)b * kSymPerBlock);
        const uint4 v = a.fm.blocks[b * 4 + t];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) c[k] = (P)k7_base(a.fm, b, k) + quad_sum(block_part(v, t, o, k));
        if (a.fm.n_exc) c[0] -= exc_le(r);  // exceptions sit in the stream as code 0
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
            const uint32_t lo = a.fm.sparse_off[e], hi = a.fm.sparse_off[e + 1];
            return k7_count_le(a.fm.exc_sym_pos, lo, hi, r) - lo;
        }
        return 0;  // in the alphabet, never in the BWT
    }
    __device__ P less_of(uint32_t s) {
        if (s >= a.less_len) {  // index out of bounds in the reference
            panic = true;
            return 0;
        }
        return s_less[s];
    }
    // fmindex.rs:527-558
    __device__ BiIv backward_ext(const BiIv& iv, uint32_t sym) {
        BiIv r = iv;
        if (iv.lower + iv.size == 0) {  // usize underflow of lower + size - 1
            panic = true;
            return r;
        }
        P cR[4], cL[4] = {0, 0, 0, 0};
        const P posR = iv.lower + iv.size - 1;
        counts(posR, cR);
        if (iv.lower > 0) counts(iv.lower - 1, cL);
        P s = 0, o = 0, l = iv.lower_rev;
        for (int idx = 0; idx < 11; idx++) {
            const uint32_t b = kExtOrder[idx];
            const uint32_t cls = s_class[b];
            l += s;
            o = iv.lower == 0 ? 0 : occ_cls(cls, iv.lower - 1, cL);
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
                                          typename FmLayout<WIDE>::Pos* s_exc) {
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

template <bool WIDE, bool OUT64>
__global__ __launch_bounds__(256) void fmd_smems_kernel(const FmdArgsT<WIDE> a) {
    using P = typename FmLayout<WIDE>::Pos;
    using BiIv = BiIvT<P>;
    using O = std::conditional_t<OUT64, uint64_t, uint32_t>;
    static_assert(OUT64 || !WIDE, "64-bit positions need 64-bit records");
    constexpr uint32_t LW = WIDE ? 2 : 1;  // uint4 per list entry
    __shared__ uint16_t s_class[256];
    __shared__ P s_less[256];
    __shared__ uint8_t s_comp[256];
    __shared__ P s_exc[kMaxExcLds];
    k7_tables<WIDE>(a, s_class, s_less, s_comp, s_exc);

    const uint32_t t = threadIdx.x & 3;
    const uint64_t slot = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const uint64_t n_slots = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint4* list0 = a.lists + slot * 2 * a.list_cap * LW;
    uint4* list1 = list0 + (uint64_t)a.list_cap * LW;

    for (uint64_t q = slot; q < a.n_p; q += n_slots) {
        const uint64_t off = a.pat_off[q];
        const uint32_t plen = (uint32_t)(a.pat_off[q + 1] - off);
        const uint8_t* pattern = a.pat + off;
        Ctx<WIDE> cx{a, s_class, s_less, s_comp, s_exc, t, false};
        uint32_t n_out = 0;
        O* out = (O*)a.out + q * (uint64_t)a.cap * 6;
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
        // one smems(pattern, i, l) call (fmindex.rs:363-434); returns the largest pos + len it found
        auto smems = [&](uint32_t i, uint32_t& reach) {
            uint4* curr = list0;
            uint4* prev = list1;
            uint32_t n_curr = 0, n_prev = 0;
            uint32_t match_len = 0;
            BiIv interval = cx.init_interval_with(pattern[i]);
            if (interval.size != 0) match_len += 1;
            for (uint32_t p = i + 1; p < plen && !cx.panic; p++) {
                const BiIv fwd = cx.forward_ext(interval, pattern[p]);
                if (interval.size != fwd.size) {
                    interval.mlen = match_len;
                    if (t == 0) list_put(curr, n_curr, interval);
                    n_curr++;
                }
                if (fwd.size == 0) break;
                interval = fwd;
                match_len += 1;
            }
            interval.mlen = match_len;
            if (t == 0) list_put(curr, n_curr, interval);
            n_curr++;
            // "reverse intervals such that longest comes first": prev is read back to front instead
            uint4* tmp = curr;
            curr = prev;
            prev = tmp;
            n_prev = n_curr;
            bool prev_reversed = true;
            int32_t j = (int32_t)plen;
            for (int32_t k = (int32_t)i - 1; k >= -1 && !cx.panic; k--) {
                const uint32_t sym = k == -1 ? (uint32_t)'$' : (uint32_t)pattern[k];
                n_curr = 0;
                bool have_last = false;
                P last_size = 0;
                for (uint32_t e = 0; e < n_prev && !cx.panic; e++) {
                    BiIv pv;
                    list_get(prev, prev_reversed ? n_prev - 1 - e : e, pv);
                    BiIv fwd = cx.backward_ext(pv, sym);
                    if ((fwd.size == 0 || k == -1) && n_curr == 0 && k < j && pv.mlen >= a.min_len) {
                        j = k;
                        emit(pv, (uint32_t)(k + 1), pv.mlen);
                        reach = max(reach, (uint32_t)(k + 1) + pv.mlen);
                    }
                    if (fwd.size != 0 && !(have_last && fwd.size == last_size)) {
                        have_last = true;
                        last_size = fwd.size;
                        fwd.mlen = pv.mlen + 1;
                        if (t == 0) list_put(curr, n_curr, fwd);
                        n_curr++;
                    }
                }
                if (n_curr == 0) break;
                tmp = curr;
                curr = prev;
                prev = tmp;
                n_prev = n_curr;
                prev_reversed = false;
            }
        };
        if (plen == 0) {
            if (!a.all) cx.panic = true;  // pattern[i] on an empty pattern
        } else if (a.all) {  // fmindex.rs:479-501
            uint32_t i0 = 0;
            while (i0 < plen && !cx.panic) {
                uint32_t reach = i0 + 1;
                smems(i0, reach);
                i0 = reach;
            }
        } else {
            const uint32_t i = a.i_pos[q];
            uint32_t reach = 0;
            if (i >= plen)
                cx.panic = true;
            else
                smems(i, reach);
        }
        if (t == 0) a.out_count[q] = cx.panic ? 0xFFFFFFFFu : n_out;
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
    k7_tables<WIDE>(a, s_class, s_less, s_comp, s_exc);
    const O* iv_in = (const O*)iv_in_;
    O* iv_out = (O*)iv_out_;
    const uint32_t t = threadIdx.x & 3;
    const uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    if (q >= n_req) return;  // quad-uniform
    Ctx<WIDE> cx{a, s_class, s_less, s_comp, s_exc, t, false};
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
    int per_cu = 0;
    const void* kfn = fm->wide ? (const void*)fmd_smems_kernel<true, true> : out64 ? (const void*)fmd_smems_kernel<false, true> : (const void*)fmd_smems_kernel<false, false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    uint64_t blocks = std::min<uint64_t>((n_p + 63) / 64, 256ull * (uint64_t)per_cu);
    const uint32_t list_cap = max_pattern_len + 2;
    int rc = bg_reserve(&ctx->bnd, &ctx->bnd_bytes, blocks * 64 * 2 * (size_t)list_cap * sizeof(uint4) * (fm->wide ? 2 : 1));
    if (rc) return rc;
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
        a.lists = (uint4*)ctx->bnd;
        a.list_cap = list_cap;
    };
    if (fm->wide) {
        auto a = k7_args<true>(fm);
        fill(a);
        fmd_smems_kernel<true, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
    } else {
        auto a = k7_args<false>(fm);
        fill(a);
        if (out64)
            fmd_smems_kernel<false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
        else
            fmd_smems_kernel<false, false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
    }
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
