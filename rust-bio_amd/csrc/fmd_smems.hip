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

#include "fm_kernels.h"

using namespace bgfm;

namespace {

struct BiIv {  // BiInterval (fmindex.rs:254-259) + the match length smems tracks next to it
    uint32_t lower, lower_rev, size;
    uint32_t msz;   // match_size
    uint32_t mlen;  // match_len of the (interval, match_len) pairs in curr / prev
};

struct FmdArgs {
    FmDev fm;
    uint32_t less_len;
    uint64_t n_p;
    const uint8_t* pat;
    const uint64_t* pat_off;
    const uint32_t* i_pos;  // smems: position every pattern's matches must overlap; nullptr with all != 0
    uint32_t min_len;
    int all;
    uint32_t cap;           // output records per pattern
    uint32_t* out_count;    // [n_p] records found (may exceed cap), 0xFFFFFFFF: the reference would panic
    uint32_t* out;          // [n_p * cap * 6]: lower, lower_rev, size, match_size, position, length
    uint4* lists;           // per quad slot: 2 * list_cap entries
    uint32_t list_cap;
};

__constant__ uint8_t kExtOrder[11] = {'$', 'T', 'G', 'C', 'N', 'A', 't', 'g', 'c', 'n', 'a'};  // fmindex.rs:536

struct Ctx {
    const FmdArgs& a;
    const uint16_t* s_class;
    const uint32_t* s_less;
    const uint8_t* s_comp;
    const uint32_t* s_exc;
    uint32_t t;
    bool panic;

    __device__ uint32_t exc_le(uint32_t r) const { return count_le(s_exc, 0u, a.fm.n_exc, r); }
    // counts of the four codes in bwt[0..=r]
    __device__ void counts(uint32_t r, uint32_t c[4]) const {
        const uint32_t b = r / kSymPerBlock, o = r - b * kSymPerBlock;
        const uint4 v = a.fm.blocks[(uint64_t)b * 4 + t];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) c[k] = quad_sum(block_part(v, t, o, k));
        if (a.fm.n_exc) c[0] -= exc_le(r);  // exceptions sit in the stream as code 0
    }
    __device__ uint32_t occ_cls(uint32_t cls, uint32_t r, const uint32_t c[4]) {
        if (cls < 4) return c[cls];
        if (cls == kClsPanic) {
            panic = true;
            return 0;
        }
        if (cls >= kClsDense) {  // a genome with many N: the fourth base and N are ranked in their bit vectors
            uint32_t o;
            const uint4 v = bv_load(a.fm, cls - kClsDense, r, t, o);
            return quad_sum(bv_part(v, t, o));
        }
        if (cls >= kClsSparse) {
            const uint32_t e = cls - kClsSparse;
            const uint32_t lo = a.fm.sparse_off[e], hi = a.fm.sparse_off[e + 1];
            return count_le(a.fm.exc_sym_pos, lo, hi, r) - lo;
        }
        return 0;  // in the alphabet, never in the BWT
    }
    __device__ uint32_t less_of(uint32_t s) {
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
        uint32_t cR[4], cL[4] = {0, 0, 0, 0};
        const uint32_t posR = iv.lower + iv.size - 1;
        counts(posR, cR);
        if (iv.lower > 0) counts(iv.lower - 1, cL);
        uint32_t s = 0, o = 0, l = iv.lower_rev;
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
        const uint32_t lo = e.lower;
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

__device__ __forceinline__ uint4 pack(const BiIv& v) { return make_uint4(v.lower, v.lower_rev, v.size, v.msz << 16 | v.mlen); }
__device__ __forceinline__ BiIv unpack(const uint4 u) { return BiIv{u.x, u.y, u.z, u.w >> 16, u.w & 0xFFFFu}; }

__global__ __launch_bounds__(256) void fmd_smems_kernel(const FmdArgs a) {
    __shared__ uint16_t s_class[256];
    __shared__ uint32_t s_less[256];
    __shared__ uint8_t s_comp[256];
    __shared__ uint32_t s_exc[kMaxExcLds];
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

    const uint32_t t = threadIdx.x & 3;
    const uint64_t slot = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const uint64_t n_slots = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint4* list0 = a.lists + slot * 2 * a.list_cap;
    uint4* list1 = list0 + a.list_cap;

    for (uint64_t q = slot; q < a.n_p; q += n_slots) {
        const uint64_t off = a.pat_off[q];
        const uint32_t plen = (uint32_t)(a.pat_off[q + 1] - off);
        const uint8_t* pattern = a.pat + off;
        Ctx cx{a, s_class, s_less, s_comp, s_exc, t, false};
        uint32_t n_out = 0;
        uint32_t* out = a.out + q * (uint64_t)a.cap * 6;
        auto emit = [&](const BiIv& iv, uint32_t pos, uint32_t len) {
            if (t == 0 && n_out < a.cap) {
                uint32_t* o = out + (uint64_t)n_out * 6;
                o[0] = iv.lower;
                o[1] = iv.lower_rev;
                o[2] = iv.size;
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
                    if (t == 0) curr[n_curr] = pack(interval);
                    n_curr++;
                }
                if (fwd.size == 0) break;
                interval = fwd;
                match_len += 1;
            }
            interval.mlen = match_len;
            if (t == 0) curr[n_curr] = pack(interval);
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
                int64_t last_size = -1;
                for (uint32_t e = 0; e < n_prev && !cx.panic; e++) {
                    const BiIv pv = unpack(prev[prev_reversed ? n_prev - 1 - e : e]);
                    BiIv fwd = cx.backward_ext(pv, sym);
                    if ((fwd.size == 0 || k == -1) && n_curr == 0 && k < j && pv.mlen >= a.min_len) {
                        j = k;
                        emit(pv, (uint32_t)(k + 1), pv.mlen);
                        reach = max(reach, (uint32_t)(k + 1) + pv.mlen);
                    }
                    if (fwd.size != 0 && (int64_t)fwd.size != last_size) {
                        last_size = (int64_t)fwd.size;
                        fwd.mlen = pv.mlen + 1;
                        if (t == 0) curr[n_curr] = pack(fwd);
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
__global__ __launch_bounds__(256) void fmd_interval_kernel(FmdArgs a, uint64_t n_req, const uint8_t* op, const uint32_t* iv_in,
                                                           const uint8_t* sym, uint32_t* iv_out, uint8_t* ok) {
    __shared__ uint16_t s_class[256];
    __shared__ uint32_t s_less[256];
    __shared__ uint8_t s_comp[256];
    __shared__ uint32_t s_exc[kMaxExcLds];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        s_class[i] = a.fm.sym_class[i];
        s_less[i] = a.fm.less[i];
        uint8_t c = (uint8_t)i;
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
    const uint32_t t = threadIdx.x & 3;
    const uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    if (q >= n_req) return;  // quad-uniform
    Ctx cx{a, s_class, s_less, s_comp, s_exc, t, false};
    BiIv in{iv_in[4 * q], iv_in[4 * q + 1], iv_in[4 * q + 2], iv_in[4 * q + 3], 0};
    BiIv r = in;
    switch (op[q]) {
        case 0: r = BiIv{0, 0, a.fm.n, 0, 0}; break;
        case 1: r = cx.init_interval_with(sym[q]); break;
        case 2: r = cx.backward_ext(in, sym[q]); break;
        default: r = cx.forward_ext(in, sym[q]); break;
    }
    if (t == 0) {
        iv_out[4 * q] = r.lower;
        iv_out[4 * q + 1] = r.lower_rev;
        iv_out[4 * q + 2] = r.size;
        iv_out[4 * q + 3] = r.msz;
        ok[q] = cx.panic ? 0 : 1;
    }
}

}  // namespace

extern "C" int bg_fmd_smems_batch_dev(bg_fm* fm, int all, uint64_t n_p, const uint8_t* d_pat, const uint64_t* d_pat_off,
                                      const uint32_t* d_i_pos, uint32_t min_len, uint32_t max_pattern_len, uint32_t cap,
                                      uint32_t* d_count, uint32_t* d_out, void* stream) {
    if (!fm || (n_p && (!d_pat_off || !d_count || (cap && !d_out))) || (!all && n_p && !d_i_pos)) return BG_ERR_INVALID_ARG;
    if (!fm->fmd_ok) return BG_ERR_UNSUPPORTED;  // FMDIndex::from's assert (fmindex.rs:323-327)
    if (max_pattern_len >= 65535) return BG_ERR_TOO_LARGE;
    if (n_p == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    hipStream_t st = (hipStream_t)stream;
    bg_scratch_guard guard(ctx, st);  // the interval lists live in the ctx's scratch
    uint64_t blocks = std::min<uint64_t>((n_p + 63) / 64, 256 * 4);
    const uint32_t list_cap = max_pattern_len + 2;
    int rc = bg_reserve(&ctx->bnd, &ctx->bnd_bytes, blocks * 64 * 2 * (size_t)list_cap * sizeof(uint4));
    if (rc) return rc;
    FmdArgs a = {};
    a.fm = fm->dev;
    a.less_len = fm->less_len;
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
    fmd_smems_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

extern "C" int bg_fmd_smems_batch(bg_fm* fm, int all, uint64_t n_p, const uint8_t* pat, const uint64_t* pat_off,
                                  const uint32_t* i_pos, uint32_t min_len, uint32_t cap, uint32_t* count, uint32_t* out) {
    if (!fm || (n_p && (!pat_off || !count || (cap && !out))) || (!all && n_p && !i_pos)) return BG_ERR_INVALID_ARG;
    if (n_p == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    uint64_t max_len = 0;
    for (uint64_t q = 0; q < n_p; q++) max_len = std::max(max_len, pat_off[q + 1] - pat_off[q]);
    const uint64_t pat_bytes = pat_off[n_p];
    uint8_t* d_pat = nullptr;
    uint64_t* d_off = nullptr;
    uint32_t *d_i = nullptr, *d_cnt = nullptr, *d_out = nullptr;
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_pat, std::max<uint64_t>(pat_bytes, 16)));
        BG_HIP(hipMalloc((void**)&d_off, (n_p + 1) * 8));
        BG_HIP(hipMalloc((void**)&d_i, n_p * 4));
        BG_HIP(hipMalloc((void**)&d_cnt, n_p * 4));
        BG_HIP(hipMalloc((void**)&d_out, std::max<uint64_t>(n_p * (uint64_t)cap * 24, 16)));
        hipStream_t st = ctx->stream;
        if (pat_bytes) BG_HIP(hipMemcpyAsync(d_pat, pat, pat_bytes, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_off, pat_off, (n_p + 1) * 8, hipMemcpyHostToDevice, st));
        if (i_pos) BG_HIP(hipMemcpyAsync(d_i, i_pos, n_p * 4, hipMemcpyHostToDevice, st));
        int rc = bg_fmd_smems_batch_dev(fm, all, n_p, d_pat, d_off, i_pos ? d_i : nullptr, min_len, (uint32_t)max_len, cap, d_cnt,
                                        d_out, st);
        if (rc) return rc;
        BG_HIP(hipMemcpyAsync(count, d_cnt, n_p * 4, hipMemcpyDeviceToHost, st));
        if (cap) BG_HIP(hipMemcpyAsync(out, d_out, n_p * (uint64_t)cap * 24, hipMemcpyDeviceToHost, st));
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

extern "C" int bg_fmd_interval_batch(bg_fm* fm, uint64_t n_req, const uint8_t* op, const uint32_t* iv_in, const uint8_t* sym,
                                     uint32_t* iv_out) {
    if (!fm || (n_req && (!op || !iv_in || !sym || !iv_out))) return BG_ERR_INVALID_ARG;
    if (!fm->fmd_ok) return BG_ERR_UNSUPPORTED;
    if (n_req == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    uint8_t *d_op = nullptr, *d_sym = nullptr, *d_ok = nullptr;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    std::vector<uint8_t> okv(n_req);
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_op, n_req));
        BG_HIP(hipMalloc((void**)&d_sym, n_req));
        BG_HIP(hipMalloc((void**)&d_ok, n_req));
        BG_HIP(hipMalloc((void**)&d_in, n_req * 16));
        BG_HIP(hipMalloc((void**)&d_out, n_req * 16));
        hipStream_t st = ctx->stream;
        BG_HIP(hipMemcpyAsync(d_op, op, n_req, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_sym, sym, n_req, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_in, iv_in, n_req * 16, hipMemcpyHostToDevice, st));
        FmdArgs a = {};
        a.fm = fm->dev;
        a.less_len = fm->less_len;
        fmd_interval_kernel<<<dim3((unsigned)((n_req + 63) / 64)), dim3(256), 0, st>>>(a, n_req, d_op, d_in, d_sym, d_out, d_ok);
        BG_HIP(hipGetLastError());
        BG_HIP(hipMemcpyAsync(iv_out, d_out, n_req * 16, hipMemcpyDeviceToHost, st));
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
