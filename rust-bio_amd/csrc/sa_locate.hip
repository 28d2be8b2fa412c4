// K6 — suffix-array lookups for FM-index hits on gfx950: batched `Interval::occ`
// (/root/reference/src/data_structures/fmindex.rs:75-79) over a raw suffix array
// (`RawSuffixArray::get`, suffix_array.rs:134-141) or a `SampledSuffixArray`
// (`get`, suffix_array.rs:157-184; built by `sample`, 86-120).
//
// Raw: a gather.  Sampled: every row LF-walks until it reaches a sampled row (row % s == 0) or a
// row whose BWT byte is the sentinel (those rows are stored on the side, sorted — a hash map in the
// reference).  The walk reuses K5's machinery: a quad of 4 lanes per row, one 64-byte block per
// rank; the BWT byte of the row is read from the same 2-bit stream (+ the exception list), both
// loads of a step are issued together, and quads fetch the next row as soon as theirs resolves.
#include <algorithm>

#include "fm_kernels.h"

using namespace bgfm;

namespace {

struct SaDev {
    const uint32_t* sa;         // raw SA or the samples
    const uint32_t* extra_row;  // sorted
    const uint32_t* extra_pos;
    const uint8_t* exc_byte;    // byte of exception e (parallel to FmDev::exc_pos)
    uint32_t n_extra;
    uint32_t rate;
    uint32_t sentinel;
    uint32_t code_byte;         // byte of code c in bits [8c, 8c+8)
};

__global__ __launch_bounds__(256) void sa_raw_get_kernel(const uint32_t* __restrict__ sa, uint32_t n_text, uint64_t n,
                                                         const uint64_t* index, uint64_t* pos_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = index[i];
    pos_out[i] = r < n_text ? (uint64_t)sa[r] : BG_SA_NONE;
}

// rows of interval v are out_off[v] .. out_off[v+1]: write lower[v] + k into slot out_off[v] + k
__global__ __launch_bounds__(256) void interval_rows_kernel(uint64_t n_iv, const uint64_t* __restrict__ lower,
                                                            const uint64_t* __restrict__ out_off, uint64_t total,
                                                            uint64_t* rows) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    uint64_t lo = 0, hi = n_iv;  // last v with out_off[v] <= k
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (out_off[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    rows[k] = lower[lo] + (k - out_off[lo]);
}

__global__ __launch_bounds__(256) void sa_sampled_get_kernel(FmDev fm, SaDev sa, uint64_t n, const uint64_t* index,
                                                             uint64_t* pos_out) {
    __shared__ uint16_t s_class[256];
    __shared__ uint32_t s_less[256];
    __shared__ uint32_t s_exc[kMaxExcLds];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        s_class[i] = fm.sym_class[i];
        s_less[i] = fm.less[i];
    }
    for (uint32_t i = threadIdx.x; i < fm.n_exc; i += blockDim.x) s_exc[i] = fm.exc_pos[i];  // n_exc <= kMaxExcLds
    __syncthreads();
    const bool raw_bwt = fm.bwt_raw != nullptr;  // dense symbols exist: the stream's code 0 does not say which byte

    const uint32_t t = threadIdx.x & 3;
    const uint64_t n_quads = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const uint32_t* blocks32 = (const uint32_t*)fm.blocks;

    bool active = false;
    uint32_t pos = 0, offset = 0;
    auto emit = [&](uint64_t v) {
        if (t == 0) pos_out[q] = v;
    };
    auto fetch = [&]() {
        active = false;
        while (q < n) {
            const uint64_t r = index[q];
            if (r < fm.n) {
                pos = (uint32_t)r;
                offset = 0;
                active = true;
                return;
            }
            emit(BG_SA_NONE);  // SuffixArray::get -> None
            q += n_quads;
        }
    };
    fetch();

    while (__any(active)) {
        if (active) {
            if (pos % sa.rate == 0) {  // suffix_array.rs:162-164
                emit((uint64_t)sa.sa[pos / sa.rate] + offset);
                q += n_quads;
                fetch();
                continue;
            }
            // both loads of the step: the block of rank(pos - 1, .) and the word holding bwt[pos]
            const uint32_t pb = pos / kSymPerBlock, po = pos - pb * kSymPerBlock;
            const uint32_t rb = (pos - 1) / kSymPerBlock, ro = (pos - 1) - rb * kSymPerBlock;
            const uint4 vr = fm.blocks[(uint64_t)rb * 4 + t];
            uint32_t c;
            if (raw_bwt) {
                c = fm.bwt_raw[pos];
            } else {
                const uint32_t word = blocks32[(uint64_t)pb * 16 + 4 + (po >> 4)];
                const uint32_t code = (word >> (2 * (po & 15))) & 3u;
                c = (sa.code_byte >> (8 * code)) & 255u;
                if (code == 0 && fm.n_exc) {  // sparse exceptions sit in the stream as code 0
                    const uint32_t e = count_le(s_exc, 0u, fm.n_exc, pos);
                    if (e > 0 && s_exc[e - 1] == pos) c = sa.exc_byte[e - 1];
                }
            }
            if (c == sa.sentinel) {  // suffix_array.rs:168-175
                uint32_t lo = 0, hi = sa.n_extra;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sa.extra_row[mid] < pos)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                emit(lo < sa.n_extra && sa.extra_row[lo] == pos ? (uint64_t)sa.extra_pos[lo] + offset : BG_SA_PANIC);
                q += n_quads;
                fetch();
                continue;
            }
            // pos = less[c] + occ.get(bwt, pos - 1, c)  (suffix_array.rs:177-178)
            const uint32_t cls = s_class[c];
            uint32_t occ = 0;
            if (cls < 4) {
                occ = quad_sum(block_part(vr, t, ro, cls));
                if (cls == 0 && fm.n_exc) occ -= count_le(s_exc, 0u, fm.n_exc, pos - 1);
            } else if (cls == kClsPanic) {
            } else if (cls >= kClsDense) {
                uint32_t o;
                const uint4 v = bv_load(fm, cls - kClsDense, pos - 1, t, o);
                occ = quad_sum(bv_part(v, t, o));
            } else if (cls >= kClsSparse) {
                const uint32_t e = cls - kClsSparse;
                const uint32_t lo = fm.sparse_off[e], hi = fm.sparse_off[e + 1];
                occ = count_le(fm.exc_sym_pos, lo, hi, pos - 1) - lo;
            }
            pos = s_less[c] + occ;
            offset += 1;
        }
    }
}

int launch_get(bg_fm* fm, uint64_t n, const uint64_t* d_index, uint64_t* d_pos, hipStream_t st) {
    if (n == 0) return BG_OK;
    if (fm->wide) return fm_wide_sa_get(fm, n, d_index, d_pos, st);  // 64-bit positions: fm_wide.hip
    if (fm->sa_kind == 1) {
        sa_raw_get_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>((const uint32_t*)fm->d_sa, fm->dev.n, n,
                                                                                d_index, d_pos);
    } else {
        SaDev sa = {};
        sa.sa = (const uint32_t*)fm->d_sa;
        sa.extra_row = (const uint32_t*)fm->d_extra_row;
        sa.extra_pos = (const uint32_t*)fm->d_extra_pos;
        sa.exc_byte = (const uint8_t*)fm->d_exc_byte;
        sa.n_extra = (uint32_t)fm->n_extra;
        sa.rate = fm->sa_rate;
        sa.sentinel = fm->sa_sentinel;
        sa.code_byte = (uint32_t)fm->code_byte[0] | (uint32_t)fm->code_byte[1] << 8 | (uint32_t)fm->code_byte[2] << 16 |
                       (uint32_t)fm->code_byte[3] << 24;
        uint64_t blocks = std::min<uint64_t>((n + 63) / 64, 256 * 8);
        sa_sampled_get_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->dev, sa, n, d_index, d_pos);
    }
    BG_HIP(hipGetLastError());
    return BG_OK;
}

// 64-bit index: the entries stay uint64 (checked against the text's length like the narrow ones)
int upload64(void** dptr, const uint64_t* src, uint64_t count, uint64_t limit, uint64_t* bytes) {
    for (uint64_t i = 0; i < count; i++)
        if (src[i] >= limit) return BG_ERR_INVALID_ARG;
    hipFree(*dptr);
    *dptr = nullptr;
    BG_HIP(hipMalloc(dptr, std::max<uint64_t>(count * 8, 16)));
    if (count) BG_HIP(hipMemcpy(*dptr, src, count * 8, hipMemcpyHostToDevice));
    *bytes += std::max<uint64_t>(count * 8, 16);
    return BG_OK;
}
inline uint64_t fm_len(const bg_fm* fm) { return fm->wide ? fm->wdev.n : (uint64_t)fm->dev.n; }

int upload32(void** dptr, const uint64_t* src, uint64_t count, uint64_t limit, uint64_t* bytes) {
    std::vector<uint32_t> tmp(count);
    for (uint64_t i = 0; i < count; i++) {
        if (src[i] >= limit) return BG_ERR_INVALID_ARG;
        tmp[i] = (uint32_t)src[i];
    }
    hipFree(*dptr);
    *dptr = nullptr;
    BG_HIP(hipMalloc(dptr, std::max<uint64_t>(count * 4, 16)));
    if (count) BG_HIP(hipMemcpy(*dptr, tmp.data(), count * 4, hipMemcpyHostToDevice));
    *bytes += std::max<uint64_t>(count * 4, 16);
    return BG_OK;
}

}  // namespace

extern "C" int bg_fm_set_suffix_array(bg_fm* fm, const uint64_t* sa, uint64_t n) {
    if (!fm || !sa || n != fm_len(fm)) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(fm->ctx->device));
    int rc = (fm->wide ? upload64 : upload32)(&fm->d_sa, sa, n, n, &fm->bytes);
    if (rc) return rc;
    fm->sa_kind = 1;
    return BG_OK;
}

extern "C" int bg_fm_set_sampled_suffix_array(bg_fm* fm, const uint64_t* sample, uint64_t n_sample, uint32_t sampling_rate,
                                              uint8_t sentinel, const uint64_t* extra_rows, const uint64_t* extra_pos,
                                              uint64_t n_extra) {
    if (!fm || !sample || sampling_rate == 0 || (n_extra && (!extra_rows || !extra_pos))) return BG_ERR_INVALID_ARG;
    const uint64_t n = fm_len(fm);
    if (n_sample != (n + sampling_rate - 1) / sampling_rate) return BG_ERR_INVALID_ARG;
    for (uint64_t i = 1; i < n_extra; i++)
        if (extra_rows[i] <= extra_rows[i - 1]) return BG_ERR_INVALID_ARG;  // sorted, unique
    BG_HIP(hipSetDevice(fm->ctx->device));
    int rc;
    auto up = fm->wide ? upload64 : upload32;
    if ((rc = up(&fm->d_sa, sample, n_sample, n, &fm->bytes))) return rc;
    if ((rc = up(&fm->d_extra_row, extra_rows, n_extra, n, &fm->bytes))) return rc;
    if ((rc = up(&fm->d_extra_pos, extra_pos, n_extra, n, &fm->bytes))) return rc;
    fm->n_sample = n_sample;
    fm->n_extra = n_extra;
    fm->sa_rate = sampling_rate;
    fm->sa_sentinel = sentinel;
    fm->sa_kind = 2;
    return BG_OK;
}

extern "C" int bg_sa_get_batch_dev(bg_fm* fm, uint64_t n_idx, const uint64_t* d_index, uint64_t* d_pos, void* stream) {
    if (!fm || (n_idx && (!d_index || !d_pos))) return BG_ERR_INVALID_ARG;
    if (fm->sa_kind == 0) return BG_ERR_INVALID_ARG;
    return launch_get(fm, n_idx, d_index, d_pos, (hipStream_t)stream);
}

extern "C" int bg_interval_occ_batch_dev(bg_fm* fm, uint64_t n_iv, const uint64_t* d_lower, const uint64_t* d_out_off,
                                         uint64_t total, uint64_t* d_pos, void* stream) {
    if (!fm || (n_iv && (!d_lower || !d_out_off)) || (total && !d_pos)) return BG_ERR_INVALID_ARG;
    if (fm->sa_kind == 0) return BG_ERR_INVALID_ARG;
    if (total == 0 || n_iv == 0) return BG_OK;
    hipStream_t st = (hipStream_t)stream;
    interval_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(n_iv, d_lower, d_out_off, total, d_pos);
    BG_HIP(hipGetLastError());
    return launch_get(fm, total, d_pos, d_pos, st);  // rows -> positions in place
}

extern "C" int bg_sa_get_batch(bg_fm* fm, uint64_t n_idx, const uint64_t* index, uint64_t* pos) {
    if (!fm || (n_idx && (!index || !pos))) return BG_ERR_INVALID_ARG;
    if (fm->sa_kind == 0) return BG_ERR_INVALID_ARG;
    if (n_idx == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    uint64_t* d = nullptr;
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d, n_idx * 8));
        BG_HIP(hipMemcpyAsync(d, index, n_idx * 8, hipMemcpyHostToDevice, ctx->stream));
        int rc = launch_get(fm, n_idx, d, d, ctx->stream);
        if (rc) return rc;
        BG_HIP(hipMemcpyAsync(pos, d, n_idx * 8, hipMemcpyDeviceToHost, ctx->stream));
        BG_HIP(hipStreamSynchronize(ctx->stream));
        return BG_OK;
    };
    int rc = run();
    hipFree(d);
    if (rc) return rc;
    for (uint64_t i = 0; i < n_idx; i++)
        if (pos[i] == BG_SA_PANIC) return BG_ERR_OUT_OF_ALPHABET;
    return BG_OK;
}

extern "C" int bg_interval_occ_batch(bg_fm* fm, uint64_t n_iv, const uint64_t* lower, const uint64_t* upper,
                                     uint64_t* out_off, uint64_t* pos, uint64_t pos_cap) {
    if (!fm || (n_iv && (!lower || !upper || !out_off))) return BG_ERR_INVALID_ARG;
    if (fm->sa_kind == 0) return BG_ERR_INVALID_ARG;
    uint64_t total = 0;
    for (uint64_t v = 0; v < n_iv; v++) {
        // "Interval out of range of suffix array" (fmindex.rs:77) — an empty range yields no rows
        if (upper[v] > lower[v] && upper[v] > fm_len(fm)) return BG_ERR_INVALID_ARG;
        out_off[v] = total;
        total += upper[v] > lower[v] ? upper[v] - lower[v] : 0;
    }
    if (n_iv) out_off[n_iv] = total;
    if (total > pos_cap) return BG_ERR_OPS_CAP;
    if (total == 0) return BG_OK;
    if (!pos) return BG_ERR_INVALID_ARG;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    uint64_t *d_lo = nullptr, *d_off = nullptr, *d_pos = nullptr;
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_lo, n_iv * 8));
        BG_HIP(hipMalloc((void**)&d_off, (n_iv + 1) * 8));
        BG_HIP(hipMalloc((void**)&d_pos, total * 8));
        BG_HIP(hipMemcpyAsync(d_lo, lower, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
        BG_HIP(hipMemcpyAsync(d_off, out_off, (n_iv + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        int rc = bg_interval_occ_batch_dev(fm, n_iv, d_lo, d_off, total, d_pos, ctx->stream);
        if (rc) return rc;
        BG_HIP(hipMemcpyAsync(pos, d_pos, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        BG_HIP(hipStreamSynchronize(ctx->stream));
        return BG_OK;
    };
    int rc = run();
    hipFree(d_lo);
    hipFree(d_off);
    hipFree(d_pos);
    if (rc) return rc;
    for (uint64_t i = 0; i < total; i++)
        if (pos[i] == BG_SA_PANIC) return BG_ERR_OUT_OF_ALPHABET;
    return BG_OK;
}
