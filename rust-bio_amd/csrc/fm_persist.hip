// Index persistence: bg_fm_save / bg_fm_load (round 5).
//
// The reference derives Serialize / Deserialize for what an index is made of — Occ (/root/reference/src/data_structures/
// bwt.rs:76), FMIndex (fmindex.rs:214), SampledSuffixArray (suffix_array.rs:124) — so that the suffix array, by far the
// expensive part of construction, is computed once per genome.  Here the device layout is the engine's own and cheap to
// lay out again (0.06 s per Gbp from a BWT in HBM, fm_index.hip), so what goes to disk is what the reference's FMIndex
// holds, not the rank blocks: the BWT, less, the alphabet and k, and the suffix array attached to the handle (raw or
// sampled + the rows kept for the sentinel) — and the text, if the handle owns a copy (seed-and-extend).  bg_fm_load
// reads the file, rebuilds the index with the builder the sizes call for (the 64-bit layout from 2^32 - 1 symbols on) and
// attaches the suffix array: the loaded handle answers every call like the one that was saved
// (tests/test_gpu_fm.py::test_index_survives_save_and_load).
//
// The BWT is not kept by a handle: it is read back out of the index — the 2-bit codes of the rank blocks turned into their
// bytes, the listed exceptions (sentinels, stray N) put back over them; an index with dense symbols keeps the raw BWT
// anyway (K6 reads it).
//
// File (little-endian; every section padded to 8 bytes): "BGFMIDX1", a header of 64-bit fields, alphabet, less, BWT,
// suffix-array arrays as uint64, text; the last 8 bytes are a checksum of everything before them (64-bit words, an
// xorshift-multiply mix): a truncated or altered file is refused with BG_ERR_IO.
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "fm_kernels.h"

using namespace bgfm;

namespace {

// symbol i of the 2-bit stream -> the byte its code stands for (the 64-byte block layout of fm_index.hip / fm_wide.hip)
__global__ __launch_bounds__(256) void fmp_codes_to_bytes_kernel(const uint32_t* __restrict__ blocks32, uint64_t n, uint32_t code_bytes,
                                                                 uint8_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t blk = i / kSymPerBlock;
        const uint32_t o = (uint32_t)(i - blk * kSymPerBlock);
        const uint32_t word = blocks32[blk * 16 + 4 + (o >> 4)];
        out[i] = (uint8_t)(code_bytes >> (8 * ((word >> (2 * (o & 15u))) & 3u)));
    }
}
template <typename P>
__global__ void fmp_exceptions_kernel(const P* __restrict__ pos, const uint8_t* __restrict__ byte, uint32_t n_exc, uint8_t* __restrict__ out) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_exc) out[pos[e]] = byte[e];
}

struct Mix {  // 64-bit words, xorshift-multiply; the tail of a section is zero-padded by the writer, so words are whole
    uint64_t h = 0x9E3779B97F4A7C15ull;
    void words(const void* p, size_t bytes) {
        const uint64_t* w = (const uint64_t*)p;
        uint64_t a = h;
        for (size_t i = 0; i < bytes / 8; i++) {
            a ^= w[i];
            a *= 0xD6E8FEB86659FD93ull;
            a ^= a >> 32;
        }
        h = a;
    }
};

struct Writer {
    FILE* f = nullptr;
    Mix mix;
    bool ok = true;
    std::vector<uint8_t> carry;  // (sections are written whole: no carry needed, kept for the final pad only)
    void put(const void* p, size_t bytes) {  // bytes is padded to 8 with zeros
        if (!ok) return;
        const size_t whole = bytes & ~(size_t)7;
        if (whole) {
            mix.words(p, whole);
            ok = fwrite(p, 1, whole, f) == whole;
        }
        if (ok && bytes != whole) {
            uint64_t last = 0;
            memcpy(&last, (const uint8_t*)p + whole, bytes - whole);
            mix.words(&last, 8);
            ok = fwrite(&last, 1, 8, f) == 8;
        }
    }
};
struct Reader {
    FILE* f = nullptr;
    Mix mix;
    bool ok = true;
    void get(void* p, size_t bytes) {  // the section as written by Writer::put
        if (!ok) return;
        const size_t whole = bytes & ~(size_t)7;
        if (whole) {
            ok = fread(p, 1, whole, f) == whole;
            if (ok) mix.words(p, whole);
        }
        if (ok && bytes != whole) {
            uint64_t last = 0;
            ok = fread(&last, 1, 8, f) == 8;
            if (ok) {
                mix.words(&last, 8);
                memcpy((uint8_t*)p + whole, &last, bytes - whole);
            }
        }
    }
};

constexpr char kMagic[8] = {'B', 'G', 'F', 'M', 'I', 'D', 'X', '1'};
struct Header {  // 64-bit fields only
    uint64_t n, occ_k, n_sym, less_len, sa_kind, sa_rate, sa_sentinel, n_sample, n_extra, text_bytes, wide, reserved;
};
constexpr size_t kChunk = 64ull << 20;  // device <-> host in pieces of 64 MB through one pinned buffer

// device array of n elements of `elem` bytes (4 or 8) -> the file, as uint64
int put_positions(Writer& w, const void* d_arr, uint64_t n, bool is64, void* pinned) {
    std::vector<uint64_t> wide;
    for (uint64_t lo = 0; lo < n;) {
        const uint64_t cnt = std::min<uint64_t>(n - lo, kChunk / 8);
        if (is64) {
            BG_HIP(hipMemcpy(pinned, (const uint64_t*)d_arr + lo, cnt * 8, hipMemcpyDeviceToHost));
            w.put(pinned, cnt * 8);
        } else {
            BG_HIP(hipMemcpy(pinned, (const uint32_t*)d_arr + lo, cnt * 4, hipMemcpyDeviceToHost));
            wide.resize(cnt);
            const uint32_t* s = (const uint32_t*)pinned;
            for (uint64_t i = 0; i < cnt; i++) wide[i] = s[i];
            w.put(wide.data(), cnt * 8);
        }
        lo += cnt;
    }
    return BG_OK;
}

}  // namespace

// The BWT of a handle, read back out of its rank blocks into n bytes of device memory: the 2-bit codes turned into their
// bytes, the listed exceptions (sentinels, stray N) put back over them; an index with dense symbols keeps the raw bytes.
int fm_decode_bwt_dev(const bg_fm* fm, uint8_t* d_out, hipStream_t st) {
    const uint64_t n = fm->wide ? fm->wdev.n : (uint64_t)fm->dev.n;
    if (!fm->wide && fm->d_bwt_raw) {
        BG_HIP(hipMemcpyAsync(d_out, fm->d_bwt_raw, n, hipMemcpyDeviceToDevice, st));
        return BG_OK;
    }
    const uint32_t code_bytes = (uint32_t)fm->code_byte[0] | (uint32_t)fm->code_byte[1] << 8 | (uint32_t)fm->code_byte[2] << 16 |
                                (uint32_t)fm->code_byte[3] << 24;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 20);
    fmp_codes_to_bytes_kernel<<<dim3(grid), dim3(256), 0, st>>>((const uint32_t*)fm->d_blocks, n, code_bytes, d_out);
    const uint32_t n_exc = fm->wide ? fm->wdev.n_exc : fm->dev.n_exc;
    if (n_exc) {
        if (fm->wide)
            fmp_exceptions_kernel<uint64_t><<<dim3((n_exc + 255) / 256), dim3(256), 0, st>>>((const uint64_t*)fm->d_exc_pos, (const uint8_t*)fm->d_exc_byte, n_exc, d_out);
        else
            fmp_exceptions_kernel<uint32_t><<<dim3((n_exc + 255) / 256), dim3(256), 0, st>>>((const uint32_t*)fm->d_exc_pos, (const uint8_t*)fm->d_exc_byte, n_exc, d_out);
    }
    BG_HIP(hipGetLastError());
    return BG_OK;
}

// what a handle has to remember for bg_fm_save (set by bg_fm_build / bg_fm_build_dev, fm_index.hip)
void fm_remember_inputs(bg_fm* fm, const uint8_t* alphabet, uint32_t n_sym, uint32_t occ_k, const uint64_t* less, uint32_t less_len) {
    fm->alphabet.assign(alphabet, alphabet + n_sym);
    fm->occ_k = occ_k;
    fm->h_less.assign(less, less + less_len);
}

extern "C" int bg_fm_save(const bg_fm* fm, const char* path) {
    if (!fm || !path || fm->alphabet.empty() || fm->h_less.empty()) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(fm->ctx->device));
    BG_HIP(hipDeviceSynchronize());  // (the handle is immutable once built: this only waits for a builder's last copies)
    const uint64_t n = fm->wide ? fm->wdev.n : (uint64_t)fm->dev.n;
    Header h = {};
    h.n = n;
    h.occ_k = fm->occ_k;
    h.n_sym = fm->alphabet.size();
    h.less_len = fm->h_less.size();
    h.sa_kind = (uint64_t)fm->sa_kind;
    h.sa_rate = fm->sa_rate;
    h.sa_sentinel = fm->sa_sentinel;
    h.n_sample = fm->sa_kind == 2 ? fm->n_sample : 0;
    h.n_extra = fm->sa_kind == 2 ? fm->n_extra : 0;
    h.text_bytes = (fm->d_text && fm->text_owned) ? fm->n_text + 1 : 0;
    h.wide = fm->wide ? 1 : 0;

    // the BWT back out of the index
    uint8_t* d_bwt = nullptr;
    const uint8_t* bwt_src = fm->wide ? nullptr : (const uint8_t*)fm->d_bwt_raw;
    if (!bwt_src) {
        BG_HIP(hipMalloc((void**)&d_bwt, std::max<uint64_t>(n, 16)));
        const int rcd = fm_decode_bwt_dev(fm, d_bwt, nullptr);
        if (rcd || hipDeviceSynchronize() != hipSuccess) {
            hipFree(d_bwt);
            return rcd ? rcd : BG_ERR_HIP;
        }
        bwt_src = d_bwt;
    }
    void* pinned = nullptr;
    if (hipHostMalloc(&pinned, kChunk) != hipSuccess) {
        hipFree(d_bwt);
        return BG_ERR_OOM;
    }
    Writer w;
    w.f = fopen(path, "wb");
    int rc = BG_OK;
    if (!w.f) rc = BG_ERR_IO;
    if (!rc) {
        w.put(kMagic, 8);
        w.put(&h, sizeof(h));
        w.put(fm->alphabet.data(), fm->alphabet.size());
        w.put(fm->h_less.data(), fm->h_less.size() * 8);
        for (uint64_t lo = 0; lo < n && !rc; lo += kChunk) {
            const uint64_t cnt = std::min<uint64_t>(n - lo, kChunk);
            if (hipMemcpy(pinned, bwt_src + lo, cnt, hipMemcpyDeviceToHost) != hipSuccess) rc = BG_ERR_HIP;
            else w.put(pinned, cnt);  // (a chunk is a multiple of 8 bytes except the last one)
        }
        if (!rc && fm->sa_kind == 1) rc = put_positions(w, fm->d_sa, n, fm->wide, pinned);
        if (!rc && fm->sa_kind == 2) {
            rc = put_positions(w, fm->d_sa, fm->n_sample, fm->wide, pinned);
            if (!rc) rc = put_positions(w, fm->d_extra_row, fm->n_extra, fm->wide, pinned);
            if (!rc) rc = put_positions(w, fm->d_extra_pos, fm->n_extra, fm->wide, pinned);
        }
        for (uint64_t lo = 0; lo < h.text_bytes && !rc; lo += kChunk) {
            const uint64_t cnt = std::min<uint64_t>(h.text_bytes - lo, kChunk);
            if (hipMemcpy(pinned, (const uint8_t*)fm->d_text + lo, cnt, hipMemcpyDeviceToHost) != hipSuccess) rc = BG_ERR_HIP;
            else w.put(pinned, cnt);
        }
        const uint64_t sum = w.mix.h;
        if (!rc && (!w.ok || fwrite(&sum, 1, 8, w.f) != 8)) rc = BG_ERR_IO;
        if (fclose(w.f) != 0 && !rc) rc = BG_ERR_IO;
        if (rc) remove(path);
    }
    hipHostFree(pinned);
    hipFree(d_bwt);
    return rc;
}

namespace {
struct LoadState {  // what an exception on the way must release
    FILE* f = nullptr;
    bg_fm* fm = nullptr;
};
int fm_load_impl(bg_ctx* ctx, const char* path, bg_fm** out, LoadState& ls) {
    Reader r;
    r.f = fopen(path, "rb");
    if (!r.f) return BG_ERR_IO;
    ls.f = r.f;
    char magic[8];
    Header h = {};
    r.get(magic, 8);
    r.get(&h, sizeof(h));
    auto fail = [&](int rc, bg_fm* fm) {
        fclose(r.f);
        ls.f = nullptr;
        if (fm) bg_fm_free(fm);
        ls.fm = nullptr;
        return rc;
    };
    if (!r.ok || memcmp(magic, kMagic, 8) != 0 || h.n == 0 || h.n > (1ull << 40) || h.n_sym == 0 || h.n_sym > 256 || h.less_len == 0 ||
        h.less_len > 257 || h.occ_k == 0 || h.occ_k > 0xFFFFFFFFull || h.sa_kind > 2 || h.text_bytes > h.n || h.n_sample > h.n || h.n_extra > h.n)
        return fail(BG_ERR_IO, nullptr);
    // (sizes are checked against the file before anything of that size is allocated)
    {
        const long at = ftell(r.f);
        fseek(r.f, 0, SEEK_END);
        const uint64_t size = (uint64_t)ftell(r.f);
        fseek(r.f, at, SEEK_SET);
        auto pad = [](uint64_t b) { return (b + 7) & ~7ull; };
        uint64_t need = 8 + sizeof(Header) + pad(h.n_sym) + h.less_len * 8 + pad(h.n) + pad(h.text_bytes) + 8;
        if (h.sa_kind == 1) need += h.n * 8;
        if (h.sa_kind == 2) need += (h.n_sample + 2 * h.n_extra) * 8;
        if (size != need) return fail(BG_ERR_IO, nullptr);
    }
    std::vector<uint8_t> alphabet(h.n_sym);
    std::vector<uint64_t> less(h.less_len);
    r.get(alphabet.data(), alphabet.size());
    r.get(less.data(), less.size() * 8);
    std::vector<uint8_t> bwt;
    try {
        bwt.resize(h.n);
    } catch (...) {
        return fail(BG_ERR_OOM, nullptr);
    }
    r.get(bwt.data(), bwt.size());
    if (!r.ok) return fail(BG_ERR_IO, nullptr);
    bg_fm* fm = nullptr;
    // the caller's less travels with the index (an index built over another less than the BWT's own answers with it)
    int rc = bg_fm_build(ctx, bwt.data(), h.n, less.data(), (uint32_t)less.size(), (uint32_t)h.occ_k, alphabet.data(), (uint32_t)alphabet.size(), &fm);
    if (rc) return fail(rc, nullptr);
    ls.fm = fm;
    std::vector<uint8_t>().swap(bwt);
    // (h.wide is informational: a small index saved under a lowered fm_wide_from — the tests — loads into the layout THIS
    //  context chooses for its size)
    if (h.sa_kind == 1) {
        std::vector<uint64_t> sa(h.n);
        r.get(sa.data(), sa.size() * 8);
        if (!r.ok) return fail(BG_ERR_IO, fm);
        if ((rc = bg_fm_set_suffix_array(fm, sa.data(), h.n))) return fail(rc, fm);
    } else if (h.sa_kind == 2) {
        std::vector<uint64_t> sample(h.n_sample), rows(h.n_extra), pos(h.n_extra);
        r.get(sample.data(), sample.size() * 8);
        r.get(rows.data(), rows.size() * 8);
        r.get(pos.data(), pos.size() * 8);
        if (!r.ok) return fail(BG_ERR_IO, fm);
        if ((rc = bg_fm_set_sampled_suffix_array(fm, sample.data(), h.n_sample, (uint32_t)h.sa_rate, (uint8_t)h.sa_sentinel, rows.data(), pos.data(), h.n_extra)))
            return fail(rc, fm);
    }
    if (h.text_bytes) {
        std::vector<uint8_t> text(h.text_bytes);
        r.get(text.data(), text.size());
        if (!r.ok) return fail(BG_ERR_IO, fm);
        if ((rc = bg_fm_set_text(fm, text.data(), h.text_bytes))) return fail(rc, fm);
    }
    uint64_t sum = 0;
    const uint64_t want = r.mix.h;
    if (fread(&sum, 1, 8, r.f) != 8 || sum != want) return fail(BG_ERR_IO, fm);
    fclose(r.f);
    ls.f = nullptr;
    ls.fm = nullptr;
    *out = fm;
    return BG_OK;
}
}  // namespace

extern "C" int bg_fm_load(bg_ctx* ctx, const char* path, bg_fm** out) {
    if (!ctx || !path || !out) return BG_ERR_INVALID_ARG;
    *out = nullptr;
    LoadState ls;
    try {  // the host arrays of a large index (a raw suffix array: 8 bytes per symbol) may not fit: nothing throws across the ABI
        return fm_load_impl(ctx, path, out, ls);
    } catch (const std::bad_alloc&) {
        if (ls.f) fclose(ls.f);
        if (ls.fm) bg_fm_free(ls.fm);
        *out = nullptr;
        return BG_ERR_OOM;
    } catch (...) {
        if (ls.f) fclose(ls.f);
        if (ls.fm) bg_fm_free(ls.fm);
        *out = nullptr;
        return BG_ERR_HIP;
    }
}

// ---- what a handle can say about itself (a loaded handle has no caller-side arrays: FMDIndex::from(FMIndex) on a
// deserialized index, fmindex.rs:311-329, reads the BWT; the C++ mirror's len() the text length)
extern "C" int bg_fm_len(const bg_fm* fm, uint64_t* n) {
    if (!fm || !n) return BG_ERR_INVALID_ARG;
    *n = fm->wide ? fm->wdev.n : (uint64_t)fm->dev.n;
    return BG_OK;
}
extern "C" int bg_fm_less(const bg_fm* fm, uint64_t* less_out, uint32_t* less_len) {
    if (!fm || !less_len) return BG_ERR_INVALID_ARG;
    if (fm->h_less.empty()) return BG_ERR_UNSUPPORTED;
    *less_len = (uint32_t)fm->h_less.size();
    if (less_out) memcpy(less_out, fm->h_less.data(), fm->h_less.size() * 8);
    return BG_OK;
}
extern "C" int bg_fm_bwt_dev(const bg_fm* fm, uint8_t* d_bwt, void* stream) {
    if (!fm || !d_bwt) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(fm->ctx->device));
    return fm_decode_bwt_dev(fm, d_bwt, (hipStream_t)stream);
}
extern "C" int bg_fm_bwt(const bg_fm* fm, uint8_t* bwt) {
    if (!fm || !bwt) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(fm->ctx->device));
    const uint64_t n = fm->wide ? fm->wdev.n : (uint64_t)fm->dev.n;
    uint8_t* d = nullptr;
    BG_HIP(hipMalloc((void**)&d, std::max<uint64_t>(n, 16)));
    int rc = fm_decode_bwt_dev(fm, d, nullptr);
    if (!rc && hipMemcpy(bwt, d, n, hipMemcpyDeviceToHost) != hipSuccess) rc = BG_ERR_HIP;
    hipFree(d);
    return rc;
}
