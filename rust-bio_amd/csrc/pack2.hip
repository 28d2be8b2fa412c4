// 2-bit sequence streams (SURVEY.md section 8(f) row 4 "FASTQ ingest -> 2-bit pack"; north_star "packed 2-bit reads").
// A byte buffer — any concatenation of sequences, e.g. what bg_fastq_parse_dev emits — becomes one stream of 2-bit codes:
// symbol s sits in bits 2 (s % 16) .. +1 of little-endian dword s / 16.  Sequence boundaries do not matter to the
// packing: the offsets the byte flavours take (x_off / pat_off / seq_off, in symbols) address the packed stream as
// they are, so a sequence may start at any symbol.  Consumers: K5 (bg_fm_backward_search_packed_dev) and K1p
// (bg_align_batch_packed_dev).  rust-bio has no packed text type on this path (its Aligner and FMIndex take &[u8]:
// pairwise/mod.rs:591, fmindex.rs:144): this is the engine's own wire format.
#include "bg_common.h"

namespace {

// one thread per dword: 16 bytes in (one 16-byte load where the source is aligned), 32 bits out
__global__ __launch_bounds__(256) void pack2_kernel(const uint8_t* __restrict__ in, uint64_t n, uint32_t codes,
                                                    uint32_t* __restrict__ out, unsigned long long* __restrict__ n_invalid) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_words = (n + 15) / 16;
    uint32_t bad = 0;
    if (w < n_words) {
        const uint64_t s0 = w * 16;
        uint32_t b[4] = {0, 0, 0, 0};
        if (s0 + 16 <= n && ((uintptr_t)(in + s0) & 15) == 0) {
            const uint4 v = *(const uint4*)(in + s0);
            b[0] = v.x, b[1] = v.y, b[2] = v.z, b[3] = v.w;
        } else {
            for (uint32_t k = 0; k < 16 && s0 + k < n; k++) b[k >> 2] |= (uint32_t)in[s0 + k] << (8 * (k & 3));
        }
        const uint32_t c0 = codes & 0xFFu, c1 = (codes >> 8) & 0xFFu, c2 = (codes >> 16) & 0xFFu, c3 = codes >> 24;
        uint32_t word = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t ch = (b[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            const uint32_t code = ch == c1 ? 1u : ch == c2 ? 2u : ch == c3 ? 3u : 0u;
            if (s0 + k < n) {
                bad += (code == 0u && ch != c0) ? 1u : 0u;
                word |= code << (2 * k);
            }
        }
        out[w] = word;
    }
    if (n_invalid) {
#pragma unroll
        for (int o = 32; o; o >>= 1) bad += (uint32_t)__shfl_xor((int)bad, o);
        if ((threadIdx.x & 63) == 0 && bad) atomicAdd(n_invalid, (unsigned long long)bad);
    }
}

__global__ __launch_bounds__(256) void unpack2_kernel(const uint32_t* __restrict__ in, uint64_t n, uint32_t codes,
                                                      uint8_t* __restrict__ out) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 16 >= n) return;
    const uint32_t word = in[w];
    for (uint32_t k = 0; k < 16 && w * 16 + k < n; k++) out[w * 16 + k] = (uint8_t)(codes >> (8 * ((word >> (2 * k)) & 3u)));
}

}  // namespace

static uint32_t code_word(const uint8_t codes[4]) {
    return (uint32_t)codes[0] | (uint32_t)codes[1] << 8 | (uint32_t)codes[2] << 16 | (uint32_t)codes[3] << 24;
}

extern "C" int bg_pack2_dev(bg_ctx* ctx, const uint8_t* d_bytes, uint64_t n, const uint8_t codes[4], uint32_t* d_packed,
                            uint64_t* d_n_invalid, void* stream) {
    if (!ctx || !codes || (n && (!d_bytes || !d_packed))) return BG_ERR_INVALID_ARG;
    for (int a = 0; a < 4; a++)
        for (int b = a + 1; b < 4; b++)
            if (codes[a] == codes[b]) return BG_ERR_INVALID_ARG;
    if (n == 0) return BG_OK;
    BG_HIP(hipSetDevice(ctx->device));
    const uint64_t n_words = (n + 15) / 16;
    pack2_kernel<<<dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        d_bytes, n, code_word(codes), d_packed, (unsigned long long*)d_n_invalid);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

extern "C" int bg_unpack2_dev(bg_ctx* ctx, const uint32_t* d_packed, uint64_t n, const uint8_t codes[4], uint8_t* d_bytes,
                              void* stream) {
    if (!ctx || !codes || (n && (!d_bytes || !d_packed))) return BG_ERR_INVALID_ARG;
    if (n == 0) return BG_OK;
    BG_HIP(hipSetDevice(ctx->device));
    const uint64_t n_words = (n + 15) / 16;
    unpack2_kernel<<<dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(d_packed, n, code_word(codes),
                                                                                                 d_bytes);
    BG_HIP(hipGetLastError());
    return BG_OK;
}
