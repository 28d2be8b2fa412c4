// Host side of the 2-bit wire format (pack2.hip packs on the device): bytes -> 16 symbols per little-endian dword, symbol s in
// bits 2 (s % 16) .. + 1 of dword s / 16 — what the host-buffer entry points put on the PCIe link instead of the caller's
// bytes (bg_fm_backward_search_batch: a quarter of the pattern bytes to stage and to copy up).
// A byte that is none of the four codes makes the range "invalid" (the caller then stages the bytes themselves: the byte
// kernels know what an out-of-alphabet symbol means, fmindex.rs:144-208).
// AVX2 flavour: 32 bytes -> two dwords per iteration.  The byte -> code map is a 16-entry table on the LOW NIBBLE of the byte
// (pshufb), which works whenever the four code bytes differ in their low nibbles — 'A' 'C' 'G' 'T' in either case do; the
// map back (code -> byte, another pshufb) compared with the input tells a foreign byte.  Other alphabets take the scalar loop.
#include <cstdint>
#include <cstring>

#include "bg_common.h"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define BG_HAVE_AVX2_PACK 1
#endif

namespace bgpack {

namespace {

bool pack_scalar(const uint8_t* src, uint64_t n, const uint8_t lut[256], uint32_t* dst) {
    bool ok = true;
    uint64_t s = 0;
    for (; s + 16 <= n; s += 16) {
        uint32_t w = 0;
        for (int k = 0; k < 16; k++) {
            const uint8_t c = lut[src[s + k]];
            ok = ok && c < 4;
            w |= (uint32_t)(c & 3) << (2 * k);
        }
        dst[s / 16] = w;
    }
    if (s < n) {
        uint32_t w = 0;
        for (uint64_t k = 0; s + k < n; k++) {
            const uint8_t c = lut[src[s + k]];
            ok = ok && c < 4;
            w |= (uint32_t)(c & 3) << (2 * k);
        }
        dst[s / 16] = w;
    }
    return ok;
}

#ifdef BG_HAVE_AVX2_PACK
__attribute__((target("avx2"))) bool pack_avx2(const uint8_t* src, uint64_t n, const uint8_t codes[4], const uint8_t lut[256],
                                              uint32_t* dst) {
    uint8_t nib_code[16], nib_byte[16];
    for (int k = 0; k < 16; k++) {
        nib_code[k] = 0;
        nib_byte[k] = (uint8_t)(codes[0] ^ 0xff);  // never equal to an input byte with this nibble... unless it is: fixed below
    }
    for (int c = 0; c < 4; c++) {
        nib_code[codes[c] & 15] = (uint8_t)c;
        nib_byte[codes[c] & 15] = codes[c];
    }
    // a nibble no code has must not accept any byte: its "expected byte" gets a different low nibble than its slot
    for (int k = 0; k < 16; k++) {
        bool used = false;
        for (int c = 0; c < 4; c++) used = used || (codes[c] & 15) == k;
        if (!used) nib_byte[k] = (uint8_t)((k + 1) & 15);
    }
    const __m256i t_code = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)nib_code));
    const __m256i t_byte = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)nib_byte));
    const __m256i low4 = _mm256_set1_epi8(0x0f);
    const __m256i m1 = _mm256_set1_epi16(0x0401);        // c0 + 4 c1 per 16-bit lane
    const __m256i m2 = _mm256_set1_epi32(0x00100001);    // ... + 16 (c2 + 4 c3) per 32-bit lane
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                            0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    __m256i bad = _mm256_setzero_si256();
    uint64_t s = 0;
    for (; s + 32 <= n; s += 32) {
        const __m256i b = _mm256_loadu_si256((const __m256i*)(src + s));
        const __m256i nib = _mm256_and_si256(b, low4);
        const __m256i code = _mm256_shuffle_epi8(t_code, nib);
        bad = _mm256_or_si256(bad, _mm256_xor_si256(_mm256_shuffle_epi8(t_byte, nib), b));
        const __m256i p16 = _mm256_maddubs_epi16(code, m1);
        const __m256i p32 = _mm256_madd_epi16(p16, m2);
        const __m256i by = _mm256_shuffle_epi8(p32, gather);  // four packed bytes at the bottom of either 128-bit lane
        dst[s / 16] = (uint32_t)_mm256_extract_epi32(by, 0);
        dst[s / 16 + 1] = (uint32_t)_mm256_extract_epi32(by, 4);
    }
    bool ok = _mm256_testz_si256(bad, bad) != 0;
    if (s < n) ok = pack_scalar(src + s, n - s, lut, dst + s / 16) && ok;
    return ok;
}
#endif

}  // namespace

// bytes src[0, n) -> dwords dst[0, ceil(n / 16)); false if a byte is none of codes[0..3].  n may be any length; ranges handed
// to different threads must start at multiples of 16 symbols.
bool pack2_host(const uint8_t* src, uint64_t n, const uint8_t codes[4], uint32_t* dst) {
    uint8_t lut[256];
    memset(lut, 0xff, sizeof lut);
    for (int c = 0; c < 4; c++) lut[codes[c]] = (uint8_t)c;
#ifdef BG_HAVE_AVX2_PACK
    bool distinct = true;
    for (int a = 0; a < 4; a++)
        for (int b = a + 1; b < 4; b++) distinct = distinct && (codes[a] & 15) != (codes[b] & 15);
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (distinct && have_avx2) return pack_avx2(src, n, codes, lut, dst);
#endif
    return pack_scalar(src, n, lut, dst);
}

}  // namespace bgpack

// exported for the CPU tests (tests/test_host_pack2.py): same contract as bgpack::pack2_host, 1 = every byte was a code
extern "C" int bg_pack2_host(const uint8_t* bytes, uint64_t n, const uint8_t* codes, uint32_t* packed) {
    if ((n && (!bytes || !packed)) || !codes) return BG_ERR_INVALID_ARG;
    for (int a = 0; a < 4; a++)
        for (int b = a + 1; b < 4; b++)
            if (codes[a] == codes[b]) return BG_ERR_INVALID_ARG;
    return bgpack::pack2_host(bytes, n, codes, packed) ? 1 : 0;
}
