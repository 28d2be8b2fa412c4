// FASTQ ingest on the device: bio::io::fastq::Reader::read / Records and Record::check
// (/root/reference/src/io/fastq.rs:266-303, 388-410, 508-527) over a text that is already in HBM, and CIGAR
// emission (bio-types 1.0 Alignment::cigar) for a batch of alignment records.
//
// The reader is a sequential state machine over LINES (a record is a header line, sequence lines up to a line
// that starts with '+', then as many quality lines as there were sequence lines), but nearly every FASTQ is four
// lines per record.  So:
//   F1  line index — newline count per 4 KB chunk, scan of the counts, scatter of the line starts;
//   F2  per line: first byte, length after str::trim_end (Unicode White_Space), UTF-8 validity;
//   F3  four-line hypothesis, one thread per record: record k is lines 4k..4k+3 iff line 4k starts with '@',
//       4k+1 does not start with '+', 4k+2 starts with '+' and the quality trims to something.  By induction the
//       reader agrees with every record before the first one that fails;
//   F4  from that record on (wrapped, truncated or malformed input), one wavefront walks the lines with the
//       reference's rules, 64 lines of look-ahead per ballot;
//   F5  lengths -> exclusive scans -> offsets (seq_off is directly the x_off of bg_align_batch_dev);
//   F6  one wavefront per record gathers the trimmed lines and evaluates Record::check on the way.
// Everything is byte/integer work bound by HBM reads of the text (about three passes).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "bg_common.h"

namespace {

constexpr uint32_t kChunk = 4096;
struct LineInfo {
    uint32_t trim;  // bytes left by trim_end (the line's '\n' is whitespace too)
    uint8_t first;  // first byte of the line
    uint8_t bad;    // not UTF-8: read_line fails with io::ErrorKind::InvalidData
    uint16_t pad;
};
struct RecLines {  // lines of one record
    uint64_t hdr;     // header line
    uint64_t qual0;   // first quality line (may be >= n_lines at the end of the text)
    uint32_t n_seq;   // sequence lines hdr+1 .. hdr+n_seq; quality lines qual0 .. qual0+n_seq-1 (those that exist)
    uint32_t pad;
};

// ---- F1 ------------------------------------------------------------------------------------------------
// 16 bytes of the text for this thread (zero past the end; a zero is neither a newline nor a high byte)
__device__ __forceinline__ uint4 load16(const uint8_t* __restrict__ t, uint64_t len, uint64_t base, bool aligned) {
    if (aligned && base + 16 <= len) return *(const uint4*)(t + base);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (base + i < len) w[i >> 2] |= (uint32_t)t[base + i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint32_t newline_bits(uint32_t w) {  // bit 7 of every byte that equals '\n'
    const uint32_t x = w ^ 0x0a0a0a0au;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}
// cnt[chunk] = newlines; hi[chunk] = the chunk has a byte >= 0x80 (only such lines need the UTF-8 machinery)
__global__ __launch_bounds__(256) void fq_count_newlines_kernel(const uint8_t* __restrict__ t, uint64_t len, uint32_t* __restrict__ cnt,
                                                                uint8_t* __restrict__ hi) {
    const uint64_t base = (uint64_t)blockIdx.x * kChunk + threadIdx.x * 16u;
    const uint4 v = load16(t, len, base, ((uintptr_t)t & 15) == 0);
    uint32_t c = __popc(newline_bits(v.x)) + __popc(newline_bits(v.y)) + __popc(newline_bits(v.z)) + __popc(newline_bits(v.w));
    uint32_t h = ((v.x | v.y | v.z | v.w) & 0x80808080u) ? 1u : 0u;
    __shared__ uint32_t s[4], sh[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    h = __any(h) ? 1u : 0u;
    if ((threadIdx.x & 63) == 0) {
        s[threadIdx.x >> 6] = c;
        sh[threadIdx.x >> 6] = h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
        hi[blockIdx.x] = (uint8_t)(sh[0] | sh[1] | sh[2] | sh[3]);
    }
}
// exclusive scan of up to 2^32 items by one block (items are per-chunk / per-block partial sums: few).  4096 items per trip:
// four per thread, wavefront scans by shuffles, the sixteen wavefront totals scanned by the first wavefront — three barriers a
// trip.  (Until round 6 a Hillis-Steele scan in LDS, twenty barriers per 1024 items: 320 us for the 14 000 partial sums of a
// seed-and-extend pass, four times per pass.)
template <typename T>
__global__ __launch_bounds__(1024) void fq_scan_small_kernel(const T* __restrict__ in, uint64_t* __restrict__ out, uint64_t n, uint64_t* total) {
    __shared__ uint64_t s_w[16], s_wb[17];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto shfl_up64 = [](uint64_t v, int o) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, o), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), o);
        return (uint64_t)hi << 32 | lo;
    };
    uint64_t carry = 0;
    for (uint64_t b = 0; b < n; b += 4096) {
        const uint64_t i0 = b + (uint64_t)tid * 4;
        uint64_t v[4], tsum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = i0 + k < n ? (uint64_t)in[i0 + k] : 0;
            tsum += v[k];
        }
        uint64_t incl = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t u = shfl_up64(incl, o);
            if ((int)lane >= o) incl += u;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            const uint64_t w = lane < 16 ? s_w[lane] : 0;
            uint64_t wi = w;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const uint64_t u = shfl_up64(wi, o);
                if ((int)lane >= o) wi += u;
            }
            if (lane < 16) s_wb[lane] = wi - w;
            if (lane == 15) s_wb[16] = wi;
        }
        __syncthreads();
        uint64_t run = carry + s_wb[wave] + incl - tsum;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < n) out[i0 + k] = run;
            run += v[k];
        }
        carry += s_wb[16];
        __syncthreads();
    }
    if (tid == 0 && total) *total = carry;
}
__global__ __launch_bounds__(256) void fq_line_starts_kernel(const uint8_t* __restrict__ t, uint64_t len, const uint64_t* __restrict__ base,
                                                             uint64_t* __restrict__ ls) {
    const uint64_t b0 = (uint64_t)blockIdx.x * kChunk + threadIdx.x * 16u;
    const uint4 v = load16(t, len, b0, ((uintptr_t)t & 15) == 0);
    const uint32_t nb[4] = {newline_bits(v.x), newline_bits(v.y), newline_bits(v.z), newline_bits(v.w)};
    const uint32_t c = __popc(nb[0]) + __popc(nb[1]) + __popc(nb[2]) + __popc(nb[3]);
    __shared__ uint32_t s[256];
    s[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t u = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += u;
        __syncthreads();
    }
    uint64_t k = base[blockIdx.x] + s[threadIdx.x] - c;  // newlines before this thread's bytes
    if (c) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (nb[i >> 2] & (0x80u << (8 * (i & 3)))) ls[++k] = b0 + i + 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ls[0] = 0;
}

// ---- F2 ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_ws(uint32_t cp) {  // char::is_whitespace
    return (cp >= 9 && cp <= 13) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
           cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
__device__ bool valid_utf8(const uint8_t* s, uint64_t n) {
    uint64_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c < 0x80) {
            i++;
            continue;
        }
        int k;
        uint32_t cp, lo;
        if (c >= 0xC2 && c <= 0xDF) {
            k = 1, cp = c & 0x1F, lo = 0x80;
        } else if (c >= 0xE0 && c <= 0xEF) {
            k = 2, cp = c & 0x0F, lo = 0x800;
        } else if (c >= 0xF0 && c <= 0xF4) {
            k = 3, cp = c & 0x07, lo = 0x10000;
        } else {
            return false;
        }
        for (int j = 1; j <= k; j++) {
            if (i + j >= n || (s[i + j] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (s[i + j] & 0x3F);
        }
        if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += k + 1;
    }
    return true;
}
__device__ uint64_t trim_end(const uint8_t* s, uint64_t n) {  // str::trim_end on valid UTF-8
    while (n) {
        uint64_t b = n - 1;
        while (b > 0 && (s[b] & 0xC0) == 0x80) b--;
        const uint8_t c = s[b];
        uint32_t cp = c < 0x80 ? c : c < 0xE0 ? (c & 0x1F) : c < 0xF0 ? (c & 0x0F) : (c & 0x07);
        for (uint64_t j = b + 1; j < n; j++) cp = (cp << 6) | (s[j] & 0x3F);
        if (!is_ws(cp)) break;
        n = b;
    }
    return n;
}
__global__ __launch_bounds__(256) void fq_line_info_kernel(const uint8_t* __restrict__ t, const uint64_t* __restrict__ ls, uint64_t n_lines,
                                                           const uint8_t* __restrict__ chunk_hi, LineInfo* __restrict__ info) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    const uint64_t a = ls[i], n = ls[i + 1] - a;
    const uint8_t* s = t + a;
    bool hi = false;
    if (n)
        for (uint64_t c = a / kChunk; c <= (a + n - 1) / kChunk; c++) hi |= chunk_hi[c] != 0;
    LineInfo li;
    li.first = n ? s[0] : 0;
    li.pad = 0;
    if (!hi) {  // plain ASCII around here: trim_end is the ASCII white space at the end of the line
        uint64_t k = n;
        while (k && (s[k - 1] == ' ' || (s[k - 1] >= 9 && s[k - 1] <= 13))) k--;
        li.bad = 0;
        li.trim = (uint32_t)k;
    } else {
        li.bad = !valid_utf8(s, n);
        li.trim = li.bad ? 0u : (uint32_t)trim_end(s, n);
    }
    info[i] = li;
}

// ---- F3 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fq_four_line_kernel(const LineInfo* __restrict__ info, uint64_t n_rec4, unsigned long long* first_bad) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rec4) return;
    const LineInfo h = info[4 * k], s = info[4 * k + 1], p = info[4 * k + 2], q = info[4 * k + 3];
    const bool ok = !h.bad && !s.bad && !p.bad && !q.bad && h.first == '@' && s.first != '+' && p.first == '+' && q.trim != 0;
    if (!ok) atomicMin(first_bad, (unsigned long long)k);
}
__global__ __launch_bounds__(256) void fq_four_line_records_kernel(uint64_t n_rec, RecLines* __restrict__ rl) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rec) return;
    RecLines r;
    r.hdr = 4 * k;
    r.qual0 = 4 * k + 3;
    r.n_seq = 1;
    r.pad = 0;
    rl[k] = r;
}

// ---- F4: the reader's state machine from line `line0` on, one wavefront -------------------------------
// out[0] = records appended, out[1] = status, out[2] = line that raised the error
__global__ __launch_bounds__(64) void fq_walk_kernel(const LineInfo* __restrict__ info, uint64_t n_lines, uint64_t line0, RecLines* __restrict__ rl,
                                                     uint64_t rec0, uint64_t rec_cap, uint64_t* __restrict__ out) {
    const int lane = threadIdx.x;
    uint64_t h = line0, nrec = 0;
    int status = BG_FASTQ_OK;
    uint64_t err_line = 0;
    while (h < n_lines) {  // fastq.rs:266-274
        const LineInfo hi = info[h];
        if (hi.bad) {
            status = BG_FASTQ_IO, err_line = h;
            break;
        }
        if (hi.first != '@') {
            status = BG_FASTQ_MISSING_AT, err_line = h;
            break;
        }
        // fastq.rs:280-288: sequence lines until a line that starts with '+' (or the end of the text)
        uint64_t i = h + 1;
        bool io = false;
        while (true) {
            const uint64_t li = i + lane;
            bool stop = li >= n_lines, bad = false;
            if (!stop) {
                const LineInfo x = info[li];
                bad = x.bad;
                stop = bad || x.first == '+';
            }
            const uint64_t m = __ballot(stop);
            if (m) {
                const int f = __ffsll((long long)m) - 1;
                i += f;
                io = (__ballot(bad) >> f) & 1;
                break;
            }
            i += 64;
        }
        if (io || (i < n_lines && info[i].bad)) {
            status = BG_FASTQ_IO, err_line = i;
            break;
        }
        const uint64_t n_seq = i - (h + 1);
        const uint64_t q0 = i < n_lines ? i + 1 : i;  // the '+' line is consumed if it exists
        // fastq.rs:290-300: n_seq quality lines (those that exist), their trimmed lengths must not sum to 0
        uint64_t qsum = 0;
        bool qbad = false;
        uint64_t qbad_line = 0;
        for (uint64_t b = 0; b < n_seq; b += 64) {
            const uint64_t li = q0 + b + lane;
            uint32_t tl = 0;
            bool bad = false;
            if (b + lane < n_seq && li < n_lines) {
                const LineInfo x = info[li];
                bad = x.bad;
                tl = x.trim;
            }
            const uint64_t mb = __ballot(bad);
            if (mb && !qbad) {
                qbad = true;
                qbad_line = q0 + b + (__ffsll((long long)mb) - 1);
            }
            uint64_t v = tl;
#pragma unroll
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
            qsum += v;
        }
        if (qbad) {
            status = BG_FASTQ_IO, err_line = qbad_line;
            break;
        }
        if (qsum == 0) {
            status = BG_FASTQ_INCOMPLETE, err_line = h;
            break;
        }
        if (lane == 0 && rec0 + nrec < rec_cap) {
            RecLines r;
            r.hdr = h;
            r.qual0 = q0;
            r.n_seq = (uint32_t)n_seq;
            r.pad = 0;
            rl[rec0 + nrec] = r;
        }
        nrec++;
        h = min(q0 + n_seq, n_lines);
    }
    if (lane == 0) {
        out[0] = nrec;
        out[1] = (uint64_t)status;
        out[2] = err_line;
    }
}

// ---- F5 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fq_measure_kernel(const uint8_t* __restrict__ t, const uint64_t* __restrict__ ls, const LineInfo* __restrict__ info,
                                                         uint64_t n_lines, const RecLines* __restrict__ rl, uint64_t n_rec,
                                                         bg_fastq_record_t* __restrict__ recs, uint32_t* __restrict__ seq_len, uint32_t* __restrict__ qual_len) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rec) return;
    const RecLines r = rl[k];
    uint64_t sl = 0, ql = 0;
    for (uint32_t i = 0; i < r.n_seq; i++) {
        sl += info[r.hdr + 1 + i].trim;
        if (r.qual0 + i < n_lines) ql += info[r.qual0 + i].trim;
    }
    bg_fastq_record_t o = {};
    {  // fastq.rs:275-277: line[1..].trim_end().splitn(2, ' ')
        const uint64_t a = ls[r.hdr] + 1;
        const uint32_t n = info[r.hdr].trim - 1;  // the '@' is not whitespace: trim >= 1
        // the first space of the header: aligned 8-byte loads with a SWAR byte test (a byte at a time this was a chain of
        // 10-20 dependent loads per record and two thirds of the kernel's 0.19 ms)
        uint32_t sp = 0;
        {
            const uint8_t* p = t + a;
            while (sp < n && ((uintptr_t)(p + sp) & 7)) {
                if (p[sp] == ' ') goto found;
                sp++;
            }
            while (sp + 8 <= n) {
                const uint64_t w = *(const uint64_t*)(p + sp) ^ 0x2020202020202020ull;
                const uint64_t z = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;
                if (z) {
                    sp += (uint32_t)(__ffsll((long long)z) - 1) >> 3;
                    goto found;
                }
                sp += 8;
            }
            while (sp < n && p[sp] != ' ') sp++;
        found:;
        }
        o.id_off = a;
        o.id_len = sp;
        if (sp < n) {
            o.desc_off = a + sp + 1;
            o.desc_len = n - sp - 1;
            o.has_desc = 1;
        }
    }
    o.seq_len = (uint32_t)sl;
    o.qual_len = (uint32_t)ql;
    recs[k] = o;
    seq_len[k] = (uint32_t)sl;
    qual_len[k] = (uint32_t)ql;
}
// exclusive scan of uint32 lengths into uint64 offsets, three kernels
__global__ __launch_bounds__(256) void fq_block_sums_kernel(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ sums) {
    const uint64_t b0 = (uint64_t)blockIdx.x * 2048;
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) {
        const uint64_t j = b0 + (uint64_t)i * 256 + threadIdx.x;
        if (j < n) v += in[j];
    }
    __shared__ uint64_t s[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void fq_scan_apply_kernel(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ base,
                                                            uint64_t* __restrict__ out) {
    __shared__ uint64_t s[256];
    const uint64_t b0 = (uint64_t)blockIdx.x * 2048 + (uint64_t)threadIdx.x * 8;
    uint64_t loc[8], v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        loc[i] = b0 + i < n ? in[b0 + i] : 0;
        v += loc[i];
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint64_t u = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += u;
        __syncthreads();
    }
    uint64_t run = base[blockIdx.x] + s[threadIdx.x] - v;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (b0 + i <= n) out[b0 + i] = run;  // index n: the closing offset
        run += loc[i];
    }
}

// ---- F6: gather + Record::check (fastq.rs:388-410) ----------------------------------------------------
// G lanes copy n bytes (arbitrary alignments): dword stores on the destination's alignment, the source
// read as aligned dword pairs and funnel-shifted; collects "has a byte >= 0x80" and, for sequences, "has a byte
// that is not alphabetic or one of - . *" (fastq.rs:392-401).  (head and tail are at most 15 bytes, one per lane: G >= 16)
template <bool SEQ, int G>
__device__ __forceinline__ void copy_line(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, int lane, const uint8_t* t_end,
                                          bool& hi, bool& bad) {
    auto classify = [&](uint32_t c) {
        hi |= c >= 0x80;
        if (SEQ) bad |= !((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '-' || c == '.' || c == '*');
    };
    // 16-byte stores on the destination's alignment; a lane's 16 source bytes are five aligned dwords funnel-shifted
    // into four (dword pieces made 8 loads and 4 stores of the same 16 bytes: the kernel went from 0.41 to 0.29 ms per 1 M records)
    const uint32_t head = min(n, (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15));
    static_assert(G >= 16, "head and tail are up to 15 bytes, one per lane");
    if ((uint32_t)lane < head) {
        const uint8_t c = src[lane];
        dst[lane] = c;
        classify(c);
    }
    const uint32_t nq = (n - head) >> 4;
    for (uint32_t j = lane; j < nq; j += G) {
        const uint8_t* p = src + head + 16 * j;
        const uint32_t sh = (uint32_t)((uintptr_t)p & 3);
        const uint8_t* pa = p - sh;
        uint32_t w[4];
        if (pa + 20 <= t_end) {
            uint32_t r[5];
#pragma unroll
            for (int q = 0; q < 5; q++) r[q] = *(const uint32_t*)(pa + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; q++) w[q] = __builtin_amdgcn_alignbyte(r[q + 1], r[q], sh);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                w[q] = (uint32_t)p[4 * q] | ((uint32_t)p[4 * q + 1] << 8) | ((uint32_t)p[4 * q + 2] << 16) | ((uint32_t)p[4 * q + 3] << 24);
        }
        *(uint4*)(dst + head + 16 * j) = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (SEQ) {
                classify(w[q] & 0xff);
                classify((w[q] >> 8) & 0xff);
                classify((w[q] >> 16) & 0xff);
                classify(w[q] >> 24);
            } else {
                hi |= (w[q] & 0x80808080u) != 0;
            }
        }
    }
    const uint32_t done = head + 16 * nq;
    if ((uint32_t)lane < n - done) {
        const uint8_t c = src[done + lane];
        dst[done + lane] = c;
        classify(c);
    }
}
// G lanes per record: a record of short reads is two lines of ~40 dwords behind a chain of dependent loads (record ->
// lines -> line starts -> text) — with a whole wavefront per record most lanes idle and the kernel is as long as the
// number of wavefronts times that chain; 16 lanes per record: a quarter of the wavefronts (0.67 -> 0.40 ms per 1 M
// records of 150 bp).  Long lines (the average line beyond 256 bytes) keep the wavefront per record.
template <int G>
__global__ __launch_bounds__(256) void fq_gather_kernel(const uint8_t* __restrict__ t, uint64_t len, const uint64_t* __restrict__ ls,
                                                        const LineInfo* __restrict__ info, uint64_t n_lines, const RecLines* __restrict__ rl,
                                                        uint64_t n_rec, bg_fastq_record_t* __restrict__ recs, const uint64_t* __restrict__ seq_off,
                                                        const uint64_t* __restrict__ qual_off, uint8_t* __restrict__ seq, uint8_t* __restrict__ qual) {
    const int lane = threadIdx.x % G;
    const uint64_t k = (uint64_t)blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
    if (k >= n_rec) return;  // (uniform over the G lanes of a record)
    const RecLines r = rl[k];
    uint64_t so = seq_off[k], qo = qual_off[k];
    bool seq_hi = false, seq_bad = false, qual_hi = false, unused = false;
    for (uint32_t i = 0; i < r.n_seq; i++) {
        const uint64_t l = r.hdr + 1 + i;
        const uint32_t n = info[l].trim;
        copy_line<true, G>(seq + so, t + ls[l], n, lane, t + len, seq_hi, seq_bad);
        so += n;
        if (r.qual0 + i < n_lines) {
            const uint64_t lq = r.qual0 + i;
            const uint32_t nq = info[lq].trim;
            copy_line<false, G>(qual + qo, t + ls[lq], nq, lane, t + len, qual_hi, unused);
            qo += nq;
        }
    }
    {  // "any" over the G lanes of the record
        const int sh = (threadIdx.x & 63) / G * G;
        const uint64_t gm = (G == 64 ? ~0ull : ((1ull << G) - 1)) << sh;
        seq_hi = (__ballot(seq_hi) & gm) != 0;
        seq_bad = (__ballot(seq_bad) & gm) != 0;
        qual_hi = (__ballot(qual_hi) & gm) != 0;
    }
    if (lane == 0) {
        bg_fastq_record_t o = recs[k];
        o.seq_off = seq_off[k];
        o.qual_off = qual_off[k];
        o.check = o.id_len == 0    ? BG_FQCHECK_EMPTY_ID
                  : seq_hi         ? BG_FQCHECK_NONASCII_SEQ
                  : seq_bad        ? BG_FQCHECK_INVALID_SEQ
                  : qual_hi        ? BG_FQCHECK_NONASCII_QUAL
                  : o.seq_len != o.qual_len ? BG_FQCHECK_UNEQUAL
                                            : BG_FQCHECK_OK;
        recs[k] = o;
    }
}

// ---- FF: the whole reader in ONE pass over the text (round 6) ---------------------------------------------------------
// F1 .. F6 above read the text five times (newline count, line starts, line info, header split, gather: 2.46 GB of HBM
// traffic for 0.70 GB of algorithmic bytes, profiles/r05_pmc_traffic.json) through seven kernels and three host round
// trips.  Nearly every FASTQ is four lines per record in plain ASCII; for such a text ONE kernel does everything, each
// byte read from HBM once:
//   * a tile of 16 KB per block, in block order (fq_lookback: why no ticket, and why nothing can hang), staged in LDS with the
//     1 KB in front of it; every thread takes four 16-byte pieces (coalesced), finds the newlines (SWAR) and their ordinals
//     (block scan of the packed piece counts) -> the tile's newline list, while one wavefront finds the four newlines in
//     front of the tile (the lines a record of this tile may begin with);
//   * which lines are headers is GUESSED from the text (a header begins with '@', the line two further on with '+': every
//     line start the block holds votes) and confirmed by the look-back below; a record belongs to the tile its fourth line
//     ends in: the thread that owns it checks the four-line hypothesis of F3 (line 4k starts with '@', 4k+1 not with '+',
//     4k+2 with '+', the quality trims to something), trims the lines (ASCII white space: str::trim_end when every byte is
//     ASCII), splits the header at its first space (fastq.rs:275-277);
//   * the records' sequence / quality lengths are scanned in the block; the tile's three totals — lines, sequence bytes,
//     quality bytes — go through ONE decoupled look-back (Merrill & Garland: a 64-bit word per tile and value holding flag +
//     value; two levels, a wavefront per value) -> the global index of its first record, seq_off / qual_off of every
//     record, the fixed fields of bg_fastq_record_t;
//   * every thread copies 16-byte pieces of the tile's sequence / quality output from LDS and evaluates Record::check on them
//     (SWAR), a table saying which record a piece begins in.
// Whatever the fast path cannot promise — a byte >= 0x80, a line count that is not a multiple of four, a record that fails
// the hypothesis, a wrong guess, a tile with more than 2048 newlines or 256 records, a line that begins more than 31 KB in
// front of the tile its record ends in — raises one flag; the host reads {lines, flag} back (the call's ONE
// synchronisation) and, if it is up, runs F1 .. F6, which are exact for any input and overwrite whatever the fused kernel
// wrote.  Every write of the fused kernel stays inside the caller's buffers whatever the text holds (a line is some tile's
// sequence line at most once, whatever the tiles guess).
// Round 6, 1 M records of 150 bp (323 MB): 0.436 ms per call, 741 GB/s of text (round 5's seven kernels: 1.5 ms); the
// steps from the first one-pass build (0.70 ms) are in profiles/r06_ingest_experiments.txt.
constexpr uint32_t kTile = 16384, kNlCap = 2048, kRecCap = 256;  // (LDS: 26 KB per block — tile + halo 17.4, newline list 4, records 4 — six blocks per CU)
constexpr int kHalo = 4;
constexpr int32_t kHaloBytes = 1024;  // bytes in front of the tile kept in LDS too (lines of the tile's first record begin there)
#ifndef FQ_LDS_TILE  // 1: the tile's bytes are staged in LDS — the text is read from memory once (FETCH_SIZE + WRITE_SIZE 1.07 x the
                     // algorithmic bytes); 0: owners and the copy phase re-read them through L2, which does not hold them: 5 - 8 %
                     // faster per call, 1.77 x the algorithmic bytes (profiles/r06_ingest_experiments.txt)
#define FQ_LDS_TILE 1
#endif
constexpr bool kLdsTile = FQ_LDS_TILE != 0;
constexpr uint64_t kFlagAgg = 1ull << 62, kFlagPre = 2ull << 62, kValMask = (1ull << 62) - 1;

struct FusedArgs {
    const uint8_t* t;
    uint64_t len;
    uint64_t* tiles;     // [2][3][n_tiles]: flag | value words of the look-back's two levels x (newlines, sequence bytes, quality bytes)
    uint64_t* out;       // [0] lines, [1] irregular, [2] sequence bytes, [3] quality bytes
    bg_fastq_record_t* recs;
    uint64_t rec_cap;
    uint8_t *seq, *qual;
    uint64_t *seq_off, *qual_off;
    uint32_t n_tiles;
};

// Exclusive prefixes of this tile's aggregates (NV values, their words sit `stride` apart) over all tiles before it: a
// decoupled look-back (Merrill & Garland) on TWO levels.  Flag and value share one 64-bit word, so the word is all that
// travels between blocks: RELAXED agent-scope atomics (coherent across the XCDs' L2s by themselves; acquire / release here
// would flush and invalidate whole caches around every word — the first build did, and ran 5 ms instead of 0.4).
//   level 1, l1[tile]: the tile's own aggregate, later its inclusive prefix;
//   level 2, l2[tile]: the sum of the 65 tiles ending with it (its level-1 window + itself), later its inclusive prefix.
// Wavefront k works on value k: lane i reads l1[id - 1 - i] and, at the same time, lanes 0 .. 15 read l2[id - 65 (i + 1)].  If
// the level-1 window holds an inclusive prefix the walk ends there; otherwise its sum (64 tiles) goes out as this tile's level-2
// word and the level-2 words carry on from tile id - 65, sixty-five tiles a lane.  Why two levels (round 6, tools/exp/
// timeline_ingest.py): a word written by one block is seen by another ~2.5 us later, and with ~1500 blocks resident, started
// 21 ns apart, the ~600 tiles in front of a tile are themselves still looking back — one level of 64- or 128-tile windows walked
// five windows deep, 13 of a block's 30 us (and the longer the look-backs take, the more tiles are in one: the walk feeds itself).
// The level-2 word of a tile is there one round trip after its aggregate, whatever the tiles in front of it are doing.
// A tile's number is its block's index (no ticket: 19 715 blocks taking one atomicAdd each on ONE address cost 225 us by
// themselves — tools/microbench/ub_ticket.hip): blocks are dispatched in index order, so the tiles a block waits for are
// running or done.  Should that ever not hold, the wait is BOUNDED: a lane that polled kMaxPolls times raises `*gave_up` (the
// kernel's "irregular" flag: the host then takes the general kernels) and goes on as if it had seen a prefix — nothing can hang.
constexpr uint32_t kMaxPolls = 1u << 20;  // (~1 s of polling; a tile normally waits a few polls)
constexpr int kL2Lanes = 16, kL2Span = 65;
template <int NV>
__device__ void fq_lookback(uint64_t* l1, uint64_t* l2, uint64_t stride, uint32_t id, const uint64_t (&agg)[NV], uint64_t (&excl)[NV], uint64_t* s_tmp /* NV words */,
                            uint64_t* gave_up, uint32_t wave /* the caller's (virtual) wavefront number */) {
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    static_assert(NV <= 4, "a wavefront per value");
#pragma unroll
    for (int k = 0; k < NV; k++) excl[k] = 0;
#ifdef FQ_KO_LOOKBACK  // (knock-out builds, tools/exp/ko_build.sh: wrong results, timing only)
    return;
#endif
    auto st = [](uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld = [](const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (id == 0) {
        if (tid == 0)
#pragma unroll
            for (int k = 0; k < NV; k++) st(&l1[k * stride], kFlagPre | agg[k]), st(&l2[k * stride], kFlagPre | agg[k]);
        return;
    }
    if ((int)wave < NV) {
        uint64_t my = 0;
#pragma unroll
        for (int k = 0; k < NV; k++)
            if ((int)wave == k) my = agg[k];
        uint64_t* const p1 = l1 + wave * stride;
        uint64_t* const p2 = l2 + wave * stride;
        if (lane == 0) st(&p1[id], kFlagAgg | my);
        auto wave_sum = [](uint64_t val) {
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)val, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(val >> 32), o);
                val += (uint64_t)hi << 32 | lo;
            }
            return val;
        };
        auto spin = [&](const uint64_t* p, uint64_t v) {  // until the word is written
            uint32_t polls = 0;
            while ((v >> 62) == 0 && ++polls < kMaxPolls) v = ld(p);
            if ((v >> 62) == 0) {  // never seen: give up (see above)
                atomicOr((unsigned long long*)gave_up, 1ull);
                v = kFlagPre;
            }
            return v;
        };
        const int64_t i1 = (int64_t)id - 1 - (int64_t)lane;
        int64_t i2 = (int64_t)id - kL2Span * ((int64_t)lane + 1);
        // both levels' first reads together (in front of tile 0: an inclusive prefix of 0)
        uint64_t v1 = i1 >= 0 ? ld(&p1[i1]) : kFlagPre;
        uint64_t v2 = (int)lane >= kL2Lanes ? kFlagAgg : i2 >= 0 ? ld(&p2[i2]) : kFlagPre;
        if (i1 >= 0) v1 = spin(&p1[i1], v1);
        const uint64_t pm1 = __ballot((v1 >> 62) == 2);
        uint64_t val = v1 & kValMask;
        if (pm1 && (int)lane > __ffsll((long long)pm1) - 1) val = 0;  // lanes beyond the nearest inclusive prefix do not count
        uint64_t e = wave_sum(val);
        if (!pm1) {
            if (lane == 0) st(&p2[id], kFlagAgg | ((e + my) & kValMask));
            for (;;) {
                if ((int)lane < kL2Lanes && i2 >= 0) v2 = spin(&p2[i2], v2);
                const uint64_t pm2 = __ballot((int)lane < kL2Lanes && (v2 >> 62) == 2);
                uint64_t val2 = (int)lane < kL2Lanes ? v2 & kValMask : 0;
                if (pm2 && (int)lane > __ffsll((long long)pm2) - 1) val2 = 0;
                e += wave_sum(val2);
                if (pm2) break;
                i2 -= (int64_t)kL2Span * kL2Lanes;
                v2 = (int)lane >= kL2Lanes ? kFlagAgg : i2 >= 0 ? ld(&p2[i2]) : kFlagPre;
            }
        }
        if (lane == 0) {
            st(&p1[id], kFlagPre | ((e + my) & kValMask));
            st(&p2[id], kFlagPre | ((e + my) & kValMask));
            s_tmp[wave] = e;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) excl[k] = s_tmp[k];
    __syncthreads();
}

__device__ __forceinline__ uint32_t newline_mask16(const uint4 v) {  // bit i: byte i of the piece is '\n'
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t b = newline_bits(w[q]);  // bit 7 of every matching byte
        m |= (((b >> 7) & 1u) | ((b >> 14) & 2u) | ((b >> 21) & 4u) | ((b >> 28) & 8u)) << (4 * q);
    }
    return m;
}

typedef int16_t fq_rel_t;   // positions relative to the tile, lengths and the newline list: 16 bits (see FusedRec)
typedef uint16_t fq_len_t;
typedef int16_t fq_nl_t;
struct FusedRec {  // what the copy phase needs of a record (LDS; 16 bytes: with the newline list in 16 bits too the block's
                   // LDS is 26 KB, six blocks per CU — at 38 KB it was four, and the kernel waits on memory most of its life)
    fq_rel_t seq_rel, qual_rel;  // line starts relative to the tile (may be negative: a line that begins in front of the tile;
                                // the halo search stops 31 KB in front of it)
    fq_len_t seq_n, qual_n;     // (a record's lines lie between 31 KB in front of the tile and its end: < 48 KB)
    fq_len_t seq_dst, qual_dst; // destinations relative to the tile's first sequence / quality byte (likewise)
    uint32_t flags;             // 1: the id is empty
};
static_assert(sizeof(FusedRec) == 16, "");
constexpr int kHaloWindows = 31;  // 1 KB windows the search for the four newlines in front of a tile may take

#ifdef FQ_TIMELINE  // (variant builds only, tools/exp/timeline_ingest.py: wall-clock stamps of thread 0 at the phase boundaries)
__device__ uint64_t g_fq_tl[16 * 32768];
#define FQ_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 32768) g_fq_tl[16 * blockIdx.x + (k)] = wall_clock64(); } while (0)
extern "C" int bg_debug_fq_timeline(uint64_t* out, size_t n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fq_tl), n * 8); }
#else
#define FQ_STAMP(k)
#endif
__global__ __launch_bounds__(256) void fq_fused_kernel(const FusedArgs a) {
    FQ_STAMP(0);
    __shared__ fq_nl_t s_nl[kHalo + kNlCap + 4];  // newline positions relative to the tile; the copy phase's piece tables later
    __shared__ FusedRec s_rec[kRecCap];
    // the tile's bytes: what the records' owners look at (first bytes, line ends, headers) and what the copy phase reads comes
    // from here — the text is read from memory ONCE (a second read of it, 16 KB per block with 2 000 blocks in flight, does not
    // stay in a 4 MB L2); only lines that begin in the previous tile are read from there
    __shared__ uint4 s_tile4[kLdsTile ? (kHaloBytes + kTile) / 16 + 2 : 1];  // [halo: the 1 KB in front of the tile][tile][pad]
    static_assert(sizeof(s_nl) >= 2 * (kTile / 16 + 4) && kRecCap <= 256, "");
    uint8_t (*s_prec)[kTile / 16 + 4] = (uint8_t (*)[kTile / 16 + 4])s_nl;  // the record a 16-byte piece of the tile's sequence / quality output begins in
#ifdef FQ_PAD_LDS
    __shared__ uint32_t s_padlds[FQ_PAD_LDS / 4];
    if (threadIdx.x == 9999) s_padlds[0] = 1;
#endif
    __shared__ uint64_t s_w[8];
    __shared__ uint64_t s_lb[24];
    __shared__ uint64_t s_bc[8];  // [4] irregular: seen before the records are looked at, [5] irregular: seen later, [6] phases of the line index the tile's lines rule out
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (one tile per block.  Persistent blocks that take tile after tile were measured on the round's first one-pass build,
    //  profiles/r06_ingest_experiments.txt: 0.74 ms per call against 0.65 — and 1.86 / 1.15 / 0.81 / 0.74 ms with 1 / 2 / 4 / 8
    //  blocks per CU: a tile's own chain of round trips is what a block's life consists of, ~30 us)
    if (tid == 0) s_bc[4] = s_bc[5] = s_bc[6] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    // The phases only one wavefront works in (the records' owners are the first threads, the look-back takes a wavefront per
    // value, the search in front of the tile one) go by a VIRTUAL wavefront number.  -DFQ_ROTATE lets it rotate with the tile,
    // in case every block's wavefront 0 lands on the same SIMD of its CU and that SIMD carries all of that work: measured in
    // round 6, no difference (0.438 against 0.439 ms per call) — it is the physical number.
#ifdef FQ_ROTATE
    const uint32_t vw = (wave - tile) & 3u;
#else
    const uint32_t vw = wave;
#endif
    const uint32_t vtid = vw * 64 + lane;
    const uint64_t t0 = (uint64_t)tile * kTile;
    const bool last_tile = tile + 1 == a.n_tiles;
    const bool aligned = ((uintptr_t)a.t & 15) == 0;
    // (the 1 KB in front of the tile — where the search for the four newlines in front of it begins — is asked for now, with
    //  the tile: one memory latency, not two in a row)
    uint4 hv = make_uint4(0, 0, 0, 0);
    if (vw == 3 && t0 >= (uint64_t)kHaloBytes) hv = load16(a.t, t0, t0 - kHaloBytes + (uint64_t)lane * 16, aligned);
    // ---- newlines of the tile
    uint32_t m[4];
    uint64_t packed = 0;
    bool hi = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t base = t0 + ((uint64_t)j * 256 + tid) * 16;
        const uint4 v = base < a.len ? load16(a.t, a.len, base, aligned) : make_uint4(0, 0, 0, 0);
        if (kLdsTile) s_tile4[kHaloBytes / 16 + j * 256 + tid] = v;
        m[j] = newline_mask16(v);
        hi |= ((v.x | v.y | v.z | v.w) & 0x80808080u) != 0;
        packed |= (uint64_t)__popc(m[j]) << (16 * j);
    }
    if (kLdsTile && tid < 2) s_tile4[(kHaloBytes + kTile) / 16 + tid] = make_uint4(0, 0, 0, 0);  // (pad: 16-byte reads may run past the tile's last byte)
    // a text that does not end in '\n': its last line ends at the end of the text (one more "newline", behind all others)
    const uint32_t extra = (last_tile && a.t[a.len - 1] != '\n') ? 1u : 0u;
    // inclusive scan of the packed counts over the block (row j = pieces j * 256 ..: four scans in one)
    uint64_t inc = packed;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, o), hh = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), o);
        if ((int)lane >= o) inc += (uint64_t)hh << 32 | lo;
    }
    if (lane == 63) s_w[wave] = inc;
    if (__any(hi) && lane == 0) s_bc[4] = 1;
    __syncthreads();
    FQ_STAMP(1);  // tile loaded, counted
    uint64_t wbase = 0, rows = 0;
    for (uint32_t w = 0; w < 4; w++) {
        if (w < wave) wbase += s_w[w];
        rows += s_w[w];
    }
    const uint64_t excl = wbase + inc - packed;  // per row: newlines of the row's pieces before this thread's
    uint32_t row_total[4], row_base[4];
    uint32_t T = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        row_total[j] = (uint32_t)(rows >> (16 * j)) & 0xFFFFu;
        row_base[j] = T;
        T += row_total[j];
    }
    const uint32_t T_all = T + extra;
    bool irregular = T_all > kNlCap;
    // ---- wavefront 3: the four newlines in front of the tile; then the whole block: lines before the tile
#ifdef FQ_KO_HALO
    if (false) {
#else
    if (vw == 3) {
#endif
        int need = kHalo;
        uint64_t end = t0;  // window [end - 1024, end)
        int windows = 0;
        while (need > 0 && end > 0 && windows < kHaloWindows) {
            const uint64_t wstart = end >= 1024 ? end - 1024 : 0;
            const uint64_t base = wstart + (uint64_t)lane * 16;
            uint32_t mm = 0;
            if (base < end) {
                const bool first = windows == 0 && wstart + (uint64_t)kHaloBytes == t0;  // the 1 KB in front of the tile: loaded above
                const uint4 v = first ? hv : load16(a.t, end, base, aligned && (wstart & 15) == 0);  // (bytes from `end` on read as 0)
                mm = newline_mask16(v);
                if (kLdsTile && first) s_tile4[lane] = v;
            }
            const uint32_t c = __popc(mm);
            uint32_t pre = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = (uint32_t)__shfl_up((int)pre, o);
                if ((int)lane >= o) pre += u;
            }
            const uint32_t total = (uint32_t)__shfl((int)pre, 63);
            uint32_t k = pre - c;  // ordinal of this lane's first newline inside the window
            const int take = min(need, (int)total);
            uint32_t bits = mm;
            while (bits) {
                const int b = __ffs((int)bits) - 1;
                bits &= bits - 1;
                const int from_end = (int)total - 1 - (int)k;  // 0: the window's last newline
                if (from_end < take) s_nl[need - 1 - from_end] = (fq_nl_t)((int64_t)(base + b) - (int64_t)t0);
                k++;
            }
            need -= take;
            end = wstart;
            windows++;
        }
        if (need > 0 && end > 0 && lane == 0) s_bc[4] = 1;  // a line that begins more than 31 KB in front of the tile: not for this path
        if ((int)lane < need) s_nl[lane] = (fq_nl_t)(-1 - (int32_t)(int64_t)t0);  // (end == 0 was reached: t0 <= 31 KB)  // in front of the text: "newline" at position -1
    }
    // ---- every thread: its newlines into the list
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t k = row_base[j] + ((uint32_t)(excl >> (16 * j)) & 0xFFFFu);
        uint32_t bits = m[j];
        const uint32_t rel = (j * 256 + tid) * 16;
        while (bits && !irregular) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            s_nl[kHalo + k++] = (fq_nl_t)(rel + b);
        }
    }
    if (extra && tid == 0 && !irregular) s_nl[kHalo + T] = (fq_nl_t)(a.len - t0);
    FQ_STAMP(2);  // newline list written (thread 0 is not in wavefront 3: the halo search is not in this)
    // ---- which lines are headers?  The line index of the tile's first line (mod 4) says so, and that is a look-back away; but
    // the text itself nearly always says it too: a header begins with '@' and the line two further on with '+' (the hypothesis
    // the records are held to anyway, below).  Every line whose first byte the block holds votes: phases it rules out are OR-ed
    // into s_bc[6]; if exactly one of the four is left the records are read on that GUESS at once, and the tile's three totals
    // (lines, sequence bytes, quality bytes) go through ONE look-back instead of two in a row (the timeline of round 6: the two
    // look-backs were 13 of a block's 35 us).  The look-back's line count then confirms the guess — a wrong one (a tile of
    // lines that all look like headers) raises the flag: the host takes the general kernels.  No unique guess (a tile inside a
    // long line): the line count first, as before.
    typedef const __attribute__((address_space(3))) uint8_t* lds_bytes_t;
    const lds_bytes_t l_tile = (lds_bytes_t)(const uint8_t*)s_tile4 + kHaloBytes;  // (tile-relative positions index it: -kHaloBytes .. kTile + pad)
    const int32_t lds_lo = kLdsTile ? (t0 ? -kHaloBytes : 0) : 1 << 30;  // positions from here on are in LDS
    __syncthreads();  // (the list, wavefront 3's four entries in front of it, the tile's bytes)
    {
        uint32_t imp = 0;
        const uint32_t n_list = irregular ? 0 : kHalo + T_all;
        const int32_t lim = (int32_t)min((uint64_t)kTile, a.len - t0);
        for (uint32_t jj = tid; jj < n_list; jj += 256) {
            const int32_t nl = s_nl[jj], pos = nl + 1;  // the line behind newline jj
            const bool dup = jj + 1 < n_list && s_nl[jj + 1] == nl;  // (in front of the text: several entries say "-1", the last one counts)
            if (kLdsTile && !dup && pos >= lds_lo && pos < lim) {
                const uint8_t c = l_tile[pos];
                if (c != '@') imp |= 1u << (jj & 3);        // a header follows newline jj  <=>  j0 == jj mod 4  (kHalo is 4)
                if (c != '+') imp |= 1u << ((jj + 2) & 3);  // the '+' line follows newline jj  <=>  j0 == jj - 2 mod 4
            }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) imp |= (uint32_t)__shfl_xor((int)imp, o);
        if (lane == 0 && imp) atomicOr((unsigned long long*)&s_bc[6], (unsigned long long)imp);
    }
    __syncthreads();
    if (s_bc[4]) irregular = true;  // (a byte >= 0x80, or the lines in front of the tile begin too far away: the list's first
                                    //  entries are not set — no record of this tile is looked at; the host takes the general kernels)
    const uint32_t cand = ~(uint32_t)s_bc[6] & 15u;
#ifdef FQ_NO_GUESS
    const bool spec = false;
#else
    const bool spec = !irregular && __popc(cand) == 1;
#endif
    uint64_t lines_before = 0;
    uint32_t j0;
    if (spec) {
        j0 = (uint32_t)__ffs((int)cand) - 1;
    } else {
        const uint64_t agg[1] = {T_all};
        uint64_t ex1[1];
        fq_lookback<1>(a.tiles, a.tiles + 3 * (uint64_t)a.n_tiles, a.n_tiles, tile, agg, ex1, s_lb, &a.out[1], vw);
        lines_before = ex1[0];
        j0 = (3u - (uint32_t)(lines_before & 3)) & 3u;
    }
    FQ_STAMP(3);  // look-back #1 done (or guessed)
    // ---- the records whose fourth line ends in this tile
    uint32_t nr = (!irregular && j0 < T_all) ? (T_all - j0 + 3) / 4 : 0;
    if (nr > kRecCap) {
        irregular = true;
        nr = 0;
    }
#ifdef FQ_KO_RECORDS
    nr = 0;
#endif
    const uint8_t* tb = a.t + t0;  // (relative positions are added to this; a line of the previous tile: negative)
    const uint8_t* s_tile = (const uint8_t*)s_tile4 + kHaloBytes;  // (tile-relative positions index it: -kHaloBytes .. kTile + pad)
    // a byte of the text by tile-relative position: from the LDS copy where that holds it.  (Two loads in two address spaces
    // under a branch, NOT one load through a selected generic pointer: flat loads faulted — aperture violation — on in-range
    // LDS addresses once the tile's buffer sat at LDS offset 0; and a DS read is the cheaper instruction.)
    auto gb = [&](int32_t rel) -> uint8_t {
        uint8_t v;
        if (rel >= lds_lo)
            v = l_tile[rel];
        else
            v = tb[rel];
        return v;
    };
    auto is_ws_ascii = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); };
    uint64_t lens = 0;  // seq | qual << 32 of this thread's record
    bg_fastq_record_t rr = {};
    bool bad_rec = false;
    if (vtid < nr) {
        const uint32_t j = kHalo + j0 + 4 * vtid;  // list index of the record's last newline
        const int32_t e_h = s_nl[j - 3], e_s = s_nl[j - 2], e_p = s_nl[j - 1], e_q = s_nl[j];
        const int32_t b_h = s_nl[j - 4] + 1, b_s = e_h + 1, b_p = e_s + 1, b_q = e_p + 1;
        // the bytes this needs — the three first bytes, the byte in front of each of the three trimmed line ends, the head of
        // the header — are loaded together before any of them is looked at (a chain of dependent L2 reads otherwise)
        const uint8_t f_h = gb(b_h), f_s = gb(b_s), f_p = gb(b_p);  // (b_s <= e_s < len, b_p <= e_p < len: bytes of the text)
        const uint8_t l_h = e_h > b_h ? gb(e_h - 1) : (uint8_t)'x', l_s = e_s > b_s ? gb(e_s - 1) : (uint8_t)'x', l_q = e_q > b_q ? gb(e_q - 1) : (uint8_t)'x';
        // header: 8 bytes at a time from the 8-byte boundary at or below its second byte (bytes in front of it masked off)
        const int32_t hrel = b_h + 1;
        const uint32_t mis = (uint32_t)hrel & 7u;  // (tile starts are multiples of 16 KB: LDS and text share the alignment when the text is 8-byte aligned)
        const int32_t hrel0 = hrel - (int32_t)mis;
        const bool h_lds = hrel0 >= lds_lo;  // the two words lie in the LDS copy (halo, tile, pad)
        const uint64_t* hw = (const uint64_t*)(tb + hrel0);
        const bool h2 = h_lds || (((uintptr_t)hw & 7) == 0 && (const uint8_t*)hw >= a.t && (const uint8_t*)(hw + 2) <= a.t + a.len);
        uint64_t w0 = 0, w1 = 0;  // (two words: most headers' ids end inside them)
        if (h_lds) {
            w0 = *(const uint64_t*)(s_tile + hrel0);
            w1 = *(const uint64_t*)(s_tile + hrel0 + 8);
        } else if (h2) {
            w0 = hw[0];
            w1 = hw[1];
        }
        auto hp_at = [&](uint32_t at) -> uint8_t { return gb(hrel + (int32_t)at); };
        // trimmed ends (the '\n' itself is white space; a virtual newline at the end of the text is not a byte)
        int32_t x_h = e_h, x_s = e_s, x_q = e_q;
        if (is_ws_ascii(l_h)) {
            x_h--;
            while (x_h > b_h && is_ws_ascii(gb(x_h - 1))) x_h--;
        }
        if (is_ws_ascii(l_s)) {
            x_s--;
            while (x_s > b_s && is_ws_ascii(gb(x_s - 1))) x_s--;
        }
        if (is_ws_ascii(l_q)) {
            x_q--;
            while (x_q > b_q && is_ws_ascii(gb(x_q - 1))) x_q--;
        }
        // F3's hypothesis (fastq.rs:266-300 on four-line records)
        if (!(f_h == '@' && f_s != '+' && f_p == '+' && x_q > b_q)) bad_rec = true;
        bg_fastq_record_t o = {};
        const uint32_t hn = (uint32_t)(x_h - b_h);  // trimmed header, '@' included ('@' is not white space: >= 1 when f_h == '@')
        const uint32_t n = hn ? hn - 1 : 0;
        {  // fastq.rs:275-277: line[1..].trim_end().splitn(2, ' ')
            uint32_t sp = n;  // (no space: the whole trimmed line is the id)
            auto spaces = [](uint64_t w) {  // bit 7 of every byte that is ' '
                const uint64_t x = w ^ 0x2020202020202020ull;
                return (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
            };
            bool done = false;
            if (h2) {
                uint64_t z = spaces(w0) & (~0ull << (8 * mis));  // bytes in front of the header's second byte do not count
                if (z) {
                    sp = ((uint32_t)(__ffsll((long long)z) - 1) >> 3) - mis;
                    done = true;
                } else if ((z = spaces(w1))) {
                    sp = 8 - mis + ((uint32_t)(__ffsll((long long)z) - 1) >> 3);
                    done = true;
                }
            }
            if (!done) {
                uint32_t at = h2 ? 16 - mis : 0;
                while (at < n && hp_at(at) != ' ') at++;
                sp = at;
            }
            if (sp > n) sp = n;  // a space behind the trimmed end is not part of the line
            o.id_off = t0 + b_h + 1;
            o.id_len = sp;
            if (sp < n) {
                o.desc_off = o.id_off + sp + 1;
                o.desc_len = n - sp - 1;
                o.has_desc = 1;
            }
        }
        o.seq_len = (uint32_t)(x_s - b_s);
        o.qual_len = (uint32_t)(x_q - b_q);
        rr = o;
        lens = (uint64_t)o.seq_len | (uint64_t)o.qual_len << 32;
        FusedRec d;
        d.seq_rel = (fq_rel_t)b_s;
        d.qual_rel = (fq_rel_t)b_q;
        d.seq_n = (fq_len_t)o.seq_len;
        d.qual_n = (fq_len_t)o.qual_len;
        d.seq_dst = d.qual_dst = 0;
        d.flags = o.id_len == 0 ? 1u : 0u;
        s_rec[vtid] = d;
    }
    // ---- offsets: block scan of the lengths in record order, then the tile totals' look-backs
    uint64_t ex, tile_sum;
    {
        uint64_t v = lens;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, o), hh = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), o);
            if ((int)lane >= o) v += (uint64_t)hh << 32 | lo;
        }
        __syncthreads();
        FQ_STAMP(4);  // records parsed
        if (lane == 63) s_w[vw] = v;
        __syncthreads();
        uint64_t wb = 0, all = 0;
        for (uint32_t w = 0; w < 4; w++) {
            if (w < vw) wb += s_w[w];
            all += s_w[w];
        }
        ex = wb + v - lens;
        tile_sum = all;  // (32-bit halves: a tile's lines hold far fewer than 2^32 bytes)
    }
    if (__any(bad_rec) && lane == 0) s_bc[5] = 1;  // (not [4]: a wavefront may still be reading that one above)
    if (vtid < nr) {
        s_rec[vtid].seq_dst = (fq_len_t)ex;
        s_rec[vtid].qual_dst = (fq_len_t)(ex >> 32);
    }
    uint64_t seq_before, qual_before;
    if (spec) {  // all three totals at once; the line count confirms the guess
        const uint64_t agg[3] = {T_all, tile_sum & 0xFFFFFFFFull, tile_sum >> 32};
        uint64_t ex3[3];
        fq_lookback<3>(a.tiles, a.tiles + 3 * (uint64_t)a.n_tiles, a.n_tiles, tile, agg, ex3, s_lb, &a.out[1], vw);
        lines_before = ex3[0];
        seq_before = ex3[1];
        qual_before = ex3[2];
        if (((3u - (uint32_t)(lines_before & 3)) & 3u) != j0) {  // (block-uniform) guessed wrong: what this tile published is wrong too
            irregular = true;
            nr = 0;
        }
    } else {
        const uint64_t agg[2] = {tile_sum & 0xFFFFFFFFull, tile_sum >> 32};
        uint64_t ex2[2];
        fq_lookback<2>(a.tiles + a.n_tiles, a.tiles + 4 * (uint64_t)a.n_tiles, a.n_tiles, tile, agg, ex2, s_lb, &a.out[1], vw);
        seq_before = ex2[0];
        qual_before = ex2[1];
    }
    __syncthreads();
    FQ_STAMP(5);  // look-back #2 done
    const uint64_t rec0 = (lines_before + j0) / 4;  // global index of the tile's first record
    if (vtid < nr) {
        const uint64_t so = seq_before + (ex & 0xFFFFFFFFull), qo = qual_before + (ex >> 32);
        const uint64_t r = rec0 + vtid;
        rr.seq_off = so;
        rr.qual_off = qo;
        if (r < a.rec_cap) {  // (the record itself: after the copy phase, with its check)
            a.seq_off[r] = so;
            a.qual_off[r] = qo;
        }
    }
    if (last_tile && tid == 0) {  // closing offsets and the totals the host reads
        const uint64_t lines = lines_before + T_all, n_rec = lines / 4;
        const uint64_t st = seq_before + (tile_sum & 0xFFFFFFFFull), qt = qual_before + (tile_sum >> 32);
        if (n_rec <= a.rec_cap) {
            a.seq_off[n_rec] = st;
            a.qual_off[n_rec] = qt;
        }
        a.out[0] = lines;
        a.out[2] = st;
        a.out[3] = qt;
    }
    if (irregular && tid == 0) s_bc[5] = 1;
    __syncthreads();
    if ((s_bc[4] | s_bc[5]) && tid == 0) atomicOr((unsigned long long*)&a.out[1], 1ull);
    // ---- copy + Record::check.  The tile's sequence bytes go to ONE contiguous range of `seq` (its records follow each other
    // there), likewise the qualities: every thread takes 16-byte pieces of that range on the DESTINATION's alignment.  Which
    // record a piece begins in comes from a table the records' owners fill (s_prec); its 16 source bytes are five aligned LDS
    // dwords funnel-shifted into four, a piece that runs into the next record merges a second such read under a byte mask, the
    // bytes are classified for Record::check four at a time (SWAR) — straight-line code for all but the first / last piece of the
    // range, pieces with three records, and lines that begin more than 1 KB in front of the tile (those go byte by byte).
    // (Round 6, SQ counters of a build without this phase against the complete one: the phase was 108 M vector + 183 M scalar
    //  instructions of the kernel's 169 + 222 M per call — a binary search and a handful of divergent branches per piece, at
    //  four wavefronts per SIMD the kernel was bound by instruction issue, not by memory.)
    {
#ifdef FQ_KO_COPY
        nr = 0;
#endif
        const uint32_t seq_total = (uint32_t)(tile_sum & 0xFFFFFFFFull), qual_total = (uint32_t)(tile_sum >> 32);
        uint8_t* const seq_base = a.seq + seq_before;
        uint8_t* const qual_base = a.qual + qual_before;
        const uint32_t d0s = (uint32_t)((uintptr_t)seq_base & 15), d0q = (uint32_t)((uintptr_t)qual_base & 15);
        // the owners' tables: piece c (output offsets 16 c - d0 .. + 15, clipped to the range) begins in record i
        // (a tile whose records bring more than the table holds — long lines from in front of it — searches instead)
        constexpr uint32_t kPrecCap = kTile / 16 + 4;
        const bool tab_s = ((d0s + seq_total + 15) >> 4) <= kPrecCap, tab_q = ((d0q + qual_total + 15) >> 4) <= kPrecCap;
        if (vtid < nr) {
            const FusedRec d = s_rec[vtid];
            if (d.seq_n && tab_s) {
                const uint32_t c_lo = d.seq_dst ? (d.seq_dst + d0s + 15) >> 4 : 0, c_hi = (d.seq_dst + d.seq_n + d0s + 15) >> 4;  // [c_lo, c_hi)
                for (uint32_t c = c_lo; c < c_hi; c++) s_prec[0][c] = (uint8_t)vtid;
            }
            if (d.qual_n && tab_q) {
                const uint32_t c_lo = d.qual_dst ? (d.qual_dst + d0q + 15) >> 4 : 0, c_hi = (d.qual_dst + d.qual_n + d0q + 15) >> 4;
                for (uint32_t c = c_lo; c < c_hi; c++) s_prec[1][c] = (uint8_t)vtid;
            }
        }
        __syncthreads();
        FQ_STAMP(6);  // piece tables filled
        auto tile_copy = [&](auto SEQ_T, uint8_t* dst_base, uint32_t total, uint32_t d0) {
            constexpr bool SEQ = decltype(SEQ_T)::value;
            const uint8_t* prec = s_prec[SEQ ? 0 : 1];
            const bool tab = SEQ ? tab_s : tab_q;
            auto rel_of = [&](uint32_t i) { return SEQ ? s_rec[i].seq_rel : s_rec[i].qual_rel; };
            auto dst_of = [&](uint32_t i) { return SEQ ? s_rec[i].seq_dst : s_rec[i].qual_dst; };
            auto end_of = [&](uint32_t i) -> uint32_t { return SEQ ? (uint32_t)s_rec[i].seq_dst + s_rec[i].seq_n : (uint32_t)s_rec[i].qual_dst + s_rec[i].qual_n; };
            auto flag = [&](uint32_t i, bool hi, bool bad) {
                const uint32_t f = (hi ? (SEQ ? 2u : 8u) : 0u) | (bad ? 4u : 0u);
                if (f) atomicOr(&s_rec[i].flags, f);
            };
            // bit 7 of every byte of w that is >= 0x80 / (SEQ) that is neither alphabetic nor one of - . *  (fastq.rs:392-401)
            auto bad_bits = [](uint32_t w) -> uint32_t {
                auto zero = [](uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; };
                const uint32_t t = (w | 0x20202020u) & 0x7F7F7F7Fu;
                const uint32_t letter = (t + 0x1F1F1F1Fu) & ~(t + 0x05050505u) & 0x80808080u;  // 'a' <= t <= 'z' (7-bit bytes)
                return ~(letter | zero(w ^ 0x2D2D2D2Du) | zero(w ^ 0x2E2E2E2Eu) | zero(w ^ 0x2A2A2A2Au)) & 0x80808080u;
            };
            const uint32_t n_pieces = (d0 + total + 15) >> 4;
            for (uint32_t c = tid; c < n_pieces && nr; c += 256) {
                const uint32_t p0 = 16 * c - d0;  // output offset of the piece's byte 0 (in front of the range for c == 0: as int)
                const uint32_t o_lo = 16 * c > d0 ? p0 : 0, o_hi = min(total, p0 + 16);
                uint32_t i;
                if (tab) {
                    i = prec[c];
                } else {  // the first record whose end lies behind the piece's first byte
                    uint32_t lo = 0, hi = nr - 1;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (end_of(mid) > o_lo) hi = mid; else lo = mid + 1;
                    }
                    i = lo;
                }
                const uint32_t e1 = end_of(i);
                const bool two = o_hi > e1;
                uint32_t i2 = two ? i + 1 : i;
                const int32_t relA = rel_of(i) + (int32_t)(p0 - dst_of(i)), relB = rel_of(i2) + (int32_t)(p0 - dst_of(i2));
                const bool whole = o_lo == p0 && o_hi == p0 + 16;
                // fast: a piece of one or two records whose source bytes are in LDS (the second record not empty).  The first and
                // the last piece of the range are partial — bytes of another tile's records lie in front / behind: the same
                // sixteen bytes, classified and stored under a byte mask (they went byte by byte through the general path first:
                // a chain of dependent LDS reads per byte on the two wavefronts every other one then waited for, 7 of the copy
                // phase's 9 us in the round-6 timeline)
                const bool fast = kLdsTile && relA >= lds_lo && relB >= lds_lo && (!two || (i2 < nr && o_hi <= end_of(i2) && end_of(i2) > e1));
                if (fast) {
                    auto lds16 = [&](int32_t rel, uint32_t (&w)[4]) {
                        const uint32_t sh = (uint32_t)rel & 3u;
                        const uint32_t* q4 = (const uint32_t*)(s_tile + (rel - (int32_t)sh));
                        uint32_t r5[5];
#pragma unroll
                        for (int q = 0; q < 5; q++) r5[q] = q4[q];
#pragma unroll
                        for (int q = 0; q < 4; q++) w[q] = __builtin_amdgcn_alignbyte(r5[q + 1], r5[q], sh);
                    };
                    auto below = [](int k) -> uint32_t { return k >= 4 ? 0xFFFFFFFFu : k <= 0 ? 0u : (1u << (8 * k)) - 1u; };  // bytes 0 .. k - 1 of a word
                    uint32_t w[4], w2[4];
                    lds16(relA, w);
                    lds16(relB, w2);  // (== relA's bytes when the piece has one record: merged under an all-ones mask)
                    const uint32_t split = two ? e1 - p0 : 16u;  // bytes [split, 16) belong to record i2
                    const uint32_t first = o_lo - p0, last = o_hi - p0;  // the piece's bytes of this range: [first, last)
                    uint32_t hiA = 0, hiB = 0, badA = 0, badB = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t keep = below((int)split - 4 * q);  // bytes of word q that belong to record i
                        w[q] = (w[q] & keep) | (w2[q] & ~keep);
                        uint32_t hb = w[q] & 0x80808080u, bb = SEQ ? bad_bits(w[q]) : 0u;
                        if (!whole) {
                            const uint32_t v = below((int)last - 4 * q) & ~below((int)first - 4 * q);
                            hb &= v;
                            bb &= v;
                        }
                        hiA |= hb & keep;
                        hiB |= hb & ~keep;
                        badA |= (bb | hb) & keep;   // (a byte >= 0x80 is no letter either)
                        badB |= (bb | hb) & ~keep;
                    }
#if defined(FQ_KO_STORE)
                    asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
#else
                    if (whole) {
                        *(uint4*)(dst_base + o_lo) = make_uint4(w[0], w[1], w[2], w[3]);
                    } else {
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++)
                            if (k >= first && k < last) dst_base[p0 + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));  // (p0 + k: mod 2^32, >= 0 here)
                    }
#endif
                    if (hiA | badA) flag(i, hiA != 0, SEQ && badA != 0);
                    if (hiB | badB) flag(i2, hiB != 0, SEQ && badB != 0);
                } else {  // three records in a piece, a line from more than 1 KB in front of the tile
                    uint32_t ii = i;
                    for (uint32_t o = o_lo; o < o_hi; o++) {
                        while (o >= end_of(ii)) ii++;  // (o < total = the last record's end: ii stays below nr)
                        const uint8_t ch = gb(rel_of(ii) + (int32_t)(o - dst_of(ii)));
                        dst_base[o] = ch;
                        const bool hi = ch >= 0x80;
                        const bool bad = SEQ && !((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z') || ch == '-' || ch == '.' || ch == '*');
                        flag(ii, hi, bad);
                    }
                }
            }
        };
        tile_copy(std::true_type{}, seq_base, seq_total, d0s);
        tile_copy(std::false_type{}, qual_base, qual_total, d0q);
        FQ_STAMP(7);  // this wavefront's pieces issued
        __syncthreads();
        FQ_STAMP(8);  // all copied (the barrier waits for the stores)
        if (vtid < nr) {
            const uint64_t r = rec0 + vtid;
            const uint32_t f = s_rec[vtid].flags;
            rr.check = (f & 1u)   ? BG_FQCHECK_EMPTY_ID
                       : (f & 2u) ? BG_FQCHECK_NONASCII_SEQ
                       : (f & 4u) ? BG_FQCHECK_INVALID_SEQ
                       : (f & 8u) ? BG_FQCHECK_NONASCII_QUAL
                       : rr.seq_len != rr.qual_len ? BG_FQCHECK_UNEQUAL
                                                   : BG_FQCHECK_OK;
            if (r < a.rec_cap) a.recs[r] = rr;
        }
        FQ_STAMP(9);
    }
}

int scan_lengths(const uint32_t* d_len, uint64_t n, uint64_t* d_off, uint64_t* d_sums, hipStream_t st) {
    const uint32_t nb = (uint32_t)(n / 2048 + 1);  // one more block than items need: it writes the closing offset
    fq_block_sums_kernel<<<dim3(nb), dim3(256), 0, st>>>(d_len, n, d_sums);
    fq_scan_small_kernel<uint64_t><<<dim3(1), dim3(1024), 0, st>>>(d_sums, d_sums + nb, nb, nullptr);
    fq_scan_apply_kernel<<<dim3(nb), dim3(256), 0, st>>>(d_len, n, d_sums + nb, d_off);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

// ---- CIGAR ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t put_num(char* o, uint32_t v) {
    char tmp[10];
    uint32_t n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (uint32_t i = 0; i < n; i++) o[i] = tmp[n - 1 - i];
    return n;
}
// one thread per alignment; out slot of `stride` chars per alignment, len[p] = chars written or a negative status
__global__ __launch_bounds__(256) void cigar_kernel(const bg_alignment_t* __restrict__ aln, const uint8_t* __restrict__ ops, uint64_t n, int hard_clip,
                                                    char* __restrict__ out, uint64_t stride, int32_t* __restrict__ len) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const bg_alignment_t a = aln[p];
    if (a.mode == BG_MODE_CUSTOM) {  // bio-types: "Cigar fn not supported for custom alignment mode" (panic)
        len[p] = BG_ERR_UNSUPPORTED;
        return;
    }
    char* o = out + p * stride;
    uint64_t w = 0;
    const char clip = hard_clip ? 'H' : 'S';
    bool overflow = false;
    auto emit = [&](uint32_t k, char c) {
        if (w + 11 > stride) {
            overflow = true;
            return;
        }
        w += put_num(o + w, k);
        o[w++] = c;
    };
    auto add = [&](uint32_t kind, uint32_t k) {
        if (kind == BG_OP_MATCH) emit(k, '=');
        else if (kind == BG_OP_SUBST) emit(k, 'X');
        else if (kind == BG_OP_DEL) emit(k, 'D');
        else if (kind == BG_OP_INS) emit(k, 'I');
    };
    if (a.n_ops) {
        const uint8_t* q = ops + a.ops_off;
        uint32_t last = q[0], k = 1;
        if (a.xstart > 0) emit(a.xstart, clip);
        for (uint32_t i = 1; i < a.n_ops; i++) {
            const uint32_t op = q[i];
            if (op == last) {
                k++;
            } else {
                add(last, k);
                k = 1;
            }
            last = op;
        }
        add(last, k);
        if (a.xlen > a.xend) emit(a.xlen - a.xend, clip);
    }
    len[p] = overflow ? BG_ERR_OPS_CAP : (int32_t)w;
}

// Alignment::pretty(x, y, ncol) (bio-types): three rows — x, the operation marks ('|' match, '\\' mismatch, '+' insertion,
// 'x' deletion, ' ' clipped), y — cut into blocks of ncol columns, every block "x row\n marks\n y row\n\n\n".  The standard
// modes print the clipped prefixes / suffixes of x and y around the operations, AlignmentMode::Custom walks its
// Xclip / Yclip operations instead (the crate prints the FIRST len symbols of the sequence for a clip operation,
// wherever the clip sits: reproduced).  One thread per alignment; two passes over the operations (length, then text).
__global__ __launch_bounds__(256) void pretty_kernel(const bg_alignment_t* __restrict__ aln, const uint8_t* __restrict__ ops, uint64_t n,
                                                     const uint8_t* __restrict__ xs, const uint64_t* __restrict__ x_off,
                                                     const uint8_t* __restrict__ ys, const uint64_t* __restrict__ y_off, uint32_t ncol,
                                                     char* __restrict__ out, uint64_t stride, int64_t* __restrict__ len) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const bg_alignment_t a = aln[p];
    const uint8_t* x = xs + x_off[p];
    const uint8_t* y = ys + y_off[p];
    const uint64_t xl = x_off[p + 1] - x_off[p], yl = y_off[p + 1] - y_off[p];
    char* o = out + p * stride;
    bool bad = xl != a.xlen || yl != a.ylen;  // not the sequences this alignment was computed from
    uint64_t ml = 0;
    for (int pass = 0; pass < 2 && !bad; pass++) {
        uint64_t col = 0;
        auto put = [&](uint8_t cx, char ci, uint8_t cy) {
            if (pass == 1) {
                const uint64_t blk = col / ncol, w = col - blk * ncol;
                const uint64_t bl = min((uint64_t)ncol, ml - blk * ncol);
                char* b = o + blk * (3ull * ncol + 5);
                b[w] = (char)cx;
                b[bl + 1 + w] = ci;
                b[2 * (bl + 1) + w] = (char)cy;
            }
            bad = bad || cx >= 0x80 || cy >= 0x80;  // from_utf8_lossy widens such a byte: the crate's length assert fires
            col++;
        };
        if (a.n_ops) {
            uint64_t xi = 0, yi = 0;
            const uint8_t* q = ops + a.ops_off;
            uint32_t clip = 0;
            if (a.mode != BG_MODE_CUSTOM) {
                xi = a.xstart;
                yi = a.ystart;
                for (uint64_t k = 0; k < a.xstart && k < xl; k++) put(x[k], ' ', ' ');
                for (uint64_t k = 0; k < a.ystart && k < yl; k++) put(' ', ' ', y[k]);
            }
            for (uint32_t i = 0; i < a.n_ops && !bad; i++) {
                const uint32_t op = q[i];
                if (op == BG_OP_MATCH || op == BG_OP_SUBST) {
                    if (xi >= xl || yi >= yl) { bad = true; break; }
                    put(x[xi++], op == BG_OP_MATCH ? '|' : '\\', y[yi++]);
                } else if (op == BG_OP_DEL) {
                    if (yi >= yl) { bad = true; break; }
                    put('-', 'x', y[yi++]);
                } else if (op == BG_OP_INS) {
                    if (xi >= xl) { bad = true; break; }
                    put(x[xi++], '+', '-');
                } else {
                    const uint32_t cl = clip < 4 ? a.clip_len[clip] : 0;
                    clip++;
                    if (op == BG_OP_XCLIP) {
                        for (uint64_t k = 0; k < cl && k < xl; k++, xi++) put(x[k], ' ', ' ');
                    } else {
                        for (uint64_t k = 0; k < cl && k < yl; k++, yi++) put(' ', ' ', y[k]);
                    }
                }
            }
            if (a.mode != BG_MODE_CUSTOM) {
                for (uint64_t k = xi; k < xl; k++) put(x[k], ' ', ' ');
                for (uint64_t k = yi; k < yl; k++) put(' ', ' ', y[k]);
            }
        }
        if (pass == 0) {
            ml = col;
            const uint64_t nb = (ml + ncol - 1) / ncol;
            if (3 * ml + 5 * nb > stride) {
                len[p] = BG_ERR_OPS_CAP;
                return;
            }
        }
    }
    if (bad) {
        len[p] = BG_ERR_UNSUPPORTED;
        return;
    }
    const uint64_t nb = (ml + ncol - 1) / ncol;
    for (uint64_t blk = 0; blk < nb; blk++) {
        const uint64_t bl = min((uint64_t)ncol, ml - blk * ncol);
        char* b = o + blk * (3ull * ncol + 5);
        b[bl] = b[2 * bl + 1] = b[3 * bl + 2] = b[3 * bl + 3] = b[3 * bl + 4] = '\n';
    }
    len[p] = (int64_t)(3 * ml + 5 * nb);
}

}  // namespace

// shared with seed_extend.hip: exclusive scan of n uint32 counts into n + 1 uint64 offsets; d_sums holds 2 * (n / 2048 + 1) words
int bg_scan_u32(const uint32_t* d_len, uint64_t n, uint64_t* d_off, uint64_t* d_sums, hipStream_t st) {
    return scan_lengths(d_len, n, d_off, d_sums, st);
}

extern "C" int bg_fastq_parse_dev(bg_ctx* ctx, const uint8_t* d_text, uint64_t len, bg_fastq_record_t* d_recs, uint64_t rec_cap, uint8_t* d_seq,
                                  uint64_t* d_seq_off, uint8_t* d_qual, uint64_t* d_qual_off, uint64_t* n_records, int32_t* status,
                                  uint64_t* err_pos, void* stream) {
    if (!ctx || !n_records || !status || !err_pos) return BG_ERR_INVALID_ARG;
    *n_records = 0;
    *status = BG_FASTQ_OK;
    *err_pos = 0;
    hipStream_t st = (hipStream_t)stream;
    bg_scratch_guard guard(ctx, st);
    BG_HIP(hipSetDevice(ctx->device));
    if (len == 0) {
        const uint64_t z = 0;
        if (d_seq_off) BG_HIP(hipMemcpyAsync(d_seq_off, &z, 8, hipMemcpyHostToDevice, st));
        if (d_qual_off) BG_HIP(hipMemcpyAsync(d_qual_off, &z, 8, hipMemcpyHostToDevice, st));
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    }
    if (!d_text || !d_recs || !d_seq || !d_seq_off || !d_qual || !d_qual_off) return BG_ERR_INVALID_ARG;
    int rc;
    // FF: the one-pass reader (four-line ASCII records: nearly every FASTQ); anything else raises its flag and takes F1 .. F6
    if (!ctx->fq_no_fused) {
        const uint64_t n_tiles = (len + kTile - 1) / kTile;
        if (n_tiles < (1ull << 31)) {
            const size_t words = 6 * n_tiles + 8;
            if ((rc = bg_reserve(&ctx->aux, &ctx->aux_bytes, words * 8))) return rc;
            BG_HIP(hipMemsetAsync(ctx->aux, 0, words * 8, st));
            FusedArgs fa = {};
            fa.t = d_text;
            fa.len = len;
            fa.tiles = (uint64_t*)ctx->aux;
            fa.out = fa.tiles + 6 * n_tiles;
            fa.recs = d_recs;
            fa.rec_cap = rec_cap;
            fa.seq = d_seq;
            fa.qual = d_qual;
            fa.seq_off = d_seq_off;
            fa.qual_off = d_qual_off;
            fa.n_tiles = (uint32_t)n_tiles;
            fq_fused_kernel<<<dim3((uint32_t)n_tiles), dim3(256), 0, st>>>(fa);
            BG_HIP(hipGetLastError());
            uint64_t res[4] = {0, 1, 0, 0};
            BG_HIP(hipMemcpyAsync(res, fa.out, sizeof(res), hipMemcpyDeviceToHost, st));
            BG_HIP(hipStreamSynchronize(st));  // the call's one host round trip
            if (res[1] == 0 && (res[0] & 3) == 0) {
                *n_records = res[0] / 4;
                return *n_records > rec_cap ? BG_ERR_TOO_LARGE : BG_OK;
            }
        }
    }
    // F1: newline counts per chunk, their scan (+ total), line starts
    const uint64_t nchunks = (len + kChunk - 1) / kChunk;
    const size_t n_part = 2 * (nchunks / 2048 + 2);  // partial sums of the two-level scan
    const size_t head = nchunks * 4 + 16 + (nchunks + 2 + n_part) * 8 + nchunks + 128;
    if ((rc = bg_reserve(&ctx->aux, &ctx->aux_bytes, head))) return rc;
    uint32_t* d_cnt = (uint32_t*)ctx->aux;
    uint64_t* d_base = (uint64_t*)((uint8_t*)ctx->aux + ((nchunks * 4 + 15) & ~(size_t)15));
    uint64_t* d_total = d_base + nchunks;  // (the scan's closing offset)
    uint64_t* d_part = d_total + 2;
    uint8_t* d_hi = (uint8_t*)(d_part + n_part);
    fq_count_newlines_kernel<<<dim3((uint32_t)nchunks), dim3(256), 0, st>>>(d_text, len, d_cnt, d_hi);
    // (a single block scanning the ~79 000 chunk counts of a 323 MB text took 148 us; block sums + their scan + apply: 20)
    if ((rc = scan_lengths(d_cnt, nchunks, d_base, d_part, st))) return rc;
    BG_HIP(hipGetLastError());
    uint64_t n_nl = 0;
    uint8_t last_byte = 0;
    BG_HIP(hipMemcpyAsync(&n_nl, d_total, 8, hipMemcpyDeviceToHost, st));
    BG_HIP(hipMemcpyAsync(&last_byte, d_text + len - 1, 1, hipMemcpyDeviceToHost, st));
    BG_HIP(hipStreamSynchronize(st));
    const uint64_t n_lines = n_nl + (last_byte != '\n' ? 1 : 0);
    const uint64_t max_rec = n_lines / 2 + 1;  // a record has at least a header and a quality line... be generous
    // scratch: line starts, line info, record lines, lengths, scan partials, walker output, first_bad
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_ls = take((n_lines + 2) * 8), o_info = take((n_lines + 1) * sizeof(LineInfo)), o_rl = take(max_rec * sizeof(RecLines)),
                 o_sl = take(max_rec * 4), o_ql = take(max_rec * 4), o_sum = take((max_rec / 2048 + 2) * 2 * 8), o_misc = take(64);
    if ((rc = bg_reserve(&ctx->tb, &ctx->tb_bytes, off))) return rc;
    uint8_t* S = (uint8_t*)ctx->tb;
    uint64_t* d_ls = (uint64_t*)(S + o_ls);
    LineInfo* d_info = (LineInfo*)(S + o_info);
    RecLines* d_rl = (RecLines*)(S + o_rl);
    uint32_t *d_sl = (uint32_t*)(S + o_sl), *d_ql = (uint32_t*)(S + o_ql);
    uint64_t* d_sum = (uint64_t*)(S + o_sum);
    uint64_t* d_misc = (uint64_t*)(S + o_misc);  // [0] first_bad, [1..3] walker output
    fq_line_starts_kernel<<<dim3((uint32_t)nchunks), dim3(256), 0, st>>>(d_text, len, d_base, d_ls);
    BG_HIP(hipMemcpyAsync(d_ls + n_lines, &len, 8, hipMemcpyHostToDevice, st));  // closing offset (a no-op rewrite if the text ends in '\n')
    // F2
    fq_line_info_kernel<<<dim3((uint32_t)((n_lines + 255) / 256)), dim3(256), 0, st>>>(d_text, d_ls, n_lines, d_hi, d_info);
    // F3
    const uint64_t n_rec4 = n_lines / 4;
    const uint64_t none = ~0ull;
    BG_HIP(hipMemcpyAsync(d_misc, &none, 8, hipMemcpyHostToDevice, st));
    if (n_rec4) fq_four_line_kernel<<<dim3((uint32_t)((n_rec4 + 255) / 256)), dim3(256), 0, st>>>(d_info, n_rec4, (unsigned long long*)d_misc);
    BG_HIP(hipGetLastError());
    uint64_t first_bad = 0;
    BG_HIP(hipMemcpyAsync(&first_bad, d_misc, 8, hipMemcpyDeviceToHost, st));
    BG_HIP(hipStreamSynchronize(st));
    uint64_t n_fast = std::min(first_bad, n_rec4);
    if (n_fast) fq_four_line_records_kernel<<<dim3((uint32_t)((n_fast + 255) / 256)), dim3(256), 0, st>>>(n_fast, d_rl);
    uint64_t n_rec = n_fast;
    // F4: whatever follows the four-line prefix
    if (4 * n_fast < n_lines) {
        fq_walk_kernel<<<dim3(1), dim3(64), 0, st>>>(d_info, n_lines, 4 * n_fast, d_rl, n_fast, max_rec, d_misc + 1);
        BG_HIP(hipGetLastError());
        uint64_t w[3];
        BG_HIP(hipMemcpyAsync(w, d_misc + 1, 24, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        n_rec += w[0];
        *status = (int32_t)w[1];
        if (*status != BG_FASTQ_OK) BG_HIP(hipMemcpy(err_pos, d_ls + w[2], 8, hipMemcpyDeviceToHost));
    }
    *n_records = n_rec;
    if (n_rec > rec_cap) return BG_ERR_TOO_LARGE;
    if (n_rec == 0) {
        const uint64_t z = 0;
        BG_HIP(hipMemcpyAsync(d_seq_off, &z, 8, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_qual_off, &z, 8, hipMemcpyHostToDevice, st));
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    }
    // F5, F6
    fq_measure_kernel<<<dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st>>>(d_text, d_ls, d_info, n_lines, d_rl, n_rec, d_recs, d_sl, d_ql);
    if ((rc = scan_lengths(d_sl, n_rec, d_seq_off, d_sum, st))) return rc;
    if ((rc = scan_lengths(d_ql, n_rec, d_qual_off, d_sum, st))) return rc;
    if (len / n_lines <= 256)
        fq_gather_kernel<16><<<dim3((uint32_t)((n_rec + 15) / 16)), dim3(256), 0, st>>>(d_text, len, d_ls, d_info, n_lines, d_rl, n_rec, d_recs, d_seq_off,
                                                                                          d_qual_off, d_seq, d_qual);
    else
        fq_gather_kernel<64><<<dim3((uint32_t)((n_rec + 3) / 4)), dim3(256), 0, st>>>(d_text, len, d_ls, d_info, n_lines, d_rl, n_rec, d_recs, d_seq_off,
                                                                                        d_qual_off, d_seq, d_qual);
    BG_HIP(hipGetLastError());
    BG_HIP(hipStreamSynchronize(st));
    return BG_OK;
}

extern "C" int bg_fastq_parse(bg_ctx* ctx, const uint8_t* text, uint64_t len, bg_fastq_record_t* recs, uint64_t rec_cap, uint8_t* seq,
                              uint64_t* seq_off, uint8_t* qual, uint64_t* qual_off, uint64_t* n_records, int32_t* status, uint64_t* err_pos) {
    if (!ctx || !n_records || !status || !err_pos) return BG_ERR_INVALID_ARG;
    if (len && (!text || !recs || !seq || !seq_off || !qual || !qual_off)) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    int rc;
    const size_t need[6] = {std::max<uint64_t>(len, 16), std::max<uint64_t>(len, 16), (rec_cap + 1) * 8, (rec_cap + 1) * 8,
                            std::max<uint64_t>(rec_cap, 1) * sizeof(bg_fastq_record_t), std::max<uint64_t>(len, 16)};
    for (int i = 0; i < 6; i++)
        if ((rc = bg_reserve(&ctx->io[i], &ctx->io_cap[i], need[i]))) return rc;
    uint8_t *d_text = (uint8_t*)ctx->io[0], *d_seq = (uint8_t*)ctx->io[1], *d_qual = (uint8_t*)ctx->io[5];
    uint64_t *d_so = (uint64_t*)ctx->io[2], *d_qo = (uint64_t*)ctx->io[3];
    bg_fastq_record_t* d_recs = (bg_fastq_record_t*)ctx->io[4];
    hipStream_t st = ctx->stream;
    if (len) BG_HIP(hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, st));
    rc = bg_fastq_parse_dev(ctx, d_text, len, d_recs, rec_cap, d_seq, d_so, d_qual, d_qo, n_records, status, err_pos, st);
    if (rc) return rc;
    const uint64_t n = *n_records;
    if (seq_off) BG_HIP(hipMemcpyAsync(seq_off, d_so, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    if (qual_off) BG_HIP(hipMemcpyAsync(qual_off, d_qo, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    if (n) BG_HIP(hipMemcpyAsync(recs, d_recs, n * sizeof(bg_fastq_record_t), hipMemcpyDeviceToHost, st));
    BG_HIP(hipStreamSynchronize(st));
    if (n) {
        BG_HIP(hipMemcpyAsync(seq, d_seq, seq_off[n], hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(qual, d_qual, qual_off[n], hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
    }
    return BG_OK;
}

extern "C" int bg_cigar_batch_dev(bg_ctx* ctx, uint64_t n, const bg_alignment_t* d_aln, const uint8_t* d_ops, int hard_clip, char* d_out,
                                  uint64_t stride, int32_t* d_len, void* stream) {
    if (!ctx) return BG_ERR_INVALID_ARG;
    if (n == 0) return BG_OK;
    if (!d_aln || !d_out || !d_len || stride < 24) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    cigar_kernel<<<dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(d_aln, d_ops, n, hard_clip, d_out, stride, d_len);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

extern "C" int bg_pretty_batch(bg_ctx* ctx, uint64_t n, const bg_alignment_t* aln, const uint8_t* ops, uint64_t ops_bytes, const uint8_t* x,
                               const uint64_t* x_off, const uint8_t* y, const uint64_t* y_off, uint32_t ncol, char* out, uint64_t out_cap,
                               uint64_t* out_off) {
    if (!ctx || !out_off || ncol == 0) return BG_ERR_INVALID_ARG;
    out_off[0] = 0;
    if (n == 0) return BG_OK;
    if (!aln || (!ops && ops_bytes) || !x_off || !y_off || !out) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    uint64_t max_ml = 0;
    for (uint64_t p = 0; p < n; p++) {
        if (aln[p].n_ops && aln[p].ops_off + aln[p].n_ops > ops_bytes) return BG_ERR_INVALID_ARG;
        max_ml = std::max<uint64_t>(max_ml, (x_off[p + 1] - x_off[p]) + (y_off[p + 1] - y_off[p]));
    }
    const uint64_t stride = (3 * max_ml + 5 * ((max_ml + ncol - 1) / ncol) + 15) & ~15ull;
    const uint64_t xb = x_off[n], yb = y_off[n];
    void* d[8] = {};
    const size_t need[8] = {n * sizeof(bg_alignment_t), std::max<uint64_t>(ops_bytes, 16), std::max<uint64_t>(xb, 16), (n + 1) * 8,
                            std::max<uint64_t>(yb, 16), (n + 1) * 8, std::max<uint64_t>(n * stride, 16), n * 8};
    const void* src[6] = {aln, ops, x, x_off, y, y_off};
    const size_t src_bytes[6] = {need[0], (size_t)ops_bytes, (size_t)xb, need[3], (size_t)yb, need[5]};
    std::vector<char> h;
    std::vector<int64_t> hl(n);
    auto run = [&]() -> int {
        hipStream_t st = ctx->stream;
        for (int i = 0; i < 8; i++) BG_HIP(hipMalloc(&d[i], need[i]));
        for (int i = 0; i < 6; i++)
            if (src_bytes[i]) BG_HIP(hipMemcpyAsync(d[i], src[i], src_bytes[i], hipMemcpyHostToDevice, st));
        pretty_kernel<<<dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st>>>((const bg_alignment_t*)d[0], (const uint8_t*)d[1], n,
                                                                              (const uint8_t*)d[2], (const uint64_t*)d[3], (const uint8_t*)d[4],
                                                                              (const uint64_t*)d[5], ncol, (char*)d[6], stride, (int64_t*)d[7]);
        BG_HIP(hipGetLastError());
        h.resize((size_t)(n * stride));
        BG_HIP(hipMemcpyAsync(h.data(), d[6], n * stride, hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(hl.data(), d[7], n * 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    };
    int rc = run();
    for (void* q : d) hipFree(q);
    if (rc) return rc;
    uint64_t used = 0;
    int status = BG_OK;
    for (uint64_t p = 0; p < n; p++) {
        if (hl[p] < 0) {
            status = (int)hl[p];
            out_off[p + 1] = used;
            continue;
        }
        if (used + (uint64_t)hl[p] > out_cap) return BG_ERR_OPS_CAP;
        memcpy(out + used, h.data() + p * stride, (size_t)hl[p]);
        used += (uint64_t)hl[p];
        out_off[p + 1] = used;
    }
    return status;
}

extern "C" int bg_cigar_batch(bg_ctx* ctx, uint64_t n, const bg_alignment_t* aln, const uint8_t* ops, uint64_t ops_bytes, int hard_clip, char* out,
                              uint64_t out_cap, uint64_t* out_off) {
    if (!ctx || !out_off) return BG_ERR_INVALID_ARG;
    out_off[0] = 0;
    if (n == 0) return BG_OK;
    if (!aln || (!ops && ops_bytes) || !out) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    uint32_t max_ops = 0;
    for (uint64_t p = 0; p < n; p++) {
        if (aln[p].n_ops && aln[p].ops_off + aln[p].n_ops > ops_bytes) return BG_ERR_INVALID_ARG;
        max_ops = std::max(max_ops, aln[p].n_ops);
    }
    const uint64_t stride = ((uint64_t)max_ops * 2 + 24 + 11 + 15) & ~15ull;  // every run is at least "1=": two chars per op, + two clips
    int rc;
    const size_t need[4] = {n * sizeof(bg_alignment_t), std::max<uint64_t>(ops_bytes, 16), n * stride, n * 4};
    for (int i = 0; i < 4; i++)
        if ((rc = bg_reserve(&ctx->io[i], &ctx->io_cap[i], need[i]))) return rc;
    hipStream_t st = ctx->stream;
    BG_HIP(hipMemcpyAsync(ctx->io[0], aln, need[0], hipMemcpyHostToDevice, st));
    if (ops_bytes) BG_HIP(hipMemcpyAsync(ctx->io[1], ops, ops_bytes, hipMemcpyHostToDevice, st));
    rc = bg_cigar_batch_dev(ctx, n, (const bg_alignment_t*)ctx->io[0], (const uint8_t*)ctx->io[1], hard_clip, (char*)ctx->io[2], stride,
                            (int32_t*)ctx->io[3], st);
    if (rc) return rc;
    std::vector<char> h((size_t)(n * stride));
    std::vector<int32_t> hl(n);
    BG_HIP(hipMemcpyAsync(h.data(), ctx->io[2], n * stride, hipMemcpyDeviceToHost, st));
    BG_HIP(hipMemcpyAsync(hl.data(), ctx->io[3], n * 4, hipMemcpyDeviceToHost, st));
    BG_HIP(hipStreamSynchronize(st));
    uint64_t used = 0;
    int status = BG_OK;
    for (uint64_t p = 0; p < n; p++) {
        if (hl[p] < 0) {
            status = hl[p];  // BG_ERR_UNSUPPORTED: AlignmentMode::Custom (the reference panics)
            out_off[p + 1] = used;
            continue;
        }
        if (used + (uint64_t)hl[p] > out_cap) return BG_ERR_OPS_CAP;
        memcpy(out + used, h.data() + p * stride, (size_t)hl[p]);
        used += (uint64_t)hl[p];
        out_off[p + 1] = used;
    }
    return status;
}
