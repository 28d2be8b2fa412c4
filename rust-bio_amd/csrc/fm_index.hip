// FM-index backward search on gfx950 (kernel K5) + the device index builder.
//
// Replaces, for a batch of patterns, FMIndexable::backward_search
// (/root/reference/src/data_structures/fmindex.rs:144-208) with Occ::get
// (/root/reference/src/data_structures/bwt.rs:129-182) and Less[a] (fmindex.rs:228-230).
//
// Device layout (ours; results equal the byte-table definition for every (r, a)):
//   * the BWT is re-coded to 2 bits/symbol: the four most frequent byte values get codes
//     0..3; every other byte value that occurs (the sentinel `$`, stray N, ...) is an
//     *exception*: stored as code 0 in the packed stream and listed (sorted) on the side;
//   * one 64-byte block per 192 symbols, self-contained for a rank query:
//       uint32 cnt[4]   occurrences of code c in bwt[0 .. 192*b)      (16 B)
//       uint32 sym[12]  192 symbols, 16 per word, symbol s at bits [2s, 2s+2)   (48 B)
//     so Occ::get(r, a) = cnt[code(a)] + popcount(matches in the first r%192+1 symbols)
//     costs exactly one 64-B line;
//   * a quad of 4 lanes serves one query: lane t of the quad loads bytes [16t, 16t+16) of
//     the block (one coalesced 64-B request per rank), counts its share, and the quad
//     reduces with two DPP adds.  rank(l-1) and rank(r) are issued together; when both fall
//     in the same block (the common case once the interval is narrow) the line is loaded once;
//   * symbol classes, Less[] and the exception list are staged in LDS;
//   * general alphabets (bwt.rs:94-182 works for any): when more than 1024 BWT positions hold a byte outside the
//     four most frequent ones (a genome with runs of N, a protein text), only the THREE most frequent bytes keep
//     2-bit codes (1..3; code 0 = "something else"), the rare bytes stay sorted lists (<= 1024 positions in all),
//     and every other byte gets a one-hot rank bit vector of its own — 64-byte blocks of a 32-bit counter + 480
//     bits — so that Occ::get is still exactly one 64-byte line, whatever the symbol.  n / 7.5 bytes per such
//     symbol (a 20-letter protein text: 2.7 bytes per symbol of index); the raw BWT is kept for K6 (bwt[pos]).
// No MFMA: this is a latency/bandwidth-bound table walk (DESIGN.md §FM roofline).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <array>
#include <numeric>
#include <thread>

#include "fm_kernels.h"

using namespace bgfm;

namespace {

// `jump` (optional): the state of the search after the LAST kJumpK symbols of a pattern, for every
// kJumpK-mer over the four coded symbols — {l, r, depth}: depth < kJumpK means the search ends there
// (the next symbol empties the interval).  A pattern whose last kJumpK symbols are all coded starts from
// that entry: one table read instead of kJumpK LF steps (2 block reads each).  The table is filled by
// this very kernel (run without it: JUMP == false, which also keeps the two apart in profiles), so the
// results cannot differ.
// SEEDS: the patterns are the seed windows of a batch of reads (seed-and-extend, seed_extend.hip) — query q is
// seed q % S of read q / S: pat[pat_off[r] + k * stride ..+ seed_len) while it fits in the read (else an empty
// pattern: Absent), so overlapping windows need no copy of the reads.
// PACKED: `pat` is a 2-bit stream (16 symbols per little-endian dword, symbol s in bits 2 (s % 16) of dword s / 16, the
// codes being the index's own: bg_fm_pattern_codes / bg_pack2_dev) and `pat_off` counts SYMBOLS: a pattern costs a dword
// load every 16 steps instead of a byte load per step, its symbols are codes already (no class lookup, none of the
// sparse / dense / panic arms).  Only for indexes whose four codes are all symbols (DNA-like BWTs).
// COUNT: the block loads of the launch are counted (bench.py: requested lines against the gather ceiling of
// tools/microbench/ub_gather64.hip); the results are the same.
// (the DEFER launch reads every tag once — n_q bytes, ~10 us per 10 M queries — and the fast kernel writes every tag it
//  owns, deferred or answered, so a caller's stale tag buffer cannot fake a deferral)
// DEFER: only the queries whose tag is kTagDeferred are searched (second launch behind fm_search_fast_kernel)
template <bool JUMP, bool SEEDS, bool PACKED = false, bool COUNT = false, bool DEFER = false>
__global__ __launch_bounds__(256) void fm_backward_search_kernel(
    FmDev fm, uint64_t n_q, const uint8_t* __restrict__ pat, const uint64_t* __restrict__ pat_off,
    uint8_t* __restrict__ tag, uint64_t* __restrict__ lower, uint64_t* __restrict__ upper,
    uint32_t* __restrict__ matched_len, const uint4* __restrict__ jump, const SeedSrc seeds) {
    static_assert(!(PACKED && JUMP), "packed patterns: no jump table");
    __shared__ uint16_t s_class[256];
    __shared__ uint32_t s_less[256];
    __shared__ uint32_t s_exc[kMaxExcLds];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        s_class[i] = fm.sym_class[i];
        // PACKED: entry c (< 4) is less[] of the byte that code c stands for
        s_less[i] = (PACKED && i < 4) ? fm.less[(seeds.code_bytes >> (8 * i)) & 0xFFu] : fm.less[i];
    }
    for (uint32_t i = threadIdx.x; i < fm.n_exc; i += blockDim.x) s_exc[i] = fm.exc_pos[i];  // n_exc <= kMaxExcLds
    __syncthreads();
    const uint32_t* __restrict__ pk = (const uint32_t*)pat;
    uint32_t pk_cur = 0;      // PACKED: the dword that holds the next symbol
    uint32_t n_lines = 0;     // COUNT

    const uint32_t t = threadIdx.x & 3;
    const uint64_t n_quads = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);

    // per-query state (uniform inside a quad)
    bool active = false;
    uint64_t off = 0;
    uint32_t len = 0, pos = 0, l = 0, r = 0, matched = 0;
    uint32_t a_next = 0;

    auto emit = [&](uint32_t tg, uint32_t lo, uint32_t hi, uint32_t ml) {
        if (t == 0) {
            tag[q] = (uint8_t)tg;
            lower[q] = lo;
            upper[q] = hi;
            matched_len[q] = ml;
        }
    };
    // load the next non-empty query into the state; empty patterns are Absent at once
    // (matched_len == 0, fmindex.rs:185-207)
    auto fetch = [&]() {
        active = false;
        while (q < n_q) {
            if (DEFER && tag[q] != kTagDeferred) {
                q += n_quads;
                continue;
            }
            if (SEEDS) {
                const uint64_t rd = q / seeds.S;
                const uint32_t k = (uint32_t)(q - rd * seeds.S);
                const uint64_t o = pat_off[rd];
                off = o + (uint64_t)k * seeds.stride;
                len = (uint64_t)k * seeds.stride + seeds.seed_len <= pat_off[rd + 1] - o ? seeds.seed_len : 0u;
            } else {
                off = pat_off[q];
                len = (uint32_t)(pat_off[q + 1] - off);
            }
            if (len) {
                pos = len;
                l = 0;
                r = fm.n - 1;  // fmindex.rs:148
                matched = 0;
                if (JUMP && len >= kJumpK) {
                    uint32_t idx = 0;
                    bool coded = true;
                    for (uint32_t u = 0; u < kJumpK; u++) {  // u-th symbol from the end
                        const uint32_t c = s_class[pat[off + len - 1 - u]];
                        coded = coded && c < 4;
                        idx = idx << 2 | (c & 3u);
                    }
                    if (coded) {
                        const uint4 e = jump[idx];
                        l = e.x;
                        r = e.y;
                        matched = e.z;
                        pos = len - e.z;
                        if (e.z < kJumpK) {  // the search ends inside the last kJumpK symbols
                            if (e.z)
                                emit(BG_FM_PARTIAL, l, r + 1, e.z);
                            else
                                emit(BG_FM_ABSENT, 0, 0, 0);
                            q += n_quads;
                            continue;
                        }
                        if (pos == 0) {
                            emit(BG_FM_COMPLETE, l, r + 1, matched);
                            q += n_quads;
                            continue;
                        }
                    }
                }
                if (PACKED) {
                    const uint64_t gs = off + pos - 1;
                    pk_cur = pk[gs >> 4];
                    a_next = (pk_cur >> (2 * ((uint32_t)gs & 15u))) & 3u;
                } else {
                    a_next = pat[off + pos - 1];
                }
                active = true;
                return;
            }
            emit(BG_FM_ABSENT, 0, 0, 0);
            q += n_quads;
        }
    };
    fetch();

    while (__any(active)) {
        if (active) {
            // one iteration of the loop at fmindex.rs:160-182
            const uint32_t a = a_next;
            pos -= 1;
            if (PACKED) {
                if (pos) {  // the next symbol: a new dword every 16 steps
                    const uint64_t gs = off + pos - 1;
                    if (((uint32_t)gs & 15u) == 15u) pk_cur = pk[gs >> 4];
                    a_next = (pk_cur >> (2 * ((uint32_t)gs & 15u))) & 3u;
                }
            } else if (pos) {
                a_next = pat[off + pos - 1];  // prefetch; address independent of the ranks
            }
            const uint32_t cls = PACKED ? a : (uint32_t)s_class[a];
            const uint32_t less_a = s_less[a];
            uint32_t occ_r = 0, occ_l = 0;
            bool stop = false;
            uint32_t stop_tag = BG_FM_PARTIAL;
            if (!PACKED && cls == kClsPanic) {
                stop = true;
                stop_tag = BG_FM_PANIC;
            } else if (PACKED || cls < 4) {
                const uint32_t br = r / kSymPerBlock, orr = r - br * kSymPerBlock;
                const uint4 vr = fm.blocks[(uint64_t)br * 4 + t];
                uint4 vl = vr;
                uint32_t ol = 0;
                if (l > 0) {
                    const uint32_t bl = (l - 1) / kSymPerBlock;
                    ol = (l - 1) - bl * kSymPerBlock;
                    if (bl != br) {
                        vl = fm.blocks[(uint64_t)bl * 4 + t];
                        if (COUNT) n_lines += 1;
                    }
                }
                if (COUNT) n_lines += 1;
                occ_r = quad_sum(block_part(vr, t, orr, cls));
                if (l > 0) occ_l = quad_sum(block_part(vl, t, ol, cls));
                if (cls == 0 && fm.n_exc) {  // sparse exceptions sit in the stream as code 0
                    if (fm.n_exc == 1) {     // a text with one sentinel and nothing else outside its four letters: no search loop
                        const uint32_t e0 = s_exc[0];
                        occ_r -= e0 <= r ? 1u : 0u;
                        if (l > 0) occ_l -= e0 <= l - 1 ? 1u : 0u;
                    } else {
                        occ_r -= count_le(s_exc, 0u, fm.n_exc, r);
                        if (l > 0) occ_l -= count_le(s_exc, 0u, fm.n_exc, l - 1);
                    }
                }
            } else if (PACKED) {
            } else if (cls >= kClsDense) {  // one-hot bit vector of this symbol: one 64-byte block per rank
                const uint32_t d = cls - kClsDense;
                uint32_t orr, ol = 0;
                const uint4 vr = bv_load(fm, d, r, t, orr);
                uint4 vl = vr;
                if (l > 0) {
                    if ((l - 1) / kBvBits != r / kBvBits)
                        vl = bv_load(fm, d, l - 1, t, ol);
                    else
                        ol = (l - 1) % kBvBits;
                }
                occ_r = quad_sum(bv_part(vr, t, orr));
                if (l > 0) occ_l = quad_sum(bv_part(vl, t, ol));
            } else if (cls >= kClsSparse) {
                const uint32_t e = cls - kClsSparse;
                const uint32_t lo = fm.sparse_off[e], hi = fm.sparse_off[e + 1];
                occ_r = count_le(fm.exc_sym_pos, lo, hi, r) - lo;
                if (l > 0) occ_l = count_le(fm.exc_sym_pos, lo, hi, l - 1) - lo;
            }  // kClsZero: both stay 0
            const uint32_t pl = l, pr = r;
            if (!stop) {
                if (occ_r == 0) {  // fmindex.rs:167-170
                    stop = true;
                } else {
                    l = less_a + occ_l;  // fmindex.rs:171
                    r = less_a + occ_r - 1;
                    if (l > r)  // fmindex.rs:177-180
                        stop = true;
                    else
                        matched += 1;
                }
            }
            if (stop) {
                if (stop_tag == BG_FM_PANIC)
                    emit(BG_FM_PANIC, 0, 0, matched);
                else if (matched)
                    emit(BG_FM_PARTIAL, pl, pr + 1, matched);
                else
                    emit(BG_FM_ABSENT, 0, 0, 0);
                q += n_quads;
                fetch();
            } else if (pos == 0) {
                emit(BG_FM_COMPLETE, l, r + 1, matched);
                q += n_quads;
                fetch();
            }
        }
    }
    if (COUNT && t == 0 && n_lines) atomicAdd(seeds.lines, (unsigned long long)n_lines);
}

// K5 fast path for byte patterns over a DNA-like index (four 2-bit codes, no dense symbols): when a quad takes its next
// query, the wavefront turns the pattern into 2-bit codes — a coalesced byte load and a class lookup per 64 symbols, the
// 16 lanes of a DPP row OR their codes into a dword — parked in the quad's LDS slot (kFastSyms symbols); the LF loop
// then is the packed kernel's: one LDS read per
// step for the symbol, no byte load, no class / less lookups by byte, none of the sparse / dense / panic arms.  A
// pattern with a byte outside the four codes (N, lower case, anything that would make the reference panic) or longer
// than the slot is tagged kTagDeferred and answered by the generic kernel launched right behind (DEFER).  Same results
// (tests/test_gpu_fm.py, test_gpu_pack2.py); 466 -> ~560 M queries/s on the 100 Mbp index.
// PACKED: `pat` is a 2-bit stream already (pack2.hip, the index's codes; offsets in symbols): taking a query is a funnel
// shift of up to 16 dwords into the slot, nothing can be "bad".
// STEP2: the LF loop takes two pattern symbols per block access from the index's 2-step rank blocks (fm_kernels.h:
// Fm2Dev; f2.blocks2 != null), single steps — the last symbol of an odd-length pattern, and the two steps of a double
// step that found nothing — from the same blocks: half the requests of a query against a request-rate limit.
template <bool SEEDS, bool COUNT, bool PACKED = false, bool STEP2 = false>
__global__ __launch_bounds__(256) void fm_search_fast_kernel(FmDev fm, uint64_t n_q, const uint8_t* __restrict__ pat,
                                                             const uint64_t* __restrict__ pat_off, uint8_t* __restrict__ tag,
                                                             uint64_t* __restrict__ lower, uint64_t* __restrict__ upper,
                                                             uint32_t* __restrict__ matched_len, const SeedSrc seeds, const Fm2Dev f2) {
    __shared__ uint16_t s_class[256];
    __shared__ uint32_t s_less4[4];
    __shared__ uint32_t s_exc[kMaxExcLds];
    __shared__ uint32_t s_pk[64 * (kFastSyms / 16) + 1];  // (+1: the 2-step funnel reads one dword past a slot's last)
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_class[i] = fm.sym_class[i];
    if (threadIdx.x < 4) s_less4[threadIdx.x] = fm.less[(seeds.code_bytes >> (8 * threadIdx.x)) & 0xFFu];
    for (uint32_t i = threadIdx.x; i < fm.n_exc; i += blockDim.x) s_exc[i] = fm.exc_pos[i];
    if (threadIdx.x == 0) s_pk[64 * (kFastSyms / 16)] = 0;
    __shared__ uint32_t s_c2[16], s_e2pos[kMaxExc2], s_e2nib[kMaxExc2];
    if (STEP2 && threadIdx.x < 16) s_c2[threadIdx.x] = f2.c2[threadIdx.x];
    if (STEP2 && threadIdx.x < kMaxExc2) {
        s_e2pos[threadIdx.x] = f2.exc_pos[threadIdx.x];
        s_e2nib[threadIdx.x] = f2.exc_nib[threadIdx.x];
    }
    __syncthreads();

    const uint32_t t = threadIdx.x & 3;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t* const slot = s_pk + (threadIdx.x >> 2) * (kFastSyms / 16);
    uint32_t* const wave_slots = s_pk + (threadIdx.x >> 6) * 16 * (kFastSyms / 16);  // the 16 slots of this wavefront
    const uint64_t n_quads = (uint64_t)gridDim.x * (blockDim.x >> 2);
    uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    bool active = false, need = true, force1 = false;
    uint32_t pos = 0, l = 0, r = 0, matched = 0, n_lines = 0;

    // Taking the next query is a job of the WHOLE wavefront, one waiting quad at a time (round 3; a quad packing its own
    // pattern ran ~300 instructions with 4 of 64 lanes active, 15 % of the kernel): the quad's query index, pattern
    // offset and length are wave-uniform (read from the quad's leader lane), lane i looks symbol base + i up — one
    // coalesced byte load per 64 symbols — and the 16 lanes of a DPP row OR their 2-bit codes into the dword of their 16
    // symbols, which the row's last lane writes into the quad's LDS slot.  ~30 instructions per query.
    auto fetch_all = [&]() {
        uint64_t waiting = __ballot(need && t == 0);  // leaders of the quads that wait
        while (waiting) {
            const uint32_t leader = (uint32_t)__ffsll((unsigned long long)waiting) - 1u;
            const bool mine = (lane >> 2) == (leader >> 2);
            const uint64_t qs = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(q >> 32), leader) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)q, leader);
            if (qs >= n_q) {  // nothing left for this quad
                if (mine) {
                    need = false;
                    active = false;
                }
                waiting &= waiting - 1;
                continue;
            }
            uint64_t off;
            uint32_t len;
            if (SEEDS) {
                const uint64_t rd = qs / seeds.S;
                const uint32_t k = (uint32_t)(qs - rd * seeds.S);
                const uint64_t o = pat_off[rd];
                off = o + (uint64_t)k * seeds.stride;
                len = (uint64_t)k * seeds.stride + seeds.seed_len <= pat_off[rd + 1] - o ? seeds.seed_len : 0u;
            } else {
                off = pat_off[qs];
                const uint64_t len64 = pat_off[qs + 1] - off;
                len = len64 > kFastSyms ? kFastSyms + 1 : (uint32_t)len64;
            }
            bool bad = len > kFastSyms;
            if (PACKED && len && !bad) {
                uint32_t* const dst = wave_slots + (leader >> 2) * (kFastSyms / 16);
                const uint32_t* pk = (const uint32_t*)pat;
                if (lane < ((len + 15u) >> 4)) {  // dword `lane` of the slot: symbols off + 16 lane ... of the stream
                    const uint64_t w0 = (off >> 4) + lane;
                    const uint32_t sh = 2u * ((uint32_t)off & 15u);
                    const uint32_t lo = pk[w0], hi = pk[w0 + 1];  // the stream is padded by one dword
                    dst[lane] = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
                }
            } else if (len && !bad) {
                uint32_t* const dst = wave_slots + (leader >> 2) * (kFastSyms / 16);
                for (uint32_t base = 0; base < len; base += 64) {
                    const uint32_t idx = base + lane;
                    uint32_t c = 0;
                    if (idx < len) {
                        c = s_class[pat[off + idx]];
                        bad = bad || c >= 4;
                    }
                    uint32_t word = (c & 3u) << (2 * (lane & 15u));
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
                    if ((lane & 15u) == 15u && base + (lane & ~15u) < len) dst[(base >> 4) + (lane >> 4)] = word;
                }
                bad = __any(bad);
            }
            if (len == 0 || bad) {  // empty: Absent (fmindex.rs:185-207); a byte without a code / too long: the generic kernel's
                if (lane == leader) {
                    if (len == 0) {
                        tag[qs] = (uint8_t)BG_FM_ABSENT;
                        lower[qs] = 0;
                        upper[qs] = 0;
                        matched_len[qs] = 0;
                    } else {
                        tag[qs] = kTagDeferred;
                    }
                }
                if (mine) q += n_quads;  // the same quad takes the next one in the next turn of this loop
                continue;
            }
            if (mine) {
                pos = len;
                l = 0;
                r = fm.n - 1;  // fmindex.rs:148
                matched = 0;
                active = true;
                need = false;
                force1 = false;
            }
            waiting &= waiting - 1;
        }
    };
    auto emit = [&](uint32_t tg, uint32_t lo, uint32_t hi, uint32_t ml) {
        if (t == 0) {
            tag[q] = (uint8_t)tg;
            lower[q] = lo;
            upper[q] = hi;
            matched_len[q] = ml;
        }
    };
    for (;;) {
        if (__any(need)) fetch_all();
        if (!__any(active)) break;
        if (STEP2 && active) {
            // two iterations of the loop at fmindex.rs:160-182 per block access (fm_kernels.h), or one (`single`: the last
            // symbol of an odd length, and the two symbols of a pair that no row of the interval has in front).  One code
            // path for both: k symbols, a 4-bit code whose low half is masked off for a single step, and ONE end
            // condition — Occ(r) == Occ(l - 1) is "the pair does not occur" for a double step and, for a single one, both
            // of the reference's (occ_r == 0 implies it; l > r is it), which report the same (pl, pr, matched_len)
            const bool single = force1 || pos == 1;
            const uint32_t k = single ? 1u : 2u;
            const uint32_t p2 = pos - k, ix = p2 >> 4;
            // symbols pos-2 (second: low bits) and pos-1 (first) as one nibble; a single step reads its symbol into the high half
            uint32_t c = __builtin_amdgcn_alignbit(slot[ix + 1], slot[ix], 2 * (p2 & 15u));
            c = single ? (c & 3u) << 2 : (c & 15u);
            const uint32_t base = single ? s_less4[c >> 2] : s_c2[c];
            const uint32_t lm1 = l ? l - 1 : 0u;
            const uint32_t br = r / kSym2PerBlock, orr = r % kSym2PerBlock, bl = lm1 / kSym2PerBlock, ol = lm1 % kSym2PerBlock;
            const uint4 rc = f2.blocks2[(uint64_t)br * 8 + t], rs = f2.blocks2[(uint64_t)br * 8 + 4 + t];
            uint4 lc = rc, ls = rs;
            if (bl != br) {
                lc = f2.blocks2[(uint64_t)bl * 8 + t];
                ls = f2.blocks2[(uint64_t)bl * 8 + 4 + t];
                if (COUNT) n_lines += 1;
            }
            if (COUNT) n_lines += 1;
            const Pair2Key key2 = pair2_key(c, single);
            uint32_t occ_r = quad_sum(block2_part(rc, rs, t, orr, c, single, key2));
            uint32_t occ_l = quad_sum(block2_part(lc, ls, t, ol, c, single, key2));
            // positions that hold a 0 for a symbol without a code: the first two inline (one sentinel: exactly two; unused
            // entries sit at position 2^32 - 1, which no rank reaches), more of them (several sentinels) in a loop
            {
                const uint32_t key = single ? 16u : c;  // what an entry must say to have been counted: "first component" / this nibble
                const uint32_t e0 = f2.exc_pos[0], n0 = single ? (f2.exc_nib[0] & 16u) | (c ? 32u : 0u) : (f2.exc_nib[0] & 15u);
                const uint32_t e1 = f2.exc_pos[1], n1 = single ? (f2.exc_nib[1] & 16u) | (c ? 32u : 0u) : (f2.exc_nib[1] & 15u);
                occ_r -= (n0 == key && e0 <= r) ? 1u : 0u;
                occ_l -= (n0 == key && e0 <= lm1) ? 1u : 0u;
                occ_r -= (n1 == key && e1 <= r) ? 1u : 0u;
                occ_l -= (n1 == key && e1 <= lm1) ? 1u : 0u;
                for (uint32_t e = 2; e < f2.n_exc; e++) {  // (uniform)
                    const uint32_t pe = s_e2pos[e], ne = s_e2nib[e];
                    const uint32_t nk = single ? (ne & 16u) | (c ? 32u : 0u) : (ne & 15u);
                    occ_r -= (nk == key && pe <= r) ? 1u : 0u;
                    occ_l -= (nk == key && pe <= lm1) ? 1u : 0u;
                }
            }
            occ_l = l ? occ_l : 0u;
            const bool empty = occ_r == occ_l;
            if (empty && !single) {
                force1 = true;  // nothing changes: the two single steps that follow say how the query ends
            } else if (empty) {
                if (matched)
                    emit(BG_FM_PARTIAL, l, r + 1, matched);
                else
                    emit(BG_FM_ABSENT, 0, 0, 0);
                need = true;
            } else {
                l = base + occ_l;  // fmindex.rs:171 (twice for a double step)
                r = base + occ_r - 1;
                pos = p2;
                matched += k;
                if (pos == 0) {
                    emit(BG_FM_COMPLETE, l, r + 1, matched);
                    need = true;
                }
            }
            if (need) {
                q += n_quads;
                active = false;
                force1 = false;
            }
        }
        if (!STEP2 && active) {
            // one iteration of the loop at fmindex.rs:160-182; the symbol is a code already
            pos -= 1;
            const uint32_t a = (slot[pos >> 4] >> (2 * (pos & 15u))) & 3u;
            const uint32_t less_a = s_less4[a];
            // straight-line up to the (rare) second block load: l == 0 ranks position 0 and drops the result
            const uint32_t lm1 = l ? l - 1 : 0u;
            const uint32_t br = r / kSymPerBlock, orr = r - br * kSymPerBlock;
            const uint32_t bl = lm1 / kSymPerBlock, ol = lm1 - bl * kSymPerBlock;
            const uint4 vr = fm.blocks[(uint64_t)br * 4 + t];
            uint4 vl = vr;
            if (bl != br) {
                vl = fm.blocks[(uint64_t)bl * 4 + t];
                if (COUNT) n_lines += 1;
            }
            if (COUNT) n_lines += 1;
            uint32_t occ_r = quad_sum(block_part_bf(vr, t, orr, a));
            uint32_t occ_l = quad_sum(block_part_bf(vl, t, ol, a));
            if (fm.n_exc == 1) {  // (uniform) the one sparse exception — the sentinel — sits in the stream as code 0
                const uint32_t e0 = s_exc[0];
                occ_r -= (a == 0 && e0 <= r) ? 1u : 0u;
                occ_l -= (a == 0 && e0 <= lm1) ? 1u : 0u;
            } else if (a == 0 && fm.n_exc) {
                occ_r -= count_le(s_exc, 0u, fm.n_exc, r);
                occ_l -= count_le(s_exc, 0u, fm.n_exc, lm1);
            }
            occ_l = l ? occ_l : 0u;
            const uint32_t pl = l, pr = r;
            bool stop = occ_r == 0;  // fmindex.rs:167-170
            if (!stop) {
                l = less_a + occ_l;  // fmindex.rs:171
                r = less_a + occ_r - 1;
                if (l > r)  // fmindex.rs:177-180
                    stop = true;
                else
                    matched += 1;
            }
            if (stop) {
                if (matched)
                    emit(BG_FM_PARTIAL, pl, pr + 1, matched);
                else
                    emit(BG_FM_ABSENT, 0, 0, 0);
                need = true;
            } else if (pos == 0) {
                emit(BG_FM_COMPLETE, l, r + 1, matched);
                need = true;
            }
            if (need) {
                q += n_quads;
                active = false;
            }
        }
    }
    if (COUNT && t == 0 && n_lines) atomicAdd(seeds.lines, (unsigned long long)n_lines);
}

// K5x — fm_search_fast2x_kernel: fm_search_fast_kernel<STEP2> with TWO queries per quad (round 5).  With its requests
// halved the search waits neither for memory throughput nor for the vector unit (0.60 of the 128-byte gather rate at
// 3 Gbp, VALU 31 % busy, eight wavefronts per SIMD — the most the hardware holds): it waits for the latency of one block
// access per step, and the only parallelism left to add is inside a wavefront.  Every quad walks two independent queries:
// phase A computes the block addresses of both and issues their loads — all of them unconditional and in one basic block,
// so that the second query's requests leave before the first one's data is waited for (the line of l - 1 is requested even
// where it is the line of r: a hit in the vector cache) — phase B ranks and updates both.  Stream (quad, u) takes queries
// (quad * 2 + u) + k * (quads * 2); results, deferral and LDS pattern slots as in fm_search_fast_kernel.
// Measured (profiles/r05_fm_ilp.txt, 10 M x 100 bp): 846 -> 1076 M queries/s on the 100 Mbp index, 608 -> 794 M at 3 Gbp
// (packed patterns 908 -> 1121 and 657 -> 860).  86 VGPRs: five wavefronts per SIMD, ten queries in flight per SIMD-slot
// against eight.  More is not better: three queries per quad (116 VGPRs, four wavefronts: twelve) reach 988 / 752, four
// (147: three wavefronts) 838 / 673, and the same two queries compiled for six wavefronts (80 VGPRs, three values in
// scratch) 1006 / 715 — the optimum is where this kernel sits.
// WIDE (round 6): the same kernel on 64-bit positions (FmLayout<true>: l, r, less, C2 and the exception positions are
// uint64; the blocks' counters are relative to a superblock whose absolute base — sixteen pair codes + four single-step sums —
// is one more load per rank, issued in phase A next to the block's).  The narrow instantiation compiles to the code it was.
template <bool SEEDS, bool COUNT, bool PACKED, int U, bool WIDE = false>
__global__ __launch_bounds__(256) void fm_search_fast2x_kernel(typename FmLayout<WIDE>::Dev fm, uint64_t n_q, const uint8_t* __restrict__ pat,
                                                               const uint64_t* __restrict__ pat_off, uint8_t* __restrict__ tag,
                                                               uint64_t* __restrict__ lower, uint64_t* __restrict__ upper,
                                                               uint32_t* __restrict__ matched_len, const SeedSrc seeds,
                                                               const typename FmLayout<WIDE>::Dev2 f2) {
    using P = typename FmLayout<WIDE>::Pos;
    constexpr uint32_t SLOT = kFastSyms / 16;  // dwords per pattern slot
    __shared__ uint16_t s_class[256];
    __shared__ P s_less4[4];
    __shared__ uint32_t s_pk[64 * U * SLOT + 1];  // (+1: the 2-step funnel reads one dword past a slot's last)
    __shared__ P s_c2[16], s_e2pos[kMaxExc2];
    __shared__ uint32_t s_e2nib[kMaxExc2];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_class[i] = fm.sym_class[i];
    if (threadIdx.x < 4) s_less4[threadIdx.x] = fm.less[(seeds.code_bytes >> (8 * threadIdx.x)) & 0xFFu];
    if (threadIdx.x == 0) s_pk[64 * U * SLOT] = 0;
    if (threadIdx.x < 16) s_c2[threadIdx.x] = f2.c2[threadIdx.x];
    if (threadIdx.x < kMaxExc2) {
        s_e2pos[threadIdx.x] = f2.exc_pos[threadIdx.x];
        s_e2nib[threadIdx.x] = f2.exc_nib[threadIdx.x];
    }
    __syncthreads();

    const uint32_t t = threadIdx.x & 3;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n_streams = (uint64_t)gridDim.x * (blockDim.x >> 2) * U;
    struct St {
        uint64_t q;
        P l, r;
        uint32_t pos, matched;
        bool active, need, force1;
    };
    St S[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        S[u].q = ((uint64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2)) * U + u;
        S[u].pos = S[u].l = S[u].r = S[u].matched = 0;
        S[u].active = false;
        S[u].need = true;
        S[u].force1 = false;
    }
    uint32_t n_lines = 0;

    // taking the next query of stream u: a job of the whole wavefront, one waiting quad at a time (fm_search_fast_kernel)
    auto fetch_all = [&](St& s, const int u) {
        uint64_t waiting = __ballot(s.need && t == 0);
        while (waiting) {
            const uint32_t leader = (uint32_t)__ffsll((unsigned long long)waiting) - 1u;
            const bool mine = (lane >> 2) == (leader >> 2);
            const uint64_t qs = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(s.q >> 32), leader) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)s.q, leader);
            if (qs >= n_q) {
                if (mine) {
                    s.need = false;
                    s.active = false;
                }
                waiting &= waiting - 1;
                continue;
            }
            uint64_t off;
            uint32_t len;
            if (SEEDS) {
                const uint64_t rd = qs / seeds.S;
                const uint32_t k = (uint32_t)(qs - rd * seeds.S);
                const uint64_t o = pat_off[rd];
                off = o + (uint64_t)k * seeds.stride;
                len = (uint64_t)k * seeds.stride + seeds.seed_len <= pat_off[rd + 1] - o ? seeds.seed_len : 0u;
            } else {
                off = pat_off[qs];
                const uint64_t len64 = pat_off[qs + 1] - off;
                len = len64 > kFastSyms ? kFastSyms + 1 : (uint32_t)len64;
            }
            bool bad = len > kFastSyms;
            uint32_t* const dst = s_pk + ((((threadIdx.x >> 6) * 16) + (leader >> 2)) * U + u) * SLOT;
            if (PACKED && len && !bad) {
                const uint32_t* pk = (const uint32_t*)pat;
                if (lane < ((len + 15u) >> 4)) {
                    const uint64_t w0 = (off >> 4) + lane;
                    const uint32_t sh = 2u * ((uint32_t)off & 15u);
                    const uint32_t lo = pk[w0], hi = pk[w0 + 1];  // the stream is padded by one dword
                    dst[lane] = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
                }
            } else if (len && !bad) {
                for (uint32_t base = 0; base < len; base += 64) {
                    const uint32_t idx = base + lane;
                    uint32_t c = 0;
                    if (idx < len) {
                        c = s_class[pat[off + idx]];
                        bad = bad || c >= 4;
                    }
                    uint32_t word = (c & 3u) << (2 * (lane & 15u));
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
                    word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
                    if ((lane & 15u) == 15u && base + (lane & ~15u) < len) dst[(base >> 4) + (lane >> 4)] = word;
                }
                bad = __any(bad);
            }
            if (len == 0 || bad) {  // empty: Absent (fmindex.rs:185-207); a byte without a code / too long: the generic kernel's
                if (lane == leader) {
                    if (len == 0) {
                        tag[qs] = (uint8_t)BG_FM_ABSENT;
                        lower[qs] = 0;
                        upper[qs] = 0;
                        matched_len[qs] = 0;
                    } else {
                        tag[qs] = kTagDeferred;
                    }
                }
                if (mine) s.q += n_streams;  // the same stream takes the next one in the next turn of this loop
                continue;
            }
            if (mine) {
                s.pos = len;
                s.l = 0;
                s.r = fm.n - 1;  // fmindex.rs:148
                s.matched = 0;
                s.active = true;
                s.need = false;
                s.force1 = false;
            }
            waiting &= waiting - 1;
        }
    };
    auto emit = [&](uint64_t q, uint32_t tg, P lo, P hi, uint32_t ml) {
        if (t == 0) {
            tag[q] = (uint8_t)tg;
            lower[q] = lo;
            upper[q] = hi;
            matched_len[q] = ml;
        }
    };
    for (;;) {
        bool any_active = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (__any(S[u].need)) fetch_all(S[u], u);
            any_active |= S[u].active;
        }
        if (!__any(any_active)) break;
        // ---- phase A: the symbols and the block loads of both streams (a stream without a query reads block 0)
        uint4 rc[U], rs[U], lc[U], ls[U];
        uint32_t c[U], p2[U], orr[U], ol[U];
        P lm1[U];
        uint64_t sbr[U], sbl[U];  // WIDE: the superblock's absolute count of this step's code (or single-step sum)
        bool single[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const St& s = S[u];
            single[u] = s.force1 || s.pos == 1;
            p2[u] = s.active ? s.pos - (single[u] ? 1u : 2u) : 0u;
            const uint32_t* const slot = s_pk + ((threadIdx.x >> 2) * U + u) * SLOT;
            const uint32_t ix = p2[u] >> 4;
            const uint32_t cc = __builtin_amdgcn_alignbit(slot[ix + 1], slot[ix], 2 * (p2[u] & 15u));
            c[u] = single[u] ? (cc & 3u) << 2 : (cc & 15u);
            lm1[u] = s.l ? s.l - 1 : 0u;
            const P r_ = s.active ? s.r : 0u, l_ = s.active ? lm1[u] : 0u;
            const P br = r_ / kSym2PerBlock, bl = l_ / kSym2PerBlock;
            orr[u] = (uint32_t)(r_ % kSym2PerBlock);
            ol[u] = (uint32_t)(l_ % kSym2PerBlock);
            rc[u] = f2.blocks2[(uint64_t)br * 8 + t];
            rs[u] = f2.blocks2[(uint64_t)br * 8 + 4 + t];
            lc[u] = f2.blocks2[(uint64_t)bl * 8 + t];
            ls[u] = f2.blocks2[(uint64_t)bl * 8 + 4 + t];
            if constexpr (WIDE) {
                const uint32_t bi = single[u] ? 16u + (c[u] >> 2) : c[u];
                sbr[u] = f2.sb2[(br >> f2.sb_shift) * 20 + bi];
                sbl[u] = f2.sb2[(bl >> f2.sb_shift) * 20 + bi];
            } else {
                sbr[u] = sbl[u] = 0;
            }
            if (COUNT && s.active) n_lines += bl != br ? 2u : 1u;
        }
        // ---- phase B: fmindex.rs:160-182, twice per block access (see fm_search_fast_kernel<STEP2>)
#pragma unroll
        for (int u = 0; u < U; u++) {
            St& s = S[u];
            if (!s.active) continue;
            const uint32_t k = single[u] ? 1u : 2u;
            const P base = single[u] ? s_less4[c[u] >> 2] : s_c2[c[u]];
            const Pair2Key key2 = pair2_key(c[u], single[u]);
            P occ_r = quad_sum(block2_part(rc[u], rs[u], t, orr[u], c[u], single[u], key2));
            P occ_l = quad_sum(block2_part(lc[u], ls[u], t, ol[u], c[u], single[u], key2));
            if constexpr (WIDE) {
                occ_r += sbr[u];
                occ_l += sbl[u];
            }
            {
                const uint32_t key = single[u] ? 16u : c[u];
                const P e0 = f2.exc_pos[0];
                const uint32_t n0 = single[u] ? (f2.exc_nib[0] & 16u) | (c[u] ? 32u : 0u) : (f2.exc_nib[0] & 15u);
                const P e1 = f2.exc_pos[1];
                const uint32_t n1 = single[u] ? (f2.exc_nib[1] & 16u) | (c[u] ? 32u : 0u) : (f2.exc_nib[1] & 15u);
                occ_r -= (n0 == key && e0 <= s.r) ? 1u : 0u;
                occ_l -= (n0 == key && e0 <= lm1[u]) ? 1u : 0u;
                occ_r -= (n1 == key && e1 <= s.r) ? 1u : 0u;
                occ_l -= (n1 == key && e1 <= lm1[u]) ? 1u : 0u;
                for (uint32_t e = 2; e < f2.n_exc; e++) {  // (uniform)
                    const P pe = s_e2pos[e];
                    const uint32_t ne = s_e2nib[e];
                    const uint32_t nk = single[u] ? (ne & 16u) | (c[u] ? 32u : 0u) : (ne & 15u);
                    occ_r -= (nk == key && pe <= s.r) ? 1u : 0u;
                    occ_l -= (nk == key && pe <= lm1[u]) ? 1u : 0u;
                }
            }
            occ_l = s.l ? occ_l : 0u;
            const bool empty = occ_r == occ_l;
            if (empty && !single[u]) {
                s.force1 = true;  // nothing changes: the two single steps that follow say how the query ends
            } else if (empty) {
                if (s.matched)
                    emit(s.q, BG_FM_PARTIAL, s.l, s.r + 1, s.matched);
                else
                    emit(s.q, BG_FM_ABSENT, 0, 0, 0);
                s.need = true;
            } else {
                s.l = base + occ_l;  // fmindex.rs:171 (twice for a double step)
                s.r = base + occ_r - 1;
                s.pos = p2[u];
                s.matched += k;
                if (s.pos == 0) {
                    emit(s.q, BG_FM_COMPLETE, s.l, s.r + 1, s.matched);
                    s.need = true;
                }
            }
            if (s.need) {
                s.q += n_streams;
                s.active = false;
                s.force1 = false;
            }
        }
    }
    if (COUNT && t == 0 && n_lines) atomicAdd(seeds.lines, (unsigned long long)n_lines);
}

// jump-table construction: every kJumpK-mer over the four coded bytes as a pattern ...
__global__ __launch_bounds__(256) void fm_jump_patterns_kernel(uint32_t code_byte, uint8_t* pat, uint64_t* pat_off) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = 1ull << (2 * kJumpK);
    if (idx > n) return;
    pat_off[idx] = idx * kJumpK;
    if (idx == n) return;
    for (uint32_t j = 0; j < kJumpK; j++) pat[idx * kJumpK + j] = (uint8_t)(code_byte >> (8 * ((idx >> (2 * j)) & 3u)));
}
// ... and its search result as a table entry {l, r, depth, 0}
__global__ __launch_bounds__(256) void fm_jump_pack_kernel(const uint8_t* tag, const uint64_t* lower, const uint64_t* upper,
                                                           const uint32_t* matched, uint4* jump) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1ull << (2 * kJumpK))) return;
    const uint32_t tg = tag[idx];
    uint4 e = make_uint4(0, 0, 0, 0);
    if (tg == BG_FM_COMPLETE || tg == BG_FM_PARTIAL) e = make_uint4((uint32_t)lower[idx], (uint32_t)upper[idx] - 1u, matched[idx], 0);
    jump[idx] = e;
}

}  // namespace

// How every byte value is ranked, decided from the BWT's histogram (shared by the host and the device builder).
struct FmClasses {
    bool gen = false;  // dense symbols exist: three coded bytes, code 0 = "something else"
    int n_codes = 0;
    int code_of[256], sparse_of[256], dense_of[256];
    std::vector<int> sparse_syms, dense_syms;
    uint16_t cls[256];
};
static void assign_classes(const uint64_t hist[256], const bool in_alpha[256], FmClasses& k) {
    // the most frequent byte values get the 2-bit codes (ties: smaller byte first)
    int order[256];
    std::iota(order, order + 256, 0);
    std::stable_sort(order, order + 256, [&](int a, int b) { return hist[a] > hist[b]; });
    uint64_t beyond4 = 0;
    for (int i = 4; i < 256; i++) beyond4 += hist[order[i]];
    k.gen = beyond4 > kMaxExcLds;  // too many exceptions for the LDS list: dense symbols get bit vectors
    std::fill(k.code_of, k.code_of + 256, -1);
    std::fill(k.sparse_of, k.sparse_of + 256, -1);
    std::fill(k.dense_of, k.dense_of + 256, -1);
    k.n_codes = 0;
    if (!k.gen) {
        for (int i = 0; i < 4 && hist[order[i]] > 0; i++) k.code_of[order[i]] = k.n_codes++;
        for (int c = 0; c < 256; c++)
            if (hist[c] && k.code_of[c] < 0) k.sparse_syms.push_back(c);
    } else {
        for (int i = 0; i < 3; i++) k.code_of[order[i]] = 1 + k.n_codes++;  // code 0 = "none of the three"
        uint64_t cum = 0;
        for (int i = 255; i >= 3; i--) {  // ascending frequency: the rare ones stay lists while they fit
            const int c = order[i];
            if (!hist[c]) continue;
            if (k.dense_syms.empty() && cum + hist[c] <= kMaxExcLds) {
                cum += hist[c];
                k.sparse_syms.push_back(c);
            } else {
                k.dense_syms.push_back(c);
            }
        }
        std::sort(k.sparse_syms.begin(), k.sparse_syms.end());
        std::sort(k.dense_syms.begin(), k.dense_syms.end());
    }
    for (size_t e = 0; e < k.sparse_syms.size(); e++) k.sparse_of[k.sparse_syms[e]] = (int)e;
    for (size_t d = 0; d < k.dense_syms.size(); d++) k.dense_of[k.dense_syms[d]] = (int)d;
    for (int c = 0; c < 256; c++) {
        if (!in_alpha[c])
            k.cls[c] = kClsPanic;
        else if (k.code_of[c] >= 0)
            k.cls[c] = (uint16_t)k.code_of[c];
        else if (hist[c] == 0)
            k.cls[c] = kClsZero;
        else if (k.sparse_of[c] >= 0)
            k.cls[c] = (uint16_t)(kClsSparse + k.sparse_of[c]);
        else
            k.cls[c] = (uint16_t)(kClsDense + k.dense_of[c]);
    }
}

static int fm_build_host_impl(bg_ctx* ctx, const uint8_t* bwt, uint64_t n, const uint64_t* less,
                              uint32_t less_len, uint32_t occ_k, const uint8_t* alphabet,
                              uint32_t n_sym, bg_fm** out) {
    if (!ctx || !bwt || !less || !alphabet || !out || n == 0 || n_sym == 0 || occ_k == 0)
        return BG_ERR_INVALID_ARG;
    if (n > (1ull << 40)) return BG_ERR_TOO_LARGE;
    if (n >= fm_wide_threshold(ctx)) {
        // 64-bit positions (fm_wide.hip): the BWT goes up and the device builder lays the index out (the caller's less[] kept)
        BG_HIP(hipSetDevice(ctx->device));
        uint8_t* d_b = nullptr;
        BG_HIP(hipMalloc((void**)&d_b, n));
        int rcw = bg_copy_pieces(d_b, bwt, n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess ? BG_OK : BG_ERR_HIP;
        if (!rcw) rcw = fm_wide_build_dev(ctx, d_b, n, alphabet, n_sym, less, less_len, nullptr, out, ctx->stream);
        hipStreamSynchronize(ctx->stream);
        hipFree(d_b);
        return rcw;
    }
    // Alphabet (alphabets/mod.rs:49-60): set of bytes; m = max_symbol + 1 (bwt.rs:96-99)
    bool in_alpha[256] = {};
    uint32_t max_symbol = 0;
    for (uint32_t i = 0; i < n_sym; i++) {
        in_alpha[alphabet[i]] = true;
        max_symbol = std::max<uint32_t>(max_symbol, alphabet[i]);
    }
    const uint32_t m = max_symbol + 1;
    if (less_len != max_symbol + 2) return BG_ERR_INVALID_ARG;
    if ((uint32_t)'$' < m) in_alpha['$'] = true;  // bwt.rs:101-104: '$' is always tabulated

    // work is cut into chunks of whole 2-bit blocks AND whole bit-vector blocks (lcm(192, 480) = 960 symbols)
    const unsigned nthreads = std::max(1u, bg_host_threads());
    const uint64_t kAlign = 960;
    const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>(nthreads * 4, (n + kAlign - 1) / kAlign));
    const uint64_t per = ((n + n_chunks - 1) / n_chunks + kAlign - 1) / kAlign * kAlign;
    auto chunk_lo = [&](uint64_t c) { return std::min(n, c * per); };
    auto run_chunks = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < std::min<uint64_t>(nthreads, n_chunks); t++)
            th.emplace_back([&, t] {
                for (uint64_t c = t; c < n_chunks; c += nthreads) fn(c);
            });
        for (auto& x : th) x.join();
    };
    std::vector<std::array<uint64_t, 256>> chist(n_chunks);
    run_chunks([&](uint64_t c) {
        std::array<uint64_t, 256> h{};
        for (uint64_t i = chunk_lo(c), e = chunk_lo(c + 1); i < e; i++) h[bwt[i]]++;
        chist[c] = h;
    });
    uint64_t hist[256] = {};
    for (auto& h : chist)
        for (int c = 0; c < 256; c++) hist[c] += h[c];
    for (uint32_t c = m; c < 256; c++)
        if (hist[c]) return BG_ERR_OUT_OF_ALPHABET;  // Occ::new: curr_occ[c] out of bounds

    FmClasses K;
    assign_classes(hist, in_alpha, K);
    const bool gen = K.gen;
    const int n_codes = K.n_codes;
    const int *code_of = K.code_of, *sparse_of = K.sparse_of, *dense_of = K.dense_of;
    const std::vector<int>&sparse_syms = K.sparse_syms, &dense_syms = K.dense_syms;
    const uint16_t* cls = K.cls;

    const uint64_t nblk = (n + kSymPerBlock - 1) / kSymPerBlock;
    const uint64_t nbv = (n + kBvBits - 1) / kBvBits;
    const size_t n_dense = dense_syms.size();
    std::vector<uint32_t> blocks(nblk * 16, 0);
    std::vector<uint32_t> bitvecs(n_dense * nbv * 16, 0);
    std::vector<std::vector<uint32_t>> c_exc(n_chunks);                  // sparse positions per chunk, in order
    std::vector<std::vector<std::vector<uint32_t>>> c_exc_sym(n_chunks);  // ... and per sparse symbol
    run_chunks([&](uint64_t ck) {
        uint64_t before[256] = {};
        for (uint64_t c2 = 0; c2 < ck; c2++)
            for (int c = 0; c < 256; c++) before[c] += chist[c2][c];
        uint32_t running[4] = {0, 0, 0, 0};
        for (int c = 0; c < 256; c++) running[code_of[c] >= 0 ? code_of[c] : 0] += (uint32_t)before[c];
        std::vector<uint32_t> drun(n_dense);
        for (size_t d = 0; d < n_dense; d++) drun[d] = (uint32_t)before[dense_syms[d]];
        c_exc_sym[ck].resize(sparse_syms.size());
        const uint64_t lo = chunk_lo(ck), hi = chunk_lo(ck + 1);
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t s = i % kSymPerBlock;
            uint32_t* blk = &blocks[(i / kSymPerBlock) * 16];
            if (s == 0)
                for (int c = 0; c < 4; c++) blk[c] = running[c];
            if (n_dense && i % kBvBits == 0)
                for (size_t d = 0; d < n_dense; d++) bitvecs[(d * nbv + i / kBvBits) * 16] = drun[d];
            const uint8_t ch = bwt[i];
            int code = code_of[ch];
            if (code < 0) {
                code = 0;
                if (sparse_of[ch] >= 0) {
                    c_exc[ck].push_back((uint32_t)i);
                    c_exc_sym[ck][sparse_of[ch]].push_back((uint32_t)i);
                } else {
                    const size_t d = (size_t)dense_of[ch];
                    const uint64_t o = i % kBvBits;
                    bitvecs[(d * nbv + i / kBvBits) * 16 + 1 + (o >> 5)] |= 1u << (o & 31);
                    drun[d]++;
                }
            }
            blk[4 + (s >> 4)] |= (uint32_t)code << (2 * (s & 15));
            running[code]++;
        }
    });
    std::vector<uint32_t> exc_pos, exc_sym_pos, sparse_off(sparse_syms.size() + 1, 0);
    for (auto& v : c_exc) exc_pos.insert(exc_pos.end(), v.begin(), v.end());
    for (size_t e = 0; e < sparse_syms.size(); e++) {
        for (uint64_t ck = 0; ck < n_chunks; ck++)
            exc_sym_pos.insert(exc_sym_pos.end(), c_exc_sym[ck][e].begin(), c_exc_sym[ck][e].end());
        sparse_off[e + 1] = (uint32_t)exc_sym_pos.size();
    }
    std::vector<uint8_t> exc_byte(exc_pos.size());
    for (size_t e = 0; e < exc_pos.size(); e++) exc_byte[e] = bwt[exc_pos[e]];
    bg_fm* fm = new bg_fm;
    fm->ctx = ctx;
    fm->less_len = less_len;
    fm->fmd_ok = true;
    for (int c = 0; c < 256; c++)
        if (hist[c] && (c == 0 || !strchr("ACGTNacgtn$", c))) fm->fmd_ok = false;
    for (int c = 0; c < 256; c++)
        if (code_of[c] >= 0) fm->code_byte[code_of[c]] = (uint8_t)c;
    fm->n_codes = gen ? 3 : n_codes;
    uint32_t less32[256] = {};
    for (uint32_t i = 0; i < less_len && i < 256; i++) less32[i] = (uint32_t)less[i];

    auto fail = [&](int rc) {
        bg_fm_free(fm);
        return rc;
    };
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(BG_ERR_NO_DEVICE);
    auto upload = [&](void** dptr, const void* src, size_t bytes) -> int {
        const size_t alloc = std::max<size_t>(bytes, 16);
        BG_HIP(hipMalloc(dptr, alloc));
        if (bytes) BG_HIP(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
        fm->bytes += alloc;
        return BG_OK;
    };
    int rc;
    if ((rc = upload(&fm->d_blocks, blocks.data(), blocks.size() * 4))) return fail(rc);
    if ((rc = upload(&fm->d_bitvecs, bitvecs.data(), bitvecs.size() * 4))) return fail(rc);
    if ((rc = upload(&fm->d_exc_pos, exc_pos.data(), exc_pos.size() * 4))) return fail(rc);
    if ((rc = upload(&fm->d_exc_sym_pos, exc_sym_pos.data(), exc_sym_pos.size() * 4))) return fail(rc);
    if ((rc = upload(&fm->d_sparse_off, sparse_off.data(), sparse_off.size() * 4))) return fail(rc);
    if ((rc = upload(&fm->d_exc_byte, exc_byte.data(), exc_byte.size()))) return fail(rc);
    if ((rc = upload(&fm->d_class, cls, 256 * sizeof(uint16_t)))) return fail(rc);
    memcpy(fm->h_class, cls, sizeof(fm->h_class));
    if ((rc = upload(&fm->d_less, less32, sizeof(less32)))) return fail(rc);
    if (gen && (rc = upload(&fm->d_bwt_raw, bwt, n))) return fail(rc);
    fm->dev.blocks = (const uint4*)fm->d_blocks;
    fm->dev.bitvecs = (const uint4*)fm->d_bitvecs;
    fm->dev.exc_pos = (const uint32_t*)fm->d_exc_pos;
    fm->dev.exc_sym_pos = (const uint32_t*)fm->d_exc_sym_pos;
    fm->dev.sparse_off = (const uint32_t*)fm->d_sparse_off;
    fm->dev.sym_class = (const uint16_t*)fm->d_class;
    fm->dev.less = (const uint32_t*)fm->d_less;
    fm->dev.bwt_raw = (const uint8_t*)fm->d_bwt_raw;
    fm->dev.n = (uint32_t)n;
    fm->dev.n_exc = gen ? 0u : (uint32_t)exc_pos.size();
    fm->dev.nbv_blocks = (uint32_t)nbv;
    fm->dev.n_dense = (uint32_t)n_dense;
    {
        // 2-step rank blocks (fm_step2.hip) lean on less[] being the BWT's own cumulative counts (LF maps the occurrences
        // of a symbol to the rows from less[symbol] on); a caller's less that says otherwise keeps single steps, where
        // the reference's arithmetic on whatever it was given is reproduced as it is
        bool consistent = true;
        uint64_t run = 0;
        for (uint32_t c = 0; c < m && consistent; c++) {
            if (hist[c] && less[c] != run) consistent = false;
            run += hist[c];
        }
        if (consistent) fm_build_step2(fm, ctx->stream);
    }
    *out = fm;
    return BG_OK;
}

// ---- the same index laid out on the device from a BWT that lives in HBM (bg_fm_build_dev) ---------------------------
namespace {

__global__ __launch_bounds__(256) void fmb_hist_kernel(const uint8_t* __restrict__ b, uint64_t n, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t s[256];
    s[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&s[b[i]], 1u);
    __syncthreads();
    if (s[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

struct ClsTab {
    uint8_t code[256];   // 2-bit code of the byte in the packed stream (0 for everything without one)
    uint8_t dense[256];  // 1 + dense id, 0: not dense
    uint8_t sparse[256]; // 1: sparse exception
};

// one thread per 2-bit block: 192 symbols -> 12 words + how many of each code it holds
__global__ __launch_bounds__(256) void fmb_blocks_kernel(const uint8_t* __restrict__ b, uint64_t n, uint64_t nblk, const ClsTab* __restrict__ tab,
                                                         uint32_t* __restrict__ blocks, uint32_t* __restrict__ cnt /* [4][nblk] */) {
    __shared__ uint8_t s_code[256];
    s_code[threadIdx.x] = tab->code[threadIdx.x];
    __syncthreads();
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    const uint64_t lo = blk * kSymPerBlock;
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t w = 0; w < 12; w++) {
        uint32_t word = 0;
        for (uint32_t t = 0; t < 16; t++) {
            const uint64_t i = lo + 16 * w + t;
            if (i < n) {
                const uint32_t code = s_code[b[i]];
                word |= code << (2 * t);
                c[code]++;
            }
        }
        blocks[blk * 16 + 4 + w] = word;
    }
    for (int k = 0; k < 4; k++) cnt[(uint64_t)k * nblk + blk] = c[k];
}
__global__ __launch_bounds__(256) void fmb_block_heads_kernel(uint64_t nblk, const uint32_t* __restrict__ scanned, uint32_t* __restrict__ blocks) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    for (int k = 0; k < 4; k++) blocks[blk * 16 + k] = scanned[(uint64_t)k * nblk + blk];
}
// one thread per (dense symbol, bit-vector block): 480 symbols -> 15 words + their population
__global__ __launch_bounds__(256) void fmb_bitvec_kernel(const uint8_t* __restrict__ b, uint64_t n, uint64_t nbv, uint32_t n_dense,
                                                         const uint8_t* __restrict__ dense_byte, uint32_t* __restrict__ bv,
                                                         uint32_t* __restrict__ cnt /* [n_dense][nbv] */) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)n_dense * nbv) return;
    const uint32_t d = (uint32_t)(idx / nbv);
    const uint64_t blk = idx - (uint64_t)d * nbv;
    const uint32_t sym = dense_byte[d];
    const uint64_t lo = blk * kBvBits;
    uint32_t total = 0;
    for (uint32_t w = 0; w < 15; w++) {
        uint32_t word = 0;
        for (uint32_t t = 0; t < 32; t++) {
            const uint64_t i = lo + 32 * w + t;
            if (i < n && b[i] == sym) word |= 1u << t;
        }
        bv[idx * 16 + 1 + w] = word;
        total += (uint32_t)__popc(word);
    }
    cnt[idx] = total;
}
__global__ __launch_bounds__(256) void fmb_bitvec_heads_kernel(uint64_t total, const uint32_t* __restrict__ scanned, uint32_t* __restrict__ bv) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) bv[idx * 16] = scanned[idx];
}
// sparse exceptions: (position, byte) appended in any order; the host sorts the few of them
__global__ __launch_bounds__(256) void fmb_sparse_kernel(const uint8_t* __restrict__ b, uint64_t n, const ClsTab* __restrict__ tab,
                                                         uint32_t cap, uint32_t* __restrict__ n_out, uint2* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t ch = b[i];
    if (tab->sparse[ch]) {
        const uint32_t k = atomicAdd(n_out, 1u);
        if (k < cap) out[k] = make_uint2((uint32_t)i, ch);
    }
}

}  // namespace

static int fm_build_dev_impl(bg_ctx* ctx, const uint8_t* d_bwt, uint64_t n, uint32_t occ_k, const uint8_t* alphabet, uint32_t n_sym,
                             uint64_t* less_out, bg_fm** out, void* stream) {
    if (!ctx || !d_bwt || !alphabet || !out || n == 0 || n_sym == 0 || occ_k == 0) return BG_ERR_INVALID_ARG;
    if (n > (1ull << 40)) return BG_ERR_TOO_LARGE;
    if (n >= fm_wide_threshold(ctx))  // 64-bit positions (fm_wide.hip)
        return fm_wide_build_dev(ctx, d_bwt, n, alphabet, n_sym, nullptr, 0, less_out, out, (hipStream_t)stream);
    hipStream_t st = (hipStream_t)stream;
    BG_HIP(hipSetDevice(ctx->device));
    bool in_alpha[256] = {};
    uint32_t max_symbol = 0;
    for (uint32_t i = 0; i < n_sym; i++) {
        in_alpha[alphabet[i]] = true;
        max_symbol = std::max<uint32_t>(max_symbol, alphabet[i]);
    }
    const uint32_t m = max_symbol + 1;
    if ((uint32_t)'$' < m) in_alpha['$'] = true;  // bwt.rs:101-104

    std::vector<void*> tmp;  // device temporaries, freed on every exit
    auto dalloc = [&](void** p, size_t bytes) -> int {
        BG_HIP(hipMalloc(p, std::max<size_t>(bytes, 16)));
        tmp.push_back(*p);
        return BG_OK;
    };
    bg_fm* fm = nullptr;
    auto body = [&]() -> int {
        int rc;
        unsigned long long* d_hist = nullptr;
        if ((rc = dalloc((void**)&d_hist, 256 * 8))) return rc;
        BG_HIP(hipMemsetAsync(d_hist, 0, 256 * 8, st));
        fmb_hist_kernel<<<dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, st>>>(d_bwt, n, d_hist);
        uint64_t hist[256];
        BG_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        for (uint32_t c = m; c < 256; c++)
            if (hist[c]) return BG_ERR_OUT_OF_ALPHABET;  // Occ::new: curr_occ[c] out of bounds
        // less(bwt, alphabet) (bwt.rs:186-199) falls out of the histogram
        const uint32_t less_len = max_symbol + 2;
        std::vector<uint64_t> less(less_len, 0);
        {
            uint64_t acc = 0;
            for (uint32_t c = 0; c < less_len; c++) {
                less[c] = acc;
                if (c < 256) acc += hist[c];
            }
        }
        if (less_out) memcpy(less_out, less.data(), less_len * 8);
        FmClasses K;
        assign_classes(hist, in_alpha, K);
        ClsTab tab = {};
        std::vector<uint8_t> dense_byte(K.dense_syms.size());
        for (int c = 0; c < 256; c++) {
            tab.code[c] = K.code_of[c] >= 0 ? (uint8_t)K.code_of[c] : 0;
            tab.dense[c] = K.dense_of[c] >= 0 ? (uint8_t)(1 + K.dense_of[c]) : 0;
            tab.sparse[c] = K.sparse_of[c] >= 0 && hist[c] ? 1 : 0;
        }
        for (size_t d = 0; d < K.dense_syms.size(); d++) dense_byte[d] = (uint8_t)K.dense_syms[d];

        fm = new bg_fm;
        fm->ctx = ctx;
        fm->less_len = less_len;
        fm->fmd_ok = true;
        for (int c = 0; c < 256; c++)
            if (hist[c] && (c == 0 || !strchr("ACGTNacgtn$", c))) fm->fmd_ok = false;
        for (int c = 0; c < 256; c++)
            if (K.code_of[c] >= 0) fm->code_byte[K.code_of[c]] = (uint8_t)c;
        fm->n_codes = K.gen ? 3 : K.n_codes;

        const uint64_t nblk = (n + kSymPerBlock - 1) / kSymPerBlock, nbv = (n + kBvBits - 1) / kBvBits;
        const size_t n_dense = K.dense_syms.size();
        ClsTab* d_tab = nullptr;
        if ((rc = dalloc((void**)&d_tab, sizeof(ClsTab)))) return rc;
        BG_HIP(hipMemcpyAsync(d_tab, &tab, sizeof(tab), hipMemcpyHostToDevice, st));
        auto keep = [&](void** p, size_t bytes) -> int {  // device memory the handle owns
            const size_t alloc = std::max<size_t>(bytes, 16);
            BG_HIP(hipMalloc(p, alloc));
            fm->bytes += alloc;
            return BG_OK;
        };
        // ---- 2-bit blocks
        uint32_t *d_cnt = nullptr, *d_scan = nullptr;
        void* d_cub = nullptr;
        const uint64_t n_cnt = std::max<uint64_t>(4 * nblk, (uint64_t)n_dense * nbv);
        if ((rc = keep(&fm->d_blocks, nblk * 64))) return rc;
        if ((rc = dalloc((void**)&d_cnt, n_cnt * 4))) return rc;
        if ((rc = dalloc((void**)&d_scan, n_cnt * 4))) return rc;
        size_t cub_bytes = 0;
        BG_HIP(rocprim::exclusive_scan(nullptr, cub_bytes, d_cnt, d_scan, 0u, std::max<uint64_t>(nblk, nbv), rocprim::plus<uint32_t>(), st));
        if ((rc = dalloc(&d_cub, cub_bytes))) return rc;
        fmb_blocks_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(d_bwt, n, nblk, d_tab, (uint32_t*)fm->d_blocks, d_cnt);
        BG_HIP(hipGetLastError());
        for (int k = 0; k < 4; k++)
            BG_HIP(rocprim::exclusive_scan(d_cub, cub_bytes, d_cnt + (uint64_t)k * nblk, d_scan + (uint64_t)k * nblk, 0u, nblk, rocprim::plus<uint32_t>(), st));
        fmb_block_heads_kernel<<<dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st>>>(nblk, d_scan, (uint32_t*)fm->d_blocks);
        // ---- one-hot bit vectors of the dense symbols
        if ((rc = keep(&fm->d_bitvecs, n_dense * nbv * 64))) return rc;
        if (n_dense) {
            uint8_t* d_db = nullptr;
            if ((rc = dalloc((void**)&d_db, n_dense))) return rc;
            BG_HIP(hipMemcpyAsync(d_db, dense_byte.data(), n_dense, hipMemcpyHostToDevice, st));
            const uint64_t tot = (uint64_t)n_dense * nbv;
            fmb_bitvec_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(d_bwt, n, nbv, (uint32_t)n_dense, d_db,
                                                                                      (uint32_t*)fm->d_bitvecs, d_cnt);
            BG_HIP(hipGetLastError());
            for (size_t d = 0; d < n_dense; d++)
                BG_HIP(rocprim::exclusive_scan(d_cub, cub_bytes, d_cnt + d * nbv, d_scan + d * nbv, 0u, nbv, rocprim::plus<uint32_t>(), st));
            fmb_bitvec_heads_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(tot, d_scan, (uint32_t*)fm->d_bitvecs);
        }
        // ---- sparse exceptions (at most kMaxExcLds positions by construction of the classes)
        uint32_t* d_ns = nullptr;
        uint2* d_sp = nullptr;
        if ((rc = dalloc((void**)&d_ns, 4))) return rc;
        if ((rc = dalloc((void**)&d_sp, (size_t)(kMaxExcLds + 8) * 8))) return rc;
        BG_HIP(hipMemsetAsync(d_ns, 0, 4, st));
        fmb_sparse_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_bwt, n, d_tab, kMaxExcLds + 8, d_ns, d_sp);
        BG_HIP(hipGetLastError());
        uint32_t ns = 0;
        BG_HIP(hipMemcpyAsync(&ns, d_ns, 4, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        if (ns > kMaxExcLds) return BG_ERR_HIP;  // cannot happen: the classes were cut so that they fit
        std::vector<uint2> sp(ns);
        if (ns) BG_HIP(hipMemcpy(sp.data(), d_sp, (size_t)ns * 8, hipMemcpyDeviceToHost));
        std::sort(sp.begin(), sp.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
        std::vector<uint32_t> exc_pos(ns), exc_sym_pos, sparse_off(K.sparse_syms.size() + 1, 0);
        std::vector<uint8_t> exc_byte(ns);
        for (uint32_t e = 0; e < ns; e++) {
            exc_pos[e] = sp[e].x;
            exc_byte[e] = (uint8_t)sp[e].y;
        }
        for (size_t e = 0; e < K.sparse_syms.size(); e++) {
            for (uint32_t q = 0; q < ns; q++)
                if ((int)sp[q].y == K.sparse_syms[e]) exc_sym_pos.push_back(sp[q].x);
            sparse_off[e + 1] = (uint32_t)exc_sym_pos.size();
        }
        uint32_t less32[256] = {};
        for (uint32_t i = 0; i < less_len && i < 256; i++) less32[i] = (uint32_t)less[i];
        auto upload = [&](void** dptr, const void* src, size_t bytes) -> int {
            int r2 = keep(dptr, bytes);
            if (r2) return r2;
            if (bytes) BG_HIP(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
            return BG_OK;
        };
        if ((rc = upload(&fm->d_exc_pos, exc_pos.data(), exc_pos.size() * 4))) return rc;
        if ((rc = upload(&fm->d_exc_sym_pos, exc_sym_pos.data(), exc_sym_pos.size() * 4))) return rc;
        if ((rc = upload(&fm->d_sparse_off, sparse_off.data(), sparse_off.size() * 4))) return rc;
        if ((rc = upload(&fm->d_exc_byte, exc_byte.data(), exc_byte.size()))) return rc;
        if ((rc = upload(&fm->d_class, K.cls, 256 * sizeof(uint16_t)))) return rc;
        memcpy(fm->h_class, K.cls, sizeof(fm->h_class));
        if ((rc = upload(&fm->d_less, less32, sizeof(less32)))) return rc;
        if (K.gen) {
            if ((rc = keep(&fm->d_bwt_raw, n))) return rc;
            BG_HIP(hipMemcpyAsync(fm->d_bwt_raw, d_bwt, n, hipMemcpyDeviceToDevice, st));
        }
        BG_HIP(hipStreamSynchronize(st));
        fm->dev.blocks = (const uint4*)fm->d_blocks;
        fm->dev.bitvecs = (const uint4*)fm->d_bitvecs;
        fm->dev.exc_pos = (const uint32_t*)fm->d_exc_pos;
        fm->dev.exc_sym_pos = (const uint32_t*)fm->d_exc_sym_pos;
        fm->dev.sparse_off = (const uint32_t*)fm->d_sparse_off;
        fm->dev.sym_class = (const uint16_t*)fm->d_class;
        fm->dev.less = (const uint32_t*)fm->d_less;
        fm->dev.bwt_raw = (const uint8_t*)fm->d_bwt_raw;
        fm->dev.n = (uint32_t)n;
        fm->dev.n_exc = K.gen ? 0u : ns;
        fm->dev.nbv_blocks = (uint32_t)nbv;
        fm->dev.n_dense = (uint32_t)n_dense;
        return BG_OK;
    };
    const int rc = body();
    for (void* p : tmp) hipFree(p);
    if (rc) {
        bg_fm_free(fm);
        return rc;
    }
    fm_build_step2(fm, st);  // 2-step rank blocks (fm_step2.hip): less[] is this builder's own
    *out = fm;
    return BG_OK;
}

// the exported builders: the implementations above + what bg_fm_save needs to write the index out again (fm_persist.hip)
extern "C" int bg_fm_build(bg_ctx* ctx, const uint8_t* bwt, uint64_t n, const uint64_t* less, uint32_t less_len, uint32_t occ_k,
                           const uint8_t* alphabet, uint32_t n_sym, bg_fm** out) {
    const int rc = fm_build_host_impl(ctx, bwt, n, less, less_len, occ_k, alphabet, n_sym, out);
    if (rc == BG_OK && out && *out) fm_remember_inputs(*out, alphabet, n_sym, occ_k, less, less_len);
    return rc;
}
extern "C" int bg_fm_build_dev(bg_ctx* ctx, const uint8_t* d_bwt, uint64_t n, uint32_t occ_k, const uint8_t* alphabet, uint32_t n_sym,
                               uint64_t* less_out, bg_fm** out, void* stream) {
    uint32_t max_symbol = 0;
    for (uint32_t i = 0; alphabet && i < n_sym; i++) max_symbol = std::max<uint32_t>(max_symbol, alphabet[i]);
    std::vector<uint64_t> less(max_symbol + 2, 0);
    const int rc = fm_build_dev_impl(ctx, d_bwt, n, occ_k, alphabet, n_sym, less.data(), out, stream);
    if (rc == BG_OK && out && *out) {
        fm_remember_inputs(*out, alphabet, n_sym, occ_k, less.data(), (uint32_t)less.size());
        if (less_out) memcpy(less_out, less.data(), less.size() * 8);
    }
    return rc;
}

extern "C" int bg_fm_free(bg_fm* fm) {
    if (!fm) return BG_OK;
    hipFree(fm->d_blocks);
    hipFree(fm->d_sb);
    hipFree(fm->d_sb2);
    hipFree(fm->d_blocks2);
    hipFree(fm->d_exc_pos);
    hipFree(fm->d_exc_sym_pos);
    hipFree(fm->d_class);
    hipFree(fm->d_less);
    hipFree(fm->d_exc_byte);
    hipFree(fm->d_sparse_off);
    hipFree(fm->d_bitvecs);
    hipFree(fm->d_bwt_raw);
    if (fm->text_owned) hipFree(fm->d_text);
    hipFree(fm->d_jump);
    hipFree(fm->d_sa);
    hipFree(fm->d_extra_row);
    hipFree(fm->d_extra_pos);
    delete fm;
    return BG_OK;
}

extern "C" uint64_t bg_fm_device_bytes(const bg_fm* fm) { return fm ? fm->bytes : 0; }
extern "C" uint64_t bg_fm_step2_bytes(const bg_fm* fm) {
    if (fm && fm->wide) return fm->wdev2.blocks2 && !fm->no_step2 ? (fm->wdev.n + kSym2PerBlock - 1) / kSym2PerBlock * 128 : 0;
    return fm && fm->dev2.blocks2 && !fm->no_step2 ? ((uint64_t)fm->dev.n + kSym2PerBlock - 1) / kSym2PerBlock * 128 : 0;
}

// the four 2-bit codes all stand for symbols and no symbol is ranked by a bit vector: the packed / fast kernels apply
static bool fm_fast_ok(const bg_fm* fm) { return !fm->dev.n_dense && fm->n_codes == 4 && !fm->no_fast; }
// ... and the index has 2-step rank blocks (fm_step2.hip): the fast kernels take two symbols per block access
static bool fm_step2_ok(const bg_fm* fm) { return fm->dev2.blocks2 != nullptr && !fm->no_step2; }
// grid of fm_search_fast2x_kernel: persistent blocks, as many as are resident (its registers decide), 128 queries each
template <typename K>
static uint64_t fm_2x_blocks(K kernel, uint64_t n_q, int u) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    return std::min<uint64_t>((n_q + 64 * (uint64_t)u - 1) / (64 * (uint64_t)u), 256ull * (uint64_t)per_cu);
}
#define FM_LAUNCH_2X(SEEDS, COUNT, PACKED, ...)                                                                                  \
    fm_search_fast2x_kernel<SEEDS, COUNT, PACKED, 2><<<dim3((unsigned)fm_2x_blocks(fm_search_fast2x_kernel<SEEDS, COUNT, PACKED, 2>, n_q, 2)), \
                                                       dim3(256), 0, st>>>(__VA_ARGS__)
#define FM_LAUNCH_2XW(SEEDS, PACKED, ...)                                                                                                    \
    fm_search_fast2x_kernel<SEEDS, false, PACKED, 2, true><<<dim3((unsigned)fm_2x_blocks(fm_search_fast2x_kernel<SEEDS, false, PACKED, 2, true>, n_q, 2)), \
                                                            dim3(256), 0, st>>>(__VA_ARGS__)
static SeedSrc fm_codes(const bg_fm* fm) {
    SeedSrc ex{};
    ex.code_bytes = (uint32_t)fm->code_byte[0] | (uint32_t)fm->code_byte[1] << 8 | (uint32_t)fm->code_byte[2] << 16 |
                    (uint32_t)fm->code_byte[3] << 24;
    return ex;
}

// the 2x fast kernel on 64-bit positions (fm_wide.hip decides when; queries it cannot hold are left tagged kTagDeferred)
int fm_wide_fast2x_launch(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat, const uint64_t* d_pat_off, uint8_t* d_tag, uint64_t* d_lower,
                          uint64_t* d_upper, uint32_t* d_matched_len, hipStream_t st, const SeedSrc* seeds, bool packed) {
    SeedSrc src = fm_codes(fm);
    if (seeds) src.S = seeds->S, src.stride = seeds->stride, src.seed_len = seeds->seed_len;
    if (seeds)
        FM_LAUNCH_2XW(true, false, fm->wdev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, fm->wdev2);
    else if (packed)
        FM_LAUNCH_2XW(false, true, fm->wdev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, fm->wdev2);
    else
        FM_LAUNCH_2XW(false, false, fm->wdev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, src, fm->wdev2);
    BG_HIP(hipGetLastError());
    return BG_OK;
}

extern "C" int bg_fm_set_option(bg_fm* fm, const char* key, int64_t value) {
    if (!fm || !key) return BG_ERR_INVALID_ARG;
    if (!strcmp(key, "no_step2")) {  // searches take single steps only, whether the index has 2-step blocks or not (tests, A/B)
        fm->no_step2 = value != 0;
        return BG_OK;
    }
    if (!strcmp(key, "ilp")) {  // queries per quad of the 2-step search: 1 fm_search_fast_kernel, 2 fm_search_fast2x_kernel (A/B)
        if (value != 1 && value != 2) return BG_ERR_INVALID_ARG;
        fm->ilp = (int)value;
        return BG_OK;
    }
    if (!strcmp(key, "jump_min_queries")) {  // batch size from which K5 builds / uses its jump table; < 0: never (default)
        std::lock_guard<std::mutex> lk(fm->jump_mu);
        fm->no_jump = value < 0;
        fm->jump_min_queries = value < 0 ? ~0ull : (uint64_t)value;
        if (fm->no_jump && fm->d_jump) {
            hipFree(fm->d_jump);
            fm->d_jump = nullptr;
        }
        return BG_OK;
    }
    if (!strcmp(key, "no_fast")) {  // tests: every search through the generic kernel
        fm->no_fast = value != 0;
        return BG_OK;
    }
    return BG_ERR_INVALID_ARG;
}

extern "C" int bg_fm_backward_search_batch_dev(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat,
                                               const uint64_t* d_pat_off, uint8_t* d_tag,
                                               uint64_t* d_lower, uint64_t* d_upper,
                                               uint32_t* d_matched_len, void* stream) {
    if (!fm || (n_q && (!d_pat_off || !d_tag || !d_lower || !d_upper || !d_matched_len)))
        return BG_ERR_INVALID_ARG;
    if (n_q == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    hipStream_t st = (hipStream_t)stream;
    if (fm->wide) {  // 64-bit positions: fm_wide.hip
        if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st));
        const int rcw = fm_wide_search_dev(fm, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, st);
        if (rcw) return rcw;
        if (ctx->timing) {
            BG_HIP(hipEventRecord(ctx->ev[1], st));
            BG_HIP(hipEventSynchronize(ctx->ev[1]));
            float ms = 0;
            BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            ctx->last.fm_ms += ms;
            ctx->last.fm_launches += 1;
        }
        return BG_OK;
    }
    const uint64_t quads_per_block = 64;
    uint64_t blocks = (n_q + quads_per_block - 1) / quads_per_block;
    blocks = std::min<uint64_t>(blocks, 256 * 8);  // 8 resident 256-thread blocks per CU
    // the jump table pays off from a few million LF steps on; it is built once per index, by the search itself
    const uint4* jump = nullptr;
    if (!fm->no_jump && n_q >= fm->jump_min_queries && fm->n_codes == 4) {
      std::lock_guard<std::mutex> lk(fm->jump_mu);
      if (!fm->d_jump && !fm->no_jump) {
        const uint64_t nk = 1ull << (2 * kJumpK);
        void* d_table = nullptr;
        uint8_t *t_pat = nullptr, *t_tag = nullptr;
        uint64_t *t_off = nullptr, *t_lo = nullptr, *t_hi = nullptr;
        uint32_t* t_ml = nullptr;
        auto build = [&]() -> int {
            BG_HIP(hipMalloc((void**)&t_pat, nk * kJumpK));
            BG_HIP(hipMalloc((void**)&t_off, (nk + 1) * 8));
            BG_HIP(hipMalloc((void**)&t_tag, nk));
            BG_HIP(hipMalloc((void**)&t_lo, nk * 8));
            BG_HIP(hipMalloc((void**)&t_hi, nk * 8));
            BG_HIP(hipMalloc((void**)&t_ml, nk * 4));
            BG_HIP(hipMalloc(&d_table, nk * sizeof(uint4)));
            const uint32_t cb = (uint32_t)fm->code_byte[0] | (uint32_t)fm->code_byte[1] << 8 | (uint32_t)fm->code_byte[2] << 16 |
                                (uint32_t)fm->code_byte[3] << 24;
            fm_jump_patterns_kernel<<<dim3((unsigned)((nk + 256) / 256)), dim3(256), 0, st>>>(cb, t_pat, t_off);
            fm_backward_search_kernel<false, false><<<dim3(256 * 8), dim3(256), 0, st>>>(fm->dev, nk, t_pat, t_off, t_tag, t_lo, t_hi, t_ml, nullptr, SeedSrc{});
            fm_jump_pack_kernel<<<dim3((unsigned)(nk / 256)), dim3(256), 0, st>>>(t_tag, t_lo, t_hi, t_ml, (uint4*)d_table);
            BG_HIP(hipGetLastError());
            BG_HIP(hipStreamSynchronize(st));
            fm->bytes += nk * sizeof(uint4);
            return BG_OK;
        };
        const int rcj = build();
        hipFree(t_pat);
        hipFree(t_off);
        hipFree(t_tag);
        hipFree(t_lo);
        hipFree(t_hi);
        hipFree(t_ml);
        if (rcj) {  // no memory for the table: search without it
            hipFree(d_table);
            fm->no_jump = true;
        } else {
            fm->d_jump = d_table;  // published complete
        }
      }
      jump = (const uint4*)fm->d_jump;
    }
    if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st));
    if (jump) {
        fm_backward_search_kernel<true, false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
            fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, jump, SeedSrc{});
    } else if (fm_fast_ok(fm)) {
        // DNA-like index: patterns become 2-bit codes in LDS when a quad takes them; what that path cannot hold (a byte
        // outside the four codes, more than kFastSyms symbols) is left to the generic kernel behind it
        if (fm_step2_ok(fm) && fm->ilp >= 2)
            FM_LAUNCH_2X(false, false, false, fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, fm_codes(fm), fm->dev2);
        else if (fm_step2_ok(fm))
            fm_search_fast_kernel<false, false, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
                fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, fm_codes(fm), fm->dev2);
        else
            fm_search_fast_kernel<false, false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
                fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, fm_codes(fm), fm->dev2);
        fm_backward_search_kernel<false, false, false, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
            fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, SeedSrc{});
    } else {
        fm_backward_search_kernel<false, false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
            fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, SeedSrc{});
    }
    BG_HIP(hipGetLastError());
    if (ctx->timing) {
        BG_HIP(hipEventRecord(ctx->ev[1], st));
        BG_HIP(hipEventSynchronize(ctx->ev[1]));
        float ms = 0;
        BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ctx->last.fm_ms += ms;
        ctx->last.fm_launches += 1;
    }
    return BG_OK;
}

// the seed windows of a batch of reads as patterns (see SeedSrc): n_reads * S queries, results indexed [read * S + seed]
int bg_fm_search_seeds_dev(bg_fm* fm, uint64_t n_reads, const uint8_t* d_reads, const uint64_t* d_read_off, uint32_t S,
                           uint32_t stride, uint32_t seed_len, uint8_t* d_tag, uint64_t* d_lower, uint64_t* d_upper,
                           uint32_t* d_matched_len, hipStream_t st) {
    const uint64_t n_q = n_reads * S;
    if (n_q == 0) return BG_OK;
    if (fm->wide) {  // 64-bit positions: fm_wide.hip
        SeedSrc src{};
        src.S = S, src.stride = stride, src.seed_len = seed_len;
        bg_ctx* cx = fm->ctx;
        if (cx && cx->timing) BG_HIP(hipEventRecord(cx->ev[0], st));
        const int rcw = fm_wide_search_dev(fm, n_q, d_reads, d_read_off, d_tag, d_lower, d_upper, d_matched_len, st, &src, false);
        if (rcw) return rcw;
        if (cx && cx->timing) {
            BG_HIP(hipEventRecord(cx->ev[1], st));
            BG_HIP(hipEventSynchronize(cx->ev[1]));
            float ms = 0;
            BG_HIP(hipEventElapsedTime(&ms, cx->ev[0], cx->ev[1]));
            cx->last.fm_ms += ms;
            cx->last.fm_launches += 1;
        }
        return BG_OK;
    }
    const uint64_t blocks = std::min<uint64_t>((n_q + 63) / 64, 256 * 8);
    SeedSrc src = fm_codes(fm);
    src.S = S, src.stride = stride, src.seed_len = seed_len;
    bg_ctx* ctx = fm->ctx;
    if (ctx && ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st));
    if (fm_fast_ok(fm) && seed_len <= kFastSyms) {
        if (fm_step2_ok(fm) && fm->ilp >= 2)
            FM_LAUNCH_2X(true, false, false, fm->dev, n_q, d_reads, d_read_off, d_tag, d_lower, d_upper, d_matched_len, src, fm->dev2);
        else if (fm_step2_ok(fm))
            fm_search_fast_kernel<true, false, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->dev, n_q, d_reads, d_read_off, d_tag,
                                                                                                         d_lower, d_upper, d_matched_len, src, fm->dev2);
        else
            fm_search_fast_kernel<true, false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->dev, n_q, d_reads, d_read_off, d_tag, d_lower,
                                                                                            d_upper, d_matched_len, src, fm->dev2);
        fm_backward_search_kernel<false, true, false, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
            fm->dev, n_q, d_reads, d_read_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, src);
    } else {
        fm_backward_search_kernel<false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
            fm->dev, n_q, d_reads, d_read_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, src);
    }
    BG_HIP(hipGetLastError());
    if (ctx && ctx->timing) {  // (the seed search inside bg_seed_extend_batch_dev: kernel_ms.seed_search of bench.py)
        BG_HIP(hipEventRecord(ctx->ev[1], st));
        BG_HIP(hipEventSynchronize(ctx->ev[1]));
        float ms = 0;
        BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ctx->last.fm_ms += ms;
        ctx->last.fm_launches += 1;
    }
    return BG_OK;
}

// ---- 2-bit packed patterns (north_star: "coalesced HBM loads of packed 2-bit reads") -------------------------------
extern "C" int bg_fm_pattern_codes(const bg_fm* fm, uint8_t codes[4]) {
    if (!fm || !codes) return BG_ERR_INVALID_ARG;
    // the four 2-bit codes must all stand for symbols of the text (a DNA-like BWT): with dense symbols code 0 means
    // "something else", and an index over fewer than four letters has codes no pattern symbol may use
    if ((!fm->wide && fm->dev.n_dense) || fm->n_codes != 4) return BG_ERR_UNSUPPORTED;
    for (int c = 0; c < 4; c++) codes[c] = fm->code_byte[c];
    return BG_OK;
}

extern "C" int bg_fm_backward_search_packed_dev(bg_fm* fm, uint64_t n_q, const uint32_t* d_packed, const uint64_t* d_sym_off,
                                                uint8_t* d_tag, uint64_t* d_lower, uint64_t* d_upper, uint32_t* d_matched_len,
                                                void* stream) {
    if (!fm || (n_q && (!d_packed || !d_sym_off || !d_tag || !d_lower || !d_upper || !d_matched_len))) return BG_ERR_INVALID_ARG;
    if ((!fm->wide && fm->dev.n_dense) || fm->n_codes != 4) return BG_ERR_UNSUPPORTED;
    if (n_q == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t blocks = std::min<uint64_t>((n_q + 63) / 64, 256 * 8);
    const SeedSrc ex = fm_codes(fm);
    if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st));
    if (fm->wide) {  // 64-bit positions: fm_wide.hip
        const int rcw = fm_wide_search_dev(fm, n_q, (const uint8_t*)d_packed, d_sym_off, d_tag, d_lower, d_upper, d_matched_len, st, nullptr, true);
        if (rcw) return rcw;
    } else if (!fm->no_fast) {  // the LDS-slot kernel; patterns beyond its 256 symbols are left to the generic packed kernel
        if (fm_step2_ok(fm) && fm->ilp >= 2)
            FM_LAUNCH_2X(false, false, true, fm->dev, n_q, (const uint8_t*)d_packed, d_sym_off, d_tag, d_lower, d_upper, d_matched_len, ex, fm->dev2);
        else if (fm_step2_ok(fm))
            fm_search_fast_kernel<false, false, true, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
                fm->dev, n_q, (const uint8_t*)d_packed, d_sym_off, d_tag, d_lower, d_upper, d_matched_len, ex, fm->dev2);
        else
            fm_search_fast_kernel<false, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
                fm->dev, n_q, (const uint8_t*)d_packed, d_sym_off, d_tag, d_lower, d_upper, d_matched_len, ex, fm->dev2);
        fm_backward_search_kernel<false, false, true, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
            fm->dev, n_q, (const uint8_t*)d_packed, d_sym_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, ex);
    } else {
        fm_backward_search_kernel<false, false, true, false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
            fm->dev, n_q, (const uint8_t*)d_packed, d_sym_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, ex);
    }
    BG_HIP(hipGetLastError());
    if (ctx->timing) {
        BG_HIP(hipEventRecord(ctx->ev[1], st));
        BG_HIP(hipEventSynchronize(ctx->ev[1]));
        float ms = 0;
        BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ctx->last.fm_ms += ms;
        ctx->last.fm_launches += 1;
    }
    return BG_OK;
}

// measurement aid: the same search with its 64-byte block loads counted (synchronous; *lines_out on the host)
extern "C" int bg_fm_backward_search_count_lines_dev(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat, const uint64_t* d_pat_off,
                                                     uint8_t* d_tag, uint64_t* d_lower, uint64_t* d_upper,
                                                     uint32_t* d_matched_len, uint64_t* lines_out, void* stream) {
    if (!fm || !lines_out || (n_q && (!d_pat_off || !d_tag || !d_lower || !d_upper || !d_matched_len))) return BG_ERR_INVALID_ARG;
    *lines_out = 0;
    if (fm->wide) return BG_ERR_UNSUPPORTED;  // (a measurement aid of the 32-bit kernels)
    if (n_q == 0) return BG_OK;
    hipStream_t st = (hipStream_t)stream;
    BG_HIP(hipSetDevice(fm->ctx->device));
    unsigned long long* d_cnt = nullptr;
    BG_HIP(hipMalloc((void**)&d_cnt, 8));
    int rc = BG_OK;
    auto run = [&]() -> int {
        BG_HIP(hipMemsetAsync(d_cnt, 0, 8, st));
        const uint64_t blocks = std::min<uint64_t>((n_q + 63) / 64, 256 * 8);
        SeedSrc ex = fm_codes(fm);
        ex.lines = d_cnt;
        if (fm_fast_ok(fm)) {
            if (fm_step2_ok(fm) && fm->ilp >= 2)
            FM_LAUNCH_2X(false, true, false, fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, ex, fm->dev2);
            else if (fm_step2_ok(fm))
                fm_search_fast_kernel<false, true, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->dev, n_q, d_pat, d_pat_off, d_tag,
                                                                                                            d_lower, d_upper, d_matched_len, ex, fm->dev2);
            else
                fm_search_fast_kernel<false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(fm->dev, n_q, d_pat, d_pat_off, d_tag,
                                                                                               d_lower, d_upper, d_matched_len, ex, fm->dev2);
            fm_backward_search_kernel<false, false, false, true, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
                fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, ex);
        } else {
            fm_backward_search_kernel<false, false, false, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(
                fm->dev, n_q, d_pat, d_pat_off, d_tag, d_lower, d_upper, d_matched_len, nullptr, ex);
        }
        BG_HIP(hipGetLastError());
        BG_HIP(hipMemcpyAsync(lines_out, d_cnt, 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    };
    rc = run();
    hipFree(d_cnt);
    return rc;
}

// ---- bg_fm_backward_search_batch: host buffers in and out -----------------------------------------------------------
// Round 2 made six hipMallocs and pageable copies per call (150 M queries/s against 460 M device-resident).  Now the
// batch flows in stages of kFmStage queries through three staging sets that persist in the ctx: host threads copy a
// stage's pattern bytes into pinned memory and rebase its offsets, the copy-in stream uploads them, the ctx stream
// searches, the copy-out stream downloads the four result arrays into pinned memory, and a second host thread copies
// them into the caller's arrays while the following stages are in flight — no allocation in a warmed-up call.
struct bg_fm_pipe {
    static constexpr int NSET = 3;
    struct Set {
        uint8_t *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        hipEvent_t in_done = nullptr, k_done = nullptr, out_done = nullptr;
    } set[NSET];
    hipStream_t s_in = nullptr, s_out = nullptr;
};
void bg_fm_pipe_free(bg_fm_pipe* p) {
    if (!p) return;
    for (auto& s : p->set) {
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        hipFree(s.d_in);
        hipFree(s.d_out);
        for (hipEvent_t e : {s.in_done, s.k_done, s.out_done})
            if (e) hipEventDestroy(e);
    }
    if (p->s_in) hipStreamDestroy(p->s_in);
    if (p->s_out) hipStreamDestroy(p->s_out);
    delete p;
}
namespace {
template <typename F>
void fm_parallel_for(uint64_t n, uint64_t grain, F&& f) {
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(bg_host_threads(), n / std::max<uint64_t>(grain, 1) + 1));
    if (nt == 1) {
        f((uint64_t)0, n);
        return;
    }
    bg_pool_run(nt, [&](unsigned t) { f(n * t / nt, n * (t + 1) / nt); });
}
constexpr uint64_t kFmStage = 1u << 20;  // queries per stage
}  // namespace

extern "C" int bg_fm_backward_search_batch(bg_fm* fm, uint64_t n_q, const uint8_t* pat,
                                           const uint64_t* pat_off, uint8_t* tag, uint64_t* lower,
                                           uint64_t* upper, uint32_t* matched_len) {
    if (!fm || (n_q && (!pat_off || !tag || !lower || !upper || !matched_len)))
        return BG_ERR_INVALID_ARG;
    if (n_q == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    if (pat_off[n_q] && !pat) return BG_ERR_INVALID_ARG;
    if (!ctx->fm_pipe) {
        ctx->fm_pipe = new bg_fm_pipe();
        BG_HIP(hipStreamCreateWithFlags(&ctx->fm_pipe->s_in, hipStreamNonBlocking));
        BG_HIP(hipStreamCreateWithFlags(&ctx->fm_pipe->s_out, hipStreamNonBlocking));
        for (auto& s : ctx->fm_pipe->set)
            for (hipEvent_t* e : {&s.in_done, &s.k_done, &s.out_done}) BG_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    bg_fm_pipe& P = *ctx->fm_pipe;
    const uint64_t chunk = std::min<uint64_t>(kFmStage, n_q);
    const uint64_t nch = (n_q + chunk - 1) / chunk;
    uint64_t max_pb = 0;
    for (uint64_t c = 0; c < nch; c++) {
        const uint64_t q0 = c * chunk, q1 = std::min(n_q, q0 + chunk);
        if (pat_off[q1] < pat_off[q0]) return BG_ERR_INVALID_ARG;
        max_pb = std::max(max_pb, pat_off[q1] - pat_off[q0]);
    }
    // staging layout: in = pattern bytes | offsets ; out = lower | upper | matched_len | tag
    const uint64_t o_off = (max_pb + 8 + 255) & ~255ull;  // (+8: the packed flavour's dword past the last symbol)
    const size_t in_need = o_off + (chunk + 1) * 8 + 256;
    const uint64_t o_hi = chunk * 8, o_ml = 2 * chunk * 8, o_tag = o_ml + chunk * 4;
    const size_t out_need = o_tag + chunk + 256;
    for (auto& s : P.set) {
        if (s.in_cap < in_need) {
            if (s.h_in) hipHostFree(s.h_in);
            hipFree(s.d_in);
            s.h_in = s.d_in = nullptr;
            s.in_cap = 0;
            BG_HIP(hipHostMalloc((void**)&s.h_in, in_need, hipHostMallocDefault));
            BG_HIP(hipMalloc((void**)&s.d_in, in_need));
            s.in_cap = in_need;
        }
        if (s.out_cap < out_need) {
            if (s.h_out) hipHostFree(s.h_out);
            hipFree(s.d_out);
            s.h_out = s.d_out = nullptr;
            s.out_cap = 0;
            BG_HIP(hipHostMalloc((void**)&s.h_out, out_need, hipHostMallocDefault));
            BG_HIP(hipMalloc((void**)&s.d_out, out_need));
            s.out_cap = out_need;
        }
    }
    hipStream_t s_k = ctx->stream;
    std::atomic<bool> any_panic{false};
    // The patterns travel as 2-bit codes where the index takes them (bg_fm_pattern_codes) — packed by the worker threads
    // into the pinned stage instead of copied there: a quarter of the bytes to write and to send over the link, and the
    // packed kernel on the other side.  A stage with a byte outside the four codes is staged as bytes after all.
    uint8_t pcodes[4] = {0, 0, 0, 0};
    const bool can_pack = !ctx->fm_host_bytes && bg_fm_pattern_codes(fm, pcodes) == BG_OK;
    uint64_t n_packed_stages = 0;
    // BG_TRACE_HOST=1: where the host side of the stages spends its time (ms, summed over the call)
    const bool trace = getenv("BG_TRACE_HOST") != nullptr;
    double t_pack = 0, t_launch = 0, t_wait_set = 0, t_d_wait = 0, t_d_copy = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto drain = [&](uint64_t c) -> int {
        bg_fm_pipe::Set& S = P.set[c % bg_fm_pipe::NSET];
        double t0 = now();
        BG_HIP(hipEventSynchronize(S.out_done));
        t_d_wait += now() - t0;
        t0 = now();
        const uint64_t q0 = c * chunk, nq = std::min(n_q, q0 + chunk) - q0;
        fm_parallel_for(nq, 1 << 16, [&](uint64_t a, uint64_t b) {
            memcpy(lower + q0 + a, S.h_out + a * 8, (b - a) * 8);
            memcpy(upper + q0 + a, S.h_out + o_hi + a * 8, (b - a) * 8);
            memcpy(matched_len + q0 + a, S.h_out + o_ml + a * 4, (b - a) * 4);
            memcpy(tag + q0 + a, S.h_out + o_tag + a, b - a);
            bool pn = false;
            for (uint64_t q = a; q < b; q++) pn = pn || S.h_out[o_tag + q] == BG_FM_PANIC;
            if (pn) any_panic = true;
        });
        t_d_copy += now() - t0;
        return BG_OK;
    };
    std::mutex mu;
    std::condition_variable cv;
    uint64_t submitted = 0, drained = 0;
    bool abort_drain = false;
    int drain_rc = BG_OK;
    std::string drain_err;
    const int device = ctx->device;
    std::thread drainer([&] {
        if (hipSetDevice(device) != hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            drain_rc = BG_ERR_HIP;
            drained = nch;
            cv.notify_all();
            return;
        }
        for (uint64_t c = 0; c < nch; c++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return submitted > c || abort_drain; });
                if (submitted <= c) break;
            }
            const int r = drain(c);
            std::lock_guard<std::mutex> lk(mu);
            if (r && drain_rc == BG_OK) {
                drain_rc = r;
                drain_err = bg_tls_error;
            }
            drained = c + 1;
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        drained = nch;
        cv.notify_all();
    });
    auto stop_drainer = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            abort_drain = true;
        }
        cv.notify_all();
        drainer.join();
    };
    for (uint64_t c = 0; c < nch; c++) {
        double t0 = now();
        if (c >= bg_fm_pipe::NSET) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return drained >= c - bg_fm_pipe::NSET + 1; });
        }
        t_wait_set += now() - t0;
        t0 = now();
        bg_fm_pipe::Set& S = P.set[c % bg_fm_pipe::NSET];
        const uint64_t q0 = c * chunk, nq = std::min(n_q, q0 + chunk) - q0;
        const uint64_t b0 = pat_off[q0], pb = pat_off[q0 + nq] - b0;
        bool packed = can_pack && pb != 0;
        if (packed) {
            std::atomic<bool> all_codes{true};
            // (ranges start at multiples of 64 symbols: whole dwords per thread)
            fm_parallel_for((pb + 63) / 64, 1 << 14, [&](uint64_t a, uint64_t b) {
                const uint64_t s0 = a * 64, s1 = std::min(pb, b * 64);
                if (!bgpack::pack2_host(pat + b0 + s0, s1 - s0, pcodes, (uint32_t*)S.h_in + s0 / 16)) all_codes = false;
            });
            packed = all_codes;
        }
        if (!packed) fm_parallel_for(pb, 1 << 20, [&](uint64_t a, uint64_t b) { memcpy(S.h_in + a, pat + b0 + a, b - a); });
        n_packed_stages += packed ? 1 : 0;
        const uint64_t in_bytes = packed ? ((pb + 15) / 16 + 1) * 4 : pb;
        uint64_t* hoff = (uint64_t*)(S.h_in + o_off);
        fm_parallel_for(nq + 1, 1 << 16, [&](uint64_t a, uint64_t b) {
            for (uint64_t q = a; q < b; q++) hoff[q] = pat_off[q0 + q] - b0;
        });
        t_pack += now() - t0;
        t0 = now();
        bool ok = true;
        if (pb) ok = ok && bg_copy_pieces(S.d_in, S.h_in, in_bytes, hipMemcpyHostToDevice, P.s_in) == hipSuccess;
        ok = ok && hipMemcpyAsync(S.d_in + o_off, S.h_in + o_off, (nq + 1) * 8, hipMemcpyHostToDevice, P.s_in) == hipSuccess;
        ok = ok && hipEventRecord(S.in_done, P.s_in) == hipSuccess && hipStreamWaitEvent(s_k, S.in_done, 0) == hipSuccess;
        int rc = !ok ? BG_ERR_HIP
                 : packed
                     ? bg_fm_backward_search_packed_dev(fm, nq, (const uint32_t*)S.d_in, (const uint64_t*)(S.d_in + o_off), S.d_out + o_tag,
                                                        (uint64_t*)S.d_out, (uint64_t*)(S.d_out + o_hi), (uint32_t*)(S.d_out + o_ml), s_k)
                     : bg_fm_backward_search_batch_dev(fm, nq, S.d_in, (const uint64_t*)(S.d_in + o_off), S.d_out + o_tag,
                                                       (uint64_t*)S.d_out, (uint64_t*)(S.d_out + o_hi), (uint32_t*)(S.d_out + o_ml), s_k);
        if (rc == BG_OK && (hipEventRecord(S.k_done, s_k) != hipSuccess || hipStreamWaitEvent(P.s_out, S.k_done, 0) != hipSuccess ||
                            bg_copy_pieces(S.h_out, S.d_out, o_tag + nq, hipMemcpyDeviceToHost, P.s_out) != hipSuccess ||
                            hipEventRecord(S.out_done, P.s_out) != hipSuccess))
            rc = BG_ERR_HIP;
        if (rc) {
            hipDeviceSynchronize();
            stop_drainer();
            return rc;
        }
        t_launch += now() - t0;
        {
            std::lock_guard<std::mutex> lk(mu);
            submitted = c + 1;
        }
        cv.notify_all();
    }
    const double t0j = now();
    drainer.join();
    if (trace)
        fprintf(stderr, "[bg fm host] %llu stages (%llu as 2-bit codes): pack %.2f launch %.2f wait-for-set %.2f join %.2f | drainer: wait %.2f copy %.2f ms\n",
                (unsigned long long)nch, (unsigned long long)n_packed_stages, t_pack, t_launch, t_wait_set, now() - t0j, t_d_wait, t_d_copy);
    if (drain_rc) {
        bg_tls_error = drain_err;
        return drain_rc;
    }
    return any_panic ? BG_ERR_OUT_OF_ALPHABET : BG_OK;
}
