// Seed-and-extend read mapping behind the C ABI (bg_seed_extend_batch[_dev]; BASELINE configs[4]).
//
// rust-bio has no read mapper: its callers compose one out of three calls
// (/root/reference/src/lib.rs:129-165, benches/fmindex.rs:20-38):
//     FMIndex::backward_search(seed)        fmindex.rs:144-208      K5 (fm_index.hip), the seed windows read in place
//     Interval::occ(&suffix_array)          fmindex.rs:75-79        K6 (sa_locate.hip), raw or sampled suffix array
//     Aligner::semiglobal(read, window)     pairwise/mod.rs:954     K1p / K1 + K2 (sw_*.hip)
// This file is the glue between them, all of it on the device: which seeds vote, hit -> proposed read start, the
// per-read sort + dedup of the proposals, the gather of the (read, window) pairs the aligner consumes, and the
// best-hit reduction that hands back one alignment (with its operations) per read.  Definition of the
// composition: include/biogpu.h (the tests hold a CPU statement of the same thing).
//
// Per batch of reads (S = seed slots per read):
//   S1 K5<SEEDS>        n_reads * S backward searches                                  -> tag, lower, upper
//   S2 votes            cnt[q] = interval size if Complete and 1 <= size <= max_occ     -> scan -> hit offsets
//   S3 K6               Interval::occ of every voting interval                          -> text positions
//   S4 propose          one wavefront per read: s = pos - seed offset, sort, dedup      -> per-read candidate lists
//      (scan of the per-read candidate / x-byte / y-byte counts; the three totals are the ONE host round trip)
//   S5 gather           (read, text window) pairs, offsets                              -> x, x_off, y, y_off
//   S6 align            Aligner::semiglobal on every candidate (bg_align_batch_dev)      -> records + operations
//   S7 best             per read: highest score, smallest start among equals           -> bg_seed_hit_t + its ops
#include <algorithm>

#include "fm_kernels.h"

struct bg_seed_scratch {
    void* p[16] = {};
    size_t cap[16] = {};
    uint64_t* h_tot = nullptr;  // pinned: totals read back between S4 and S5
};
void bg_seed_scratch_free(bg_seed_scratch* s) {
    if (!s) return;
    for (void* q : s->p) hipFree(q);
    if (s->h_tot) hipHostFree(s->h_tot);
    delete s;
}

namespace {

constexpr uint32_t kMaxProposals = 1024;  // seed slots x max_occ per read (sorted in LDS by one wavefront)
// proposals are sorted as uint32 on an index with 32-bit positions and as uint64 on one with 64-bit positions (round 6:
// the kernels below are templates over that type; ~P(0) marks a dropped proposal)
struct SeedPrm {
    uint32_t S, stride, seed_len, max_occ, pad;
    uint64_t n_text;  // text length without the final sentinel
};

// S2: votes of every seed slot
__global__ __launch_bounds__(256) void se_votes_kernel(uint64_t n_q, const uint8_t* __restrict__ tag, const uint64_t* __restrict__ lower,
                                                       const uint64_t* __restrict__ upper, uint32_t max_occ, uint32_t* __restrict__ cnt,
                                                       uint32_t* __restrict__ panics) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    uint32_t c = 0;
    if (tag[q] == BG_FM_PANIC) atomicOr(panics, 1u);  // the seed reached a byte outside the alphabet: fmindex.rs:229 panics
    if (tag[q] == BG_FM_COMPLETE) {
        const uint64_t sz = upper[q] - lower[q];
        if (sz >= 1 && sz <= max_occ) c = (uint32_t)sz;
    }
    cnt[q] = c;
}

// S4: one wavefront per read.  The read's hits are pos[hoff[r * S] .. hoff[(r + 1) * S)), grouped by seed slot.  Every
// hit proposes s = p - k * stride (dropped if negative or >= n_text); the proposals are sorted, merged (equal ones, and those
// within pad / 2 of the last start kept), and written back over the read's own slice of `pos`; per read: candidates, hits,
// y bytes, x bytes.
template <typename T>
__global__ __launch_bounds__(64) void se_propose_kernel(SeedPrm prm, uint64_t n_reads, const uint64_t* __restrict__ read_off,
                                                        const uint64_t* __restrict__ hoff, uint64_t* __restrict__ pos,
                                                        uint32_t* __restrict__ n_cand, uint32_t* __restrict__ n_hits,
                                                        uint32_t* __restrict__ x_bytes, uint32_t* __restrict__ y_bytes) {
    constexpr T kNoStart = ~(T)0;
    __shared__ T s_val[kMaxProposals];
    __shared__ uint64_t s_off[65];
    const uint64_t r = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t L = (uint32_t)(read_off[r + 1] - read_off[r]);
    for (uint32_t k = lane; k <= prm.S; k += 64) s_off[k] = hoff[r * prm.S + k];  // S <= 64 (checked by the host)
    __syncthreads();
    const uint64_t h0 = s_off[0];
    const uint32_t nh = (uint32_t)(s_off[prm.S] - h0);
    uint32_t n_unique = 0;
    if (nh) {
        uint32_t P = 64;
        while (P < nh) P <<= 1;
        auto proposal = [&](uint32_t i) -> T {
            T v = kNoStart;
            if (i < nh) {
                uint32_t k = 0;  // the seed slot of hit i: last k with s_off[k] - h0 <= i
                while (k + 1 < prm.S && s_off[k + 1] - h0 <= i) k++;
                const uint64_t p = pos[h0 + i];
                const uint64_t o = (uint64_t)k * prm.stride;
                if (p >= o && p - o < prm.n_text) v = (T)(p - o);  // also drops BG_SA_NONE / BG_SA_PANIC
            }
            return v;
        };
        if (nh <= 64) {
            // the usual read (a handful of hits): every lane finds its proposal's rank among the wavefront's by looking at each
            // of the nh values once (a broadcast per value) — no LDS passes, no barriers (the bitonic network below takes 21)
            const T v = proposal(lane);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nh; j++) {
                T u;
                if constexpr (sizeof(T) == 8)
                    u = (T)((uint64_t)(uint32_t)__shfl((int)(uint32_t)((uint64_t)v >> 32), (int)j) << 32 | (uint32_t)__shfl((int)(uint32_t)v, (int)j));
                else
                    u = (T)(uint32_t)__shfl((int)(uint32_t)v, (int)j);
                rank += (u < v || (u == v && j < lane)) ? 1u : 0u;
            }
            // (lanes >= nh hold kNoStart, the largest value: their ranks are nh .. 63 in lane order)
            s_val[lane < nh ? rank : lane] = v;
            __syncthreads();
        } else {
        for (uint32_t i = lane; i < P; i += 64) s_val[i] = proposal(i);
        __syncthreads();
        // bitonic sort of P values by the 64 lanes
        for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                for (uint32_t i = lane; i < P; i += 64) {
                    const uint32_t ixj = i ^ j;
                    if (ixj > i) {
                        const T a = s_val[i], b = s_val[ixj];
                        const bool up = (i & k2) == 0;
                        if ((a > b) == up) {
                            s_val[i] = b;
                            s_val[ixj] = a;
                        }
                    }
                }
                __syncthreads();
            }
        }
        }
        // merge equal proposals (compacted in place: a value never moves up, and a step reads before it writes) ...
        uint32_t base = 0;
        for (uint32_t b0 = 0; b0 < P; b0 += 64) {
            const uint32_t i = b0 + lane;
            const T v = s_val[i];
            const bool keep = v != kNoStart && (i == 0 || s_val[i - 1] != v);
            const uint64_t m = __ballot(keep);
            if (keep) s_val[base + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = v;
            base += (uint32_t)__popcll(m);
        }
        __syncthreads();
        // ... and starts within pad / 2 of the last one kept (the seeds either side of an indel propose the same locus a few
        // bases apart: its window holds both alignments) — in order, so a run of proposals a few bases apart each (a tandem
        // repeat) keeps a start every pad / 2 + 1 bases; the kept starts go back over the read's own slots of `pos`
        if (lane == 0) {
            const T merge = (T)(prm.pad / 2);
            uint32_t kept = 0;
            T last = 0;
            for (uint32_t i = 0; i < base; i++) {
                const T v = s_val[i];
                if (kept == 0 || v - last > merge) {
                    pos[h0 + kept++] = v;
                    last = v;
                }
            }
            s_off[0] = kept;  // (s_off is not read again)
        }
        __syncthreads();
        n_unique = (uint32_t)s_off[0];
    }
    // window bytes of this read's candidates (second pass: the starts are final now)
    __syncthreads();
    uint32_t yl = 0;
    for (uint32_t c = lane; c < n_unique; c += 64) {
        const uint64_t v = pos[h0 + c];
        const uint64_t lo = v > prm.pad ? v - prm.pad : 0u;
        const uint64_t hi = min(prm.n_text, v + L + prm.pad);
        yl += (uint32_t)(hi - lo);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) yl += (uint32_t)__shfl_xor((int)yl, o);
    if (lane == 0) {
        n_cand[r] = n_unique;
        n_hits[r] = nh;
        x_bytes[r] = n_unique * L;
        y_bytes[r] = yl;
    }
}

// S5: one wavefront per read: the (read, window) pairs of its candidates + their offsets
__global__ __launch_bounds__(64) void se_gather_kernel(SeedPrm prm, uint64_t n_reads, const uint8_t* __restrict__ reads,
                                                       const uint64_t* __restrict__ read_off, const uint8_t* __restrict__ text,
                                                       const uint64_t* __restrict__ hoff, const uint64_t* __restrict__ pos,
                                                       const uint64_t* __restrict__ coff, const uint64_t* __restrict__ xoff,
                                                       const uint64_t* __restrict__ yoff, uint8_t* __restrict__ x,
                                                       uint64_t* __restrict__ x_off, uint8_t* __restrict__ y, uint64_t* __restrict__ y_off,
                                                       uint64_t* __restrict__ w_lo) {
    const uint64_t r = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t c0 = coff[r];
    const uint32_t nc = (uint32_t)(coff[r + 1] - c0);
    if (r + 1 == n_reads && lane == 0) {  // closing offsets
        x_off[coff[n_reads]] = xoff[n_reads];
        y_off[coff[n_reads]] = yoff[n_reads];
    }
    if (!nc) return;
    const uint64_t ro = read_off[r];
    const uint32_t L = (uint32_t)(read_off[r + 1] - ro);
    const uint64_t h0 = hoff[r * prm.S];
    uint64_t yo = yoff[r];
    for (uint32_t c = 0; c < nc; c++) {
        const uint64_t v = pos[h0 + c];
        const uint64_t lo = v > prm.pad ? v - prm.pad : 0u;
        const uint64_t hi = min(prm.n_text, v + L + prm.pad);
        const uint64_t xo = xoff[r] + (uint64_t)c * L;
        if (lane == 0) {
            x_off[c0 + c] = xo;
            y_off[c0 + c] = yo;
            w_lo[c0 + c] = lo;
        }
        for (uint32_t i = lane; i < L; i += 64) x[xo + i] = reads[ro + i];
        for (uint32_t i = lane; i < (uint32_t)(hi - lo); i += 64) y[yo + i] = text[lo + i];
        yo += hi - lo;
    }
}

// S7: 16 lanes per read: best candidate (highest score, first = smallest start among equals), record + operations
// `hits` / `ops` are the caller's whole arrays, `r0` the first read of this pass: read r0 + r of the call owns
// ops[(r0 + r) * ops_stride, (r0 + r + 1) * ops_stride) and its ops_off is relative to the caller's `ops`
__global__ __launch_bounds__(256) void se_best_kernel(uint64_t n_reads, uint64_t r0, const uint64_t* __restrict__ coff,
                                                      const uint32_t* __restrict__ n_hits,
                                                      const bg_alignment_t* __restrict__ aln, const uint8_t* __restrict__ c_ops,
                                                      const uint64_t* __restrict__ w_lo, bg_seed_hit_t* __restrict__ hits,
                                                      uint8_t* __restrict__ ops, uint64_t ops_stride) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t l16 = threadIdx.x & 15;
    if (r >= n_reads) return;  // uniform per group of 16
    const uint64_t c0 = coff[r];
    const uint32_t nc = (uint32_t)(coff[r + 1] - c0);
    // key: score (biased to unsigned) in the high word, ~candidate index in the low one: max = best score, first wins
    uint64_t best = 0;
    for (uint32_t c = l16; c < nc; c += 16) {
        const uint32_t sc = (uint32_t)aln[c0 + c].score ^ 0x80000000u;
        const uint64_t key = ((uint64_t)sc << 32) | (uint32_t)~c;
        best = max(best, key);
    }
#pragma unroll
    for (int o = 8; o; o >>= 1) {
        const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(best >> 32), o, 16) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, o, 16);
        best = max(best, other);
    }
    bg_seed_hit_t h;
    memset(&h, 0, sizeof(h));
    h.aln.score = BG_MIN_SCORE;
    h.window_start = h.ref_start = h.ref_end = ~0ull;
    h.n_candidates = nc;
    h.n_seed_hits = n_hits[r];
    h.aln.ops_off = (r0 + r + 1) * ops_stride;
    if (nc) {
        const uint32_t c = ~(uint32_t)best;
        const bg_alignment_t a = aln[c0 + c];
        h.aln = a;
        h.aln.ops_off = (r0 + r + 1) * ops_stride - a.n_ops;
        h.window_start = w_lo[c0 + c];
        h.ref_start = w_lo[c0 + c] + a.ystart;
        h.ref_end = w_lo[c0 + c] + a.yend;
        if (ops && c_ops)
            for (uint32_t i = l16; i < a.n_ops; i += 16) ops[h.aln.ops_off + i] = c_ops[a.ops_off + i];
    }
    if (l16 == 0) hits[r0 + r] = h;
}

}  // namespace

extern "C" int bg_fm_set_text(bg_fm* fm, const uint8_t* text, uint64_t n) {
    if (!fm || !text || n != (fm->wide ? fm->wdev.n : (uint64_t)fm->dev.n)) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(fm->ctx->device));
    if (fm->text_owned) hipFree(fm->d_text);
    fm->d_text = nullptr;
    fm->text_owned = false;
    BG_HIP(hipMalloc(&fm->d_text, n));
    fm->text_owned = true;
    if (bg_copy_pieces(fm->d_text, text, n, hipMemcpyHostToDevice, fm->ctx->stream) != hipSuccess || hipStreamSynchronize(fm->ctx->stream) != hipSuccess)
        return BG_ERR_HIP;
    fm->n_text = n - 1;
    fm->bytes += n;
    return BG_OK;
}

extern "C" int bg_fm_set_text_dev(bg_fm* fm, const uint8_t* d_text, uint64_t n) {
    if (!fm || !d_text || n != (fm->wide ? fm->wdev.n : (uint64_t)fm->dev.n)) return BG_ERR_INVALID_ARG;
    if (fm->text_owned) hipFree(fm->d_text);
    fm->d_text = (void*)d_text;
    fm->text_owned = false;
    fm->n_text = n - 1;
    return BG_OK;
}

extern "C" int bg_seed_extend_batch_dev(bg_fm* fm, const bg_scoring_t* sc, const bg_seed_params_t* prm_in, uint64_t n_reads,
                                        const uint8_t* d_reads, const uint64_t* d_read_off, uint32_t max_read_len,
                                        bg_seed_hit_t* d_hits, uint8_t* d_ops, uint64_t ops_stride, uint64_t* totals,
                                        void* stream) {
    if (!fm || !sc || !prm_in || (n_reads && (!d_read_off || !d_hits))) return BG_ERR_INVALID_ARG;
    if (!fm->d_text || fm->sa_kind == 0) return BG_ERR_INVALID_ARG;  // needs bg_fm_set_text + a suffix array
    if (prm_in->seed_len == 0 || prm_in->stride == 0 || prm_in->max_occ == 0) return BG_ERR_INVALID_ARG;
    if (max_read_len > 65535 || prm_in->pad > 65535) return BG_ERR_TOO_LARGE;
    const uint32_t win_max = max_read_len + 2 * prm_in->pad;
    if (d_ops && ops_stride < (uint64_t)max_read_len + win_max + 4) return BG_ERR_OPS_CAP;
    if (totals) totals[0] = totals[1] = 0;
    if (n_reads == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    hipStream_t st = (hipStream_t)stream;
    BG_HIP(hipSetDevice(ctx->device));
    bg_scratch_guard guard(ctx, st);  // ctx->seed is one scratch set: calls on other streams wait for this one's last kernel
    SeedPrm prm;
    prm.S = max_read_len >= prm_in->seed_len ? (max_read_len - prm_in->seed_len) / prm_in->stride + 1 : 0;
    prm.stride = prm_in->stride;
    prm.seed_len = prm_in->seed_len;
    prm.max_occ = prm_in->max_occ;
    prm.pad = prm_in->pad;
    prm.n_text = fm->n_text;
    if (prm.S > 64 || (uint64_t)prm.S * prm.max_occ > kMaxProposals) return BG_ERR_UNSUPPORTED;
    if (!ctx->seed) ctx->seed = new bg_seed_scratch();
    bg_seed_scratch& W = *ctx->seed;
    if (!W.h_tot) BG_HIP(hipHostMalloc((void**)&W.h_tot, 64, hipHostMallocDefault));
    int rc;
    auto need = [&](int i, size_t bytes) -> int { return bg_reserve(&W.p[i], &W.cap[i], std::max<size_t>(bytes, 64)); };

    uint64_t done_hits = 0, done_cand = 0;
    bool any_panic = false;
    // reads per pass: bounds the scratch (seed slots, proposals, candidate pairs); bg_set_option("seed_chunk_reads") for tests
    // (default: up to 2^21 reads per pass, the passes of a call of equal size — 1.25 M reads went as 2^20 + 0.2 M until round 6:
    //  two host round trips and a set of under-filled launches for a sixth of the reads)
    const uint64_t chunk_cap = ctx->seed_chunk_reads > 0 ? (uint64_t)ctx->seed_chunk_reads : (1u << 21);
    const uint64_t n_pass = (n_reads + chunk_cap - 1) / chunk_cap;
    const uint64_t chunk = ctx->seed_chunk_reads > 0 ? chunk_cap : (n_reads + n_pass - 1) / n_pass;
    for (uint64_t r0 = 0; r0 < n_reads; r0 += chunk) {
        const uint64_t nr = std::min(chunk, n_reads - r0);
        const uint64_t nq = nr * std::max<uint32_t>(prm.S, 1);
        const uint64_t* roff = d_read_off + r0;
        // ---- S1/S2: seeds -> votes -> hit offsets
        if ((rc = need(0, nq))) return rc;              // tag
        if ((rc = need(1, nq * 8))) return rc;          // lower
        if ((rc = need(2, nq * 8))) return rc;          // upper
        if ((rc = need(3, nq * 4))) return rc;          // matched_len, then votes
        if ((rc = need(4, (nq + 1) * 8))) return rc;    // hit offsets
        if ((rc = need(5, 2 * (nq / 2048 + 2) * 8 + 64))) return rc;  // scan partials (+ the panic flag behind them)
        BG_HIP(hipMemsetAsync((uint8_t*)W.p[5] + 2 * (nq / 2048 + 2) * 8, 0, 8, st));
        uint8_t* d_tag = (uint8_t*)W.p[0];
        uint64_t *d_lo = (uint64_t*)W.p[1], *d_hi = (uint64_t*)W.p[2], *d_hoff = (uint64_t*)W.p[4], *d_sums = (uint64_t*)W.p[5];
        uint32_t* d_cnt = (uint32_t*)W.p[3];
        if (prm.S) {
            if ((rc = bg_fm_search_seeds_dev(fm, nr, d_reads, roff, prm.S, prm.stride, prm.seed_len, d_tag, d_lo, d_hi, d_cnt, st))) return rc;
            se_votes_kernel<<<dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st>>>(nq, d_tag, d_lo, d_hi, prm.max_occ, d_cnt,
                                                                                      (uint32_t*)(d_sums + 2 * (nq / 2048 + 2)));
        } else {
            BG_HIP(hipMemsetAsync(d_cnt, 0, nq * 4, st));
        }
        if ((rc = bg_scan_u32(d_cnt, nq, d_hoff, d_sums, st))) return rc;
        BG_HIP(hipMemcpyAsync(&W.h_tot[0], d_hoff + nq, 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(&W.h_tot[4], d_sums + 2 * (nq / 2048 + 2), 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));  // sizes the position array
        const uint64_t n_hits = W.h_tot[0];
        if (W.h_tot[4] & 1) any_panic = true;
        // ---- S3: Interval::occ of the voting intervals
        if ((rc = need(6, n_hits * 8))) return rc;
        uint64_t* d_pos = (uint64_t*)W.p[6];
        if (n_hits && (rc = bg_interval_occ_batch_dev(fm, nq, d_lo, d_hoff, n_hits, d_pos, st))) return rc;
        // ---- S4: proposals -> sorted unique candidates per read, scans of the per-read counts
        if ((rc = need(7, 4 * nr * 4))) return rc;           // n_cand | n_hits | x_bytes | y_bytes
        if ((rc = need(8, 3 * (nr + 1) * 8))) return rc;      // coff | xoff | yoff
        uint32_t* d_nc = (uint32_t*)W.p[7];
        uint32_t *d_nh = d_nc + nr, *d_xb = d_nh + nr, *d_yb = d_xb + nr;
        uint64_t* d_coff = (uint64_t*)W.p[8];
        uint64_t *d_xoff = d_coff + (nr + 1), *d_yoff = d_xoff + (nr + 1);
        if (fm->wide)
            se_propose_kernel<uint64_t><<<dim3((unsigned)nr), dim3(64), 0, st>>>(prm, nr, roff, d_hoff, d_pos, d_nc, d_nh, d_xb, d_yb);
        else
            se_propose_kernel<uint32_t><<<dim3((unsigned)nr), dim3(64), 0, st>>>(prm, nr, roff, d_hoff, d_pos, d_nc, d_nh, d_xb, d_yb);
        BG_HIP(hipGetLastError());
        if ((rc = bg_scan_u32(d_nc, nr, d_coff, d_sums, st))) return rc;
        if ((rc = bg_scan_u32(d_xb, nr, d_xoff, d_sums, st))) return rc;
        if ((rc = bg_scan_u32(d_yb, nr, d_yoff, d_sums, st))) return rc;
        BG_HIP(hipMemcpyAsync(&W.h_tot[1], d_coff + nr, 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(&W.h_tot[2], d_xoff + nr, 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipMemcpyAsync(&W.h_tot[3], d_yoff + nr, 8, hipMemcpyDeviceToHost, st));
        BG_HIP(hipStreamSynchronize(st));  // sizes the candidate pairs
        const uint64_t C = W.h_tot[1], X = W.h_tot[2], Y = W.h_tot[3];
        // ---- S5: the pairs
        const uint64_t cstride = d_ops ? (uint64_t)max_read_len + win_max + 4 : 0;
        if ((rc = need(9, X))) return rc;
        if ((rc = need(10, Y))) return rc;
        if ((rc = need(11, 2 * (C + 1) * 8))) return rc;
        if ((rc = need(12, C * 8))) return rc;
        if ((rc = need(13, C * sizeof(bg_alignment_t)))) return rc;
        if ((rc = need(14, C * cstride))) return rc;
        uint8_t *d_x = (uint8_t*)W.p[9], *d_y = (uint8_t*)W.p[10], *d_cops = d_ops ? (uint8_t*)W.p[14] : nullptr;
        uint64_t* d_cxoff = (uint64_t*)W.p[11];
        uint64_t* d_cyoff = d_cxoff + (C + 1);
        uint64_t* d_wlo = (uint64_t*)W.p[12];
        bg_alignment_t* d_aln = (bg_alignment_t*)W.p[13];
        se_gather_kernel<<<dim3((unsigned)nr), dim3(64), 0, st>>>(prm, nr, d_reads, roff, (const uint8_t*)fm->d_text, d_hoff, d_pos, d_coff,
                                                                  d_xoff, d_yoff, d_x, d_cxoff, d_y, d_cyoff, d_wlo);
        BG_HIP(hipGetLastError());
        // ---- S6: Aligner::semiglobal on every candidate
        if (C && (rc = bg_align_batch_dev_hint(ctx, sc, BG_MODE_SEMIGLOBAL, C, d_x, d_cxoff, d_y, d_cyoff, max_read_len, win_max, d_aln,
                                               d_cops, cstride, st, -1)))
            return rc;
        // ---- S7: best hit per read
        se_best_kernel<<<dim3((unsigned)((nr * 16 + 255) / 256)), dim3(256), 0, st>>>(nr, r0, d_coff, d_nh, d_aln, d_cops, d_wlo, d_hits,
                                                                                      d_ops, ops_stride);
        BG_HIP(hipGetLastError());
        done_hits += n_hits;
        done_cand += C;
    }
    if (totals) {
        totals[0] = done_hits;
        totals[1] = done_cand;
    }
    // a seed that reaches a byte outside the alphabet makes the reference's backward_search panic; here it does not
    // vote, every read is still answered, and the call says so
    return any_panic ? BG_ERR_OUT_OF_ALPHABET : BG_OK;
}

extern "C" int bg_seed_extend_batch(bg_fm* fm, const bg_scoring_t* sc, const bg_seed_params_t* prm, uint64_t n_reads,
                                    const uint8_t* reads, const uint64_t* read_off, bg_seed_hit_t* hits, uint8_t* ops_buf,
                                    uint64_t ops_cap, uint64_t* ops_used) {
    if (!fm || !sc || !prm || (n_reads && (!read_off || !hits))) return BG_ERR_INVALID_ARG;
    if (ops_used) *ops_used = 0;
    if (n_reads == 0) return BG_OK;
    bg_ctx* ctx = fm->ctx;
    BG_HIP(hipSetDevice(ctx->device));
    uint64_t max_len = 0;
    for (uint64_t r = 0; r < n_reads; r++) max_len = std::max(max_len, read_off[r + 1] - read_off[r]);
    if (max_len > 65535) return BG_ERR_TOO_LARGE;
    const uint64_t stride = ops_buf ? 2 * max_len + 2 * (uint64_t)prm->pad + 4 : 0;
    const uint64_t bytes = read_off[n_reads];
    uint8_t *d_reads = nullptr, *d_ops = nullptr;
    uint64_t* d_off = nullptr;
    bg_seed_hit_t* d_hits = nullptr;
    std::vector<uint8_t> h_ops;
    int panic_rc = BG_OK;
    auto run = [&]() -> int {
        hipStream_t st = ctx->stream;
        BG_HIP(hipMalloc((void**)&d_reads, std::max<uint64_t>(bytes, 16)));
        BG_HIP(hipMalloc((void**)&d_off, (n_reads + 1) * 8));
        BG_HIP(hipMalloc((void**)&d_hits, n_reads * sizeof(bg_seed_hit_t)));
        if (stride) BG_HIP(hipMalloc((void**)&d_ops, n_reads * stride));
        if (bytes) BG_HIP(hipMemcpyAsync(d_reads, reads, bytes, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_off, read_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, st));
        int rc = bg_seed_extend_batch_dev(fm, sc, prm, n_reads, d_reads, d_off, (uint32_t)max_len, d_hits, d_ops, stride, nullptr, st);
        if (rc && rc != BG_ERR_OUT_OF_ALPHABET) return rc;
        panic_rc = rc;
        BG_HIP(hipMemcpyAsync(hits, d_hits, n_reads * sizeof(bg_seed_hit_t), hipMemcpyDeviceToHost, st));
        if (stride) {
            h_ops.resize(n_reads * stride);
            BG_HIP(hipMemcpyAsync(h_ops.data(), d_ops, n_reads * stride, hipMemcpyDeviceToHost, st));
        }
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    };
    int rc = run();
    hipFree(d_reads);
    hipFree(d_off);
    hipFree(d_hits);
    hipFree(d_ops);
    if (rc) return rc;
    // compact the winners' operations into the caller's buffer, in read order
    uint64_t used = 0;
    int status = BG_OK;
    for (uint64_t r = 0; r < n_reads; r++) {
        bg_alignment_t& a = hits[r].aln;
        if (a.status) status = a.status;
        if (ops_buf) {
            if (used + a.n_ops <= ops_cap)
                memcpy(ops_buf + used, h_ops.data() + a.ops_off, a.n_ops);
            else if (status == BG_OK)
                status = BG_ERR_OPS_CAP;
        }
        a.ops_off = used;
        used += ops_buf ? a.n_ops : 0;
    }
    if (ops_used) *ops_used = used;
    return status ? status : panic_rc;
}
