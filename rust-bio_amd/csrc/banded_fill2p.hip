// K3p — banded_fill2p_kernel<R, LP>: the interior runs of K3i (banded_fill2i.hip) with TWO pairs per lane group.
//
// K3i spends ~24 lane-instructions per cell, no HBM or LDS limit anywhere near — and, as this kernel's first measurements
// showed, most of its time waiting for its own dependency chain (a dependent vector instruction issues 8.7 cycles behind
// its producer on this chip: tools/microbench/ub_dep; notes at issue_chunk / hand_over below).  The lever on the count is the one
// K1p pulled for the short reads (sw_fill_pk16.inc): every VGPR of the recurrence holds the values of two pairs, 16 bits each,
// and the packed-math VALU (v_pk_sub_u16 clamp, v_pk_max_u16, v_pk_mad_u16) advances a cell of both pairs per instruction.
// K3i's strips of 32 rows as 16 lanes x 2 rows (a lane group is one DPP row; columns skewed by one step per lane), same memory formats on
// every side (bnd / gSn / gLy in K3v2's int32 domain, the interior traceback byte of tb_cell_norm), so K3v2's phases 1 and 2
// (banded_fill2.inc) and K4 do not know which of the two kernels ran.
//
// 16 bits do not hold a 10 kb alignment's scores; they hold a STRIP's: the keys are those of K3i (score << 4 | candidate
// priority << 1 | "opened here") taken RELATIVE to a per-strip, per-pair base — the maximum of the row above the strip is put
// at kTarget, close to the top of the range, because inside a strip a score can rise by at most 32 matches above it and fall
// by a band's width of gap penalties below.  The domain is UNSIGNED with saturating subtraction: 0 is "minus infinity", and
// whatever falls below the floor (base - kTarget / 16) sticks there.  That is not exact — a value that was clamped is too
// HIGH, and what derives from it is wrong — but it is wrong in a bounded way: a clamped value can only climb by one match
// per row, so inside a strip everything that derives from the floor stays <= fb = 32 * match * 16 + 16.  Hence the rule
//     a value above fb is exact, and a maximum above fb was decided between exact candidates,
// and the check the kernel makes: EVERY band cell's S of every interior strip must exceed fb + |gap open| + 32 (so that the
// I / D values opened from it are above fb too).  A pair with one cell at or below that threshold is flagged (aux[5]) and
// recomputed by the int32 kernels behind this launch (banded_fill2.hip: phase 1 and K3i again with BandArgs::redo = 1) —
// detect and recompute, bit-exact either way.  With 10 kb reads at PacBio-like error rates nothing is flagged: a band
// cell sits at most a band's width of gap extensions below its row's best.
// Overflow at the top cannot happen: the base is the exact maximum of the row above (scanned from bnd for a pair's first
// interior strip, carried from Sn of the strip's last row afterwards), the y-prefix-clip candidate of a row is below the
// maximum of every row above it, and kTarget + 32 matches + (match - mismatch) fits 16 bits (host check, banded_api.hip).
// Reference semantics: /root/reference/src/alignment/pairwise/banded.rs:556-680 (see banded_fill2i.hip for what an interior
// strip leaves out of that loop).
#include <type_traits>

#include "banded_kernels.h"

namespace bgband_dev {

namespace {

typedef uint32_t pk;  // two uint16: [15:0] = the even pair of the lane group, [31:16] = the odd one
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u(pk v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ pk bits(u16x2 v) { return __builtin_bit_cast(pk, v); }
__device__ __forceinline__ pk pk_add(pk a, pk b) { return bits((u16x2)(as_u(a) + as_u(b))); }
__device__ __forceinline__ pk pk_sub(pk a, pk b) { return bits((u16x2)(as_u(a) - as_u(b))); }
__device__ __forceinline__ pk pk_subs(pk a, pk b) { return bits(__builtin_elementwise_sub_sat(as_u(a), as_u(b))); }  // max(a - b, 0)
__device__ __forceinline__ pk pk_max(pk a, pk b) { return bits(__builtin_elementwise_max(as_u(a), as_u(b))); }
__device__ __forceinline__ pk pk_min(pk a, pk b) { return bits(__builtin_elementwise_min(as_u(a), as_u(b))); }
__device__ __forceinline__ pk pk_mad(pk a, pk b, pk c) { return bits((u16x2)(as_u(a) * as_u(b) + as_u(c))); }
__device__ __forceinline__ pk dup16(uint32_t v) { return (v & 0xffffu) | (v << 16); }
__device__ __forceinline__ pk pack2(uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | (hi << 16); }
__device__ __forceinline__ uint32_t half_of(pk v, int h) { return h ? v >> 16 : v & 0xffffu; }
__device__ __forceinline__ pk sel(pk mask, pk a, pk b) {  // (a & mask) | (b & ~mask), mask uniform
    pk r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ pk selv(pk mask, pk a, pk b) { return (a & mask) | (b & ~mask); }
// lane L of a row of 16 takes v of lane L - 1; the row's first lane keeps `first` (DPP row_shr:1 without bound_ctrl: a lane
// without a source keeps the old destination) — the lane group is exactly a DPP row, so the value from above the strip
// enters the pipeline without a select
__device__ __forceinline__ pk row_shr1_first(pk first, pk v) {
    return (pk)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x111, 0xf, 0xf, false);
}
// 0xffff in every half that is non-zero
__device__ __forceinline__ pk nz_mask(pk v, pk one) { return pk_sub(pk_subs(one, v), one); }

// WB: wavefronts per block.  4: round 4's launch (187 VGPRs, two blocks per CU) — the default.  8 (`band_p_block512`): ONE
// block per CU compiled for 168 VGPRs (the compiler parks 14 values of the strip prologue in scratch; the step loops are
// the same instructions), so that two fill wavefronts per SIMD leave 176 VGPRs and the chaining of the next sub-batch
// (chain_rows_kernel: 40 VGPRs, four wavefronts per SIMD for 16 384 pairs) is resident as a whole next to the fill instead
// of three wavefronts in four.  Measured in round 5 (profiles/r05_banded_pipeline_experiments.txt): the chaining does
// drop from 22 to 16 ms under the fill, the fill itself runs 3 % longer, and the call does not gain — what runs next to a
// fill (chaining, raster, join, K4: ~7 900 VGPR-ms per SIMD and sub-batch) does not fit the registers a fill leaves
// (176 x 30 ms) whatever the order; the rest runs in the gap between two fills, which is what the cycle already does.
template <int R, int LP, int WB>
__global__ __launch_bounds__(64 * WB) __attribute__((amdgpu_waves_per_eu(WB == 8 ? 3 : 2, WB == 8 ? 3 : 2))) void banded_fill2p_kernel(const BandArgs a) {
    constexpr int RING = 32;   // bytes of LDS per row and pair, indexed by step (banded_fill2i.hip)
    constexpr int FLUSH = 16;  // steps between two hand-overs of complete 16-byte groups == the blocks of the Sn / Ly merge
    static_assert(LP == 16, "a block of 16 steps is one chunk: lane ll prepares / hands over step t0 + ll");
    // The rings of a wavefront are interleaved dword by dword: dword d of ring (pair h, row r) of lane L sits at
    // ((h * R + r) * 8 + d) * 256 + L * 4 of the wavefront's 8 KB — whatever dword the lanes touch, lane L is in bank L (no
    // padding: two blocks are 68 KB, and the 91 KB k-mer join of the next sub-batch fits next to them, banded_api.hip)
    constexpr int WAVE_LDS = 64 * 2 * R * RING;
    __shared__ __align__(16) uint8_t s_tb_all[WB * WAVE_LDS];
    uint8_t* const s_row = s_tb_all + (threadIdx.x >> 6) * WAVE_LDS + (threadIdx.x & 63) * 4;
    auto ring_at = [&](int hr, uint32_t byte) -> uint8_t* { return s_row + ((uint32_t)hr * 8u + (byte >> 2)) * 256u + (byte & 3u); };
    // the strip's last row on its way to bnd: (S, I) of both pairs per step, eight steps per lane group (33 KB per block in
    // all, what K3i takes: the LDS of a CU is handed out in pieces, and one more of them per block keeps the join out)
    __shared__ uint2 s_hand_all[64 * WB / LP][8];
    constexpr int32_t NEGS = kNarrowFloor * 16;
    auto to_s = [](int32_t v) -> int32_t {  // the reference's integers -> the scaled domain (K3v2's map)
        if (v <= NEG / 2) return NEGS + (int32_t)((uint32_t)(max(v, NEG - (1 << 20)) - NEG) << 4);
        return (int32_t)((uint32_t)v << 4);
    };
    auto from_s = [](int32_t v) -> int32_t { return v < -(1 << 29) ? NEG + ((v - NEGS) >> 4) : (v >> 4); };
    const int32_t sn_bias = to_s(a.sc.ys);  // Sn[] is kept without its constant term (banded_fill2.inc)
    constexpr int PW = 64 / LP;  // lane groups per wavefront, two pairs each
    constexpr int RS = LP * R;
    static_assert(RS == (int)kSplitStripRows, "K4 tells the rows of an interior run by this");
    const int lane = threadIdx.x & 63;
    const int g = lane / LP, ll = lane % LP;
    uint2* const s_hand = s_hand_all[threadIdx.x / LP];
    // (the host counts K3i-sized blocks of 32 pairs: launch_band_wait_started)
    if (a.started && threadIdx.x == 0) atomicAdd(a.started, (uint32_t)(2 * WB * PW / 32));
    const uint32_t job = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((uint64_t)job * 2 * PW >= a.n_pairs) return;  // wave-uniform
    const SwScoring sc = a.sc;

    struct Half {
        bool live;
        uint32_t m, n, s_lo, s_hi;
        const uint8_t *x, *y;
        const int2* rowc;
        const uint32_t* roff;
        uint8_t* tb;
        int32_t* aux;
        int32_t base;  // the strip's base: S maximum of the row above it, K3v2's scaled domain
        bool bad;      // a band cell at or below the threshold: the int32 kernels redo this pair
    };
    Half P[2];
    uint32_t s_lo_w = 0xffffffffu, s_hi_w = 0u;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        Half& p = P[h];
        const uint32_t pair = (job * PW + g) * 2 + h;
        p.live = pair < a.n_pairs;
        BandPair bp = {};
        uint64_t xo = 0, yo = 0;
        p.m = p.n = 0;
        if (p.live) {
            bp = a.pairs[pair];
            xo = a.x_off[a.pair0 + pair];
            yo = a.y_off[a.pair0 + pair];
            p.m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo);
            p.n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
            p.live = bp.flags == BP_OK && p.m != 0;
        }
        p.x = a.x + xo;
        p.y = a.y + yo;
        p.rowc = a.rowc + bp.rowc_off;
        p.roff = a.row_off + bp.rowc_off;
        p.tb = a.tb + bp.tb_off;
        p.aux = a.aux + bp.aux_off;
        p.base = NEGS;
        p.bad = false;
        p.s_lo = p.s_hi = 0;
        if (p.live && !(a.split && band_split(sc, bp, p.m, p.rowc, (uint32_t)RS, p.s_lo, p.s_hi))) p.s_lo = p.s_hi = 0;
        if (p.s_lo < p.s_hi) {
            s_lo_w = min(s_lo_w, p.s_lo);
            s_hi_w = max(s_hi_w, p.s_hi);
        }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        s_lo_w = min(s_lo_w, (uint32_t)__shfl_xor((int)s_lo_w, o));
        s_hi_w = max(s_hi_w, (uint32_t)__shfl_xor((int)s_hi_w, o));
    }
    s_lo_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_lo_w);
    s_hi_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_hi_w);

    // K3i's keys (score << 4 | candidate priority << 1 | "the gap was opened here"), as unsigned distances:
    constexpr uint32_t kI = C_INS << 1, kD = C_DEL << 1;
    const uint32_t match_k = ((uint32_t)sc.match << 4) | (C_MATCH << 1);
    const uint32_t misc = ((uint32_t)(-sc.mismatch) << 4) - (C_SUBST << 1);  // -(mismatch key) > 0
    const pk DELTA = dup16(match_k + misc), MISC = dup16(misc);              // m_key = diag + e * DELTA - MISC
    const pk GE = dup16((uint32_t)(-sc.ge) << 4);
    const pk GOI = dup16(((uint32_t)(-sc.go) << 4) - (kI | 1)), GOD = dup16(((uint32_t)(-sc.go) << 4) - (kD | 1));
    const pk ONE = 0x00010001u, LOW4 = 0x000f000fu, BIT4 = 0x00100010u;
    // base of a strip -> kTarget; the threshold every band cell has to exceed (header)
    const int32_t target = (int32_t)((0xfff0u - (match_k + misc) - ((uint32_t)sc.match << 9) - 32u) & ~15u);
    const uint32_t thresh = a.pk_thresh ? (uint32_t)a.pk_thresh : ((uint32_t)sc.match << 9) + 16u + ((uint32_t)(-sc.go) << 4) + 32u;

    for (uint32_t strip = s_lo_w; strip < s_hi_w; strip++) {
        const uint32_t rb = (strip * LP + ll) * R;  // rows rb + 1 .. rb + R, all of them in [2, m - 1] with columns >= 1
        bool act[2];
        int32_t jlo[2], span[2], shift[2];
        uint32_t sn_init[2];
        uint32_t trow[2][R];
        pk Sl[R], Dl[R], Sn[R], SnB[R], ycl[R], Ly[R], px[R], off[R], wn[R];
#pragma unroll
        for (int r = 0; r < R; r++) px[r] = ycl[r] = off[r] = wn[r] = Sn[r] = 0;
        // the base first: everything below is relative to it
        bool scan_any = false;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            act[h] = P[h].live && strip >= P[h].s_lo && strip < P[h].s_hi;
            scan_any |= act[h] && strip == P[h].s_lo;
        }
        if (__builtin_amdgcn_ballot_w64(scan_any) != 0) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (act[h] && strip == P[h].s_lo) {  // (the same for the LP lanes of the group)
                    const BandAux L(P[h].m, P[h].n);
                    const int4* bnd = (const int4*)(P[h].aux + L.off_bnd());
                    const int2 rca = P[h].rowc[strip * RS];
                    int32_t mx = NEGS;
                    for (int j = rca.x + ll; j <= rca.y; j += LP) mx = max(mx, bnd[j].x);
#pragma unroll
                    for (int o = LP / 2; o; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
                    P[h].base = mx;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            shift[h] = P[h].base - target;
            int32_t cf[R], cl[R];
            int lo = 0x7fffffff, hi = -1;
            uint32_t pxh[R], yclh[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t i = rb + r + 1;
                cf[r] = 1;
                cl[r] = 0;
                trow[h][r] = 0;
                pxh[r] = yclh[r] = 0;
                if (act[h]) {
                    const int2 rc = P[h].rowc[i];
                    cf[r] = rc.x;
                    cl[r] = rc.y;
                    if (rc.y >= rc.x) {
                        trow[h][r] = P[h].roff[i];
                        pxh[r] = P[h].x[i - 1];
                        const int32_t yv = (int32_t)((uint32_t)to_s(sc.yp + sc.go + sc.ge * ((int32_t)i - 1)) | (C_YP << 1));
                        yclh[r] = (uint32_t)min(max(yv - shift[h], 0), 0xffff);  // (below the row above's maximum: never the cap)
                        lo = min(lo, rc.x);
                        hi = max(hi, rc.y);
                    }
                }
            }
#pragma unroll
            for (int o = LP / 2; o; o >>= 1) {  // over the LP lanes of the pair
                lo = min(lo, __shfl_xor(lo, o));
                hi = max(hi, __shfl_xor(hi, o));
            }
            if (lo <= hi) lo = max(1, lo - 1);  // one extra column on the left: the diagonal arrives through the pipeline
            jlo[h] = lo;
            span[h] = hi >= lo ? hi - lo : -1;
            // Sn[i] starts at MIN_SCORE (K3i: NEGS - sn_bias); clamped into the range it still decides every "S + ys > Sn[i]" as
            // the exact value would (below the floor: every band cell is above; above the cap: none is)
            const uint32_t sn0 = (uint32_t)min(max(NEGS - sn_bias - shift[h], 0), 0xffff);
            sn_init[h] = sn0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t w = (uint32_t)max(cl[r] - cf[r] + 1, 0);
                const uint32_t o = w ? (uint32_t)(cf[r] - lo) : 0u;
                px[r] |= pxh[r] << (16 * h);
                ycl[r] |= yclh[r] << (16 * h);
                off[r] |= o << (16 * h);
                wn[r] |= w << (16 * h);
                Sn[r] |= sn0 << (16 * h);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            Sl[r] = Dl[r] = SnB[r] = Ly[r] = 0;
        }
        const int span_max = max(span[0], span[1]);
        const int nsteps = span_max >= 0 ? span_max + 1 + (LP - 1) : 0;
        int nsteps_w = nsteps;
#pragma unroll
        for (int o = 32; o; o >>= 1) nsteps_w = max(nsteps_w, __shfl_xor(nsteps_w, o));
        nsteps_w = __builtin_amdgcn_readfirstlane(nsteps_w);
        if (nsteps_w == 0) continue;

        // hand-over of the complete 16-byte groups of every row (banded_fill2i.hip), steps tprev + 1 .. tnow of this lane
        auto flush_tb = [&](int tnow, int tprev, bool final_pass) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int W = (int)half_of(wn[r], h), o = (int)half_of(off[r], h);
                    const int cn = final_pass ? W : min(max(tnow - o + 1, 0), W);
                    const int cp = min(max(tprev - o + 1, 0), W);
                    const int g1 = cn == W ? (W + 15) >> 4 : cn >> 4;
                    const int g0 = cp == W ? g1 : cp >> 4;
#pragma unroll
                    for (int k = 0; k < FLUSH / 16 + 1; k++) {  // (at most 16 new cells per row: two groups when the row ends)
                        const int gk = g0 + k;
                        if (gk < g1) {
                            const uint32_t sbyte = ((uint32_t)gk * 16u + (uint32_t)(o + ll)) & (uint32_t)(RING - 1);
                            const uint32_t d0 = sbyte & ~3u, sh = sbyte & 3u;
                            uint32_t w[5];
#pragma unroll
                            for (int q = 0; q < 5; q++) w[q] = *(const uint32_t*)ring_at(h * R + r, (d0 + 4u * q) & (uint32_t)(RING - 1));
                            *(uint4*)(P[h].tb + trow[h][r] + (uint32_t)gk * kTbGroupStride) =
                                make_uint4(__builtin_amdgcn_alignbyte(w[1], w[0], sh), __builtin_amdgcn_alignbyte(w[2], w[1], sh),
                                           __builtin_amdgcn_alignbyte(w[3], w[2], sh), __builtin_amdgcn_alignbyte(w[4], w[3], sh));
                        }
                    }
                }
            }
        };

        const int4* bnd_r[2];
        int4* bnd_w[2];
        int2 rc_above[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const BandAux L(P[h].m, P[h].n);
            bnd_w[h] = (int4*)(P[h].aux + L.off_bnd());
            bnd_r[h] = bnd_w[h];
            rc_above[h] = act[h] ? P[h].rowc[strip * RS] : make_int2(1, 0);  // strip >= 1
        }
        pk diag0 = 0;  // S(rb, jlo - 1): outside the band, or it arrives through the pipeline

        struct Chunk {
            pk q, S, I;
        };
        struct Raw {  // a chunk's loads in flight
            uint32_t q[2];
            int2 b[2];
        };
        // what a group's first lane needs at step t (the y symbols and the cells above the strip, as the strip above left them
        // in bnd), prepared 16 steps at a time: lane ll prepares step t0 + ll.  The loads are issued a block ahead and
        // converted after that block's steps (a wait for memory right behind the loads costs a round trip per block).
        auto issue_chunk = [&](int t0) -> Raw {
            Raw w;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                w.q[h] = 0;
                w.b[h] = make_int2(NEGS, NEGS);
                const int jj = jlo[h] + t0 + ll;
                if (span[h] >= 0 && jj >= 1 && jj <= jlo[h] + span[h]) {
                    w.q[h] = P[h].y[jj - 1];
                    if (rc_above[h].y >= rc_above[h].x && jj >= rc_above[h].x && jj <= rc_above[h].y) w.b[h] = *(const int2*)&bnd_r[h][jj];
                }
            }
            return w;
        };
        auto finish_chunk = [&](const Raw& w) -> Chunk {
            uint32_t S[2], I[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                S[h] = (uint32_t)min(max(w.b[h].x - shift[h], 0), 0xffff);
                I[h] = (uint32_t)min(max(w.b[h].y - shift[h], 0), 0xffff) | kI;  // (bnd holds K3v2's clean values)
            }
            return Chunk{pack2(w.q[0], w.q[1]), pack2(S[0], S[1]), pack2(I[0], I[1])};
        };
        pk S_out = 0, I_out = 0, q_out = 0;
        pk lo_acc = 0xffffffffu;  // minimum over the band cells of this lane
        pk last_acc = 0;          // maximum over this lane's last row (the group's last lane: the next strip's base)
        auto step = [&](const int t, Chunk& c, auto all_in_tag) {
            constexpr bool ALL_IN = decltype(all_in_tag)::value;
            const pk tpri = dup16(15u - ((uint32_t)t & 15u));
            // (the chunk moves on first: its old registers then die in the three moves below, which write them in place)
            const pk cS = c.S, cI = c.I, cq = c.q;
            c.q = wave_shl1z(cq);
            c.S = wave_shl1z(cS);
            c.I = wave_shl1z(cI);
            pk S_up = row_shr1_first(cS, S_out), I_up = row_shr1_first(cI, I_out), q = row_shr1_first(cq, q_out);
            const int tl = t - ll;  // column jlo[h] + tl of either pair
            if (ALL_IN || (tl >= 0 && tl <= span_max)) {  // (all-in blocks: every lane has a column)
                const pk tlp = dup16((uint32_t)tl);
                pk diag = diag0;
                diag0 = S_up;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const pk left_S = Sl[r];
                    const pk e = pk_subs(ONE, px[r] ^ q);  // 1 where the characters agree
                    const pk m_key = pk_subs(pk_mad(e, DELTA, diag), MISC);
                    const pk Iv_t = pk_max(pk_subs(I_up, GE), pk_subs(S_up, GOI));     // banded.rs:580-588
                    const pk Dv_t = pk_max(pk_subs(Dl[r], GE), pk_subs(left_S, GOD));  // banded.rs:598-607
                    // banded.rs:609-642 (i != m): first maximum == max over the keys
                    const pk kb = pk_max(pk_max(m_key, Iv_t), pk_max(Dv_t, ycl[r]));
                    const pk best = kb & ~LOW4;
                    if (ALL_IN) {
                        Sl[r] = best;
                        Dl[r] = Dv_t & ~ONE;
                        I_up = Iv_t & ~ONE;
                        lo_acc = pk_min(lo_acc, best);
                    } else {
                        // inside the band: off <= tl < off + wn, per half; outside: 0 towards every neighbour
                        const pk inb = nz_mask(pk_subs(wn[r], pk_sub(tlp, off[r])), ONE);
                        const pk inb1 = inb & ~ONE;
                        Sl[r] = best & inb;
                        Dl[r] = Dv_t & inb1;
                        I_up = Iv_t & inb1;
                        lo_acc = pk_min(lo_acc, best | ~inb);
                    }
                    S_up = Sl[r];
                    SnB[r] = pk_max(SnB[r], Sl[r] | tpri);  // banded.rs:655-660, per block of 16 steps
                    const pk cell = sel(BIT4, Dv_t << 4, sel(ONE, Iv_t, kb));
                    const uint32_t slot = (uint32_t)t & (uint32_t)(RING - 1);
                    *ring_at(r, slot) = (uint8_t)cell;
                    *ring_at(R + r, slot) = (uint8_t)(cell >> 16);
                    diag = left_S;
                }
                S_out = S_up;
                I_out = I_up;
                q_out = q;
                last_acc = pk_max(last_acc, S_up);
                if (ll == LP - 1) s_hand[t & 7] = make_uint2(S_up, I_up);
            }
        };
        auto merge_rows = [&](const int t_end) {  // the block that ends at step t_end into (Sn, Ly)
            // Ly = n - j of the block's first maximum: n - (jlo + t_end - ll) + (steps before t_end)
            const pk nmj = pack2((uint32_t)((int32_t)P[0].n - (jlo[0] + t_end - ll)), (uint32_t)((int32_t)P[1].n - (jlo[1] + t_end - ll)));
#pragma unroll
            for (int r = 0; r < R; r++) {
                const pk nb = SnB[r] & ~LOW4;
                const pk up = nz_mask(pk_subs(nb, Sn[r]), ONE);  // nb > Sn
                Ly[r] = selv(up, pk_add(nmj, SnB[r] & LOW4), Ly[r]);
                Sn[r] = pk_max(Sn[r], nb);
                SnB[r] = 0;
            }
        };
        // steps T1 .. T2: every lane of every pair that takes part in this strip has all its R rows inside their bands
        int T1 = -0x40000000, T2 = 0x40000000;  // (a pair that sits this strip out has no say)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (span[h] >= 0) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int W = (int)half_of(wn[r], h), o = (int)half_of(off[r], h);
                    T1 = W ? max(T1, o + ll) : 0x40000000;
                    T2 = min(T2, o + W - 1 + ll);
                }
            }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            T1 = max(T1, __shfl_xor(T1, o));
            T2 = min(T2, __shfl_xor(T2, o));
        }
        T1 = __builtin_amdgcn_readfirstlane(T1);
        T2 = __builtin_amdgcn_readfirstlane(T2);
        // the last row of the strip, eight steps at a time: lane ll < 8 takes the step t0 + ll of the group's last lane to bnd
        // (eight consecutive columns) — one store per lane instead of one per step of the last lane, and nothing a later wait
        // for the chunk loads has to sit out
        auto hand_over = [&](int t0, int t_end) {
            const uint2 v = s_hand[ll & 7];
            const int t = t0 + ll, tl = t - (LP - 1);
            if (ll < 8 && t < t_end && tl >= 0) {
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (tl <= span[h])
                        bnd_w[h][jlo[h] + tl] =
                            make_int4((int32_t)half_of(v.x, h) + shift[h], (int32_t)(half_of(v.y, h) & ~15u) + shift[h], NEGS, 0);
            }
        };
        Chunk c0 = finish_chunk(issue_chunk(0));
        for (int t0 = 0; t0 < nsteps_w; t0 += 16) {
            // (all-in by half block: then the half's last step is below nsteps_w as well)
            const bool all_in_a = t0 >= T1 && t0 + 7 <= T2, all_in_b = t0 + 8 >= T1 && t0 + 15 <= T2;
            const Raw raw = issue_chunk(t0 + 16);
            const int t_end = min(t0 + 16, nsteps_w);
            if (all_in_a) {
#pragma unroll
                for (int k = 0; k < 8; k++) step(t0 + k, c0, std::true_type{});
            } else {
#pragma unroll 1
                for (int t = t0; t < min(t0 + 8, t_end); t++) step(t, c0, std::false_type{});
            }
            hand_over(t0, t_end);
            if (all_in_b) {
#pragma unroll
                for (int k = 8; k < 16; k++) step(t0 + k, c0, std::true_type{});
            } else {
#pragma unroll 1
                for (int t = t0 + 8; t < t_end; t++) step(t, c0, std::false_type{});
            }
            c0 = finish_chunk(raw);  // (waits for these loads and the first half's stores; the stores below are issued behind it)
            merge_rows(t0 + 15);
            hand_over(t0 + 8, t_end);
            if ((t_end & (FLUSH - 1)) == 0) flush_tb(t_end - 1 - ll, t_end - 1 - ll - FLUSH, false);
        }
        {
            const int t_last = (nsteps_w & ~(FLUSH - 1)) - 1;
            flush_tb(0, t_last < 0 ? -0x40000000 : t_last - ll, true);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const BandAux L(P[h].m, P[h].n);
            int32_t* gLy = P[h].aux + L.off_Ly();
            int32_t* gSn = P[h].aux + L.off_Sn();
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t i = rb + r + 1;
                if (act[h] && half_of(wn[r], h) != 0) {
                    // (a row that never raised its Sn keeps the exact initial value, not the clamped one)
                    const uint32_t sn = half_of(Sn[r], h);
                    gSn[i] = from_s(sn == sn_init[h] ? NEGS : (int32_t)sn + shift[h] + sn_bias);
                    gLy[i] = (int32_t)half_of(Ly[r], h);
                }
            }
            if (act[h] && half_of(lo_acc, h) <= thresh) P[h].bad = true;
            // the next strip's base: the maximum of this strip's last row (the last lane's row R - 1); a last row without a
            // band cell leaves nothing to hang the next strip's range on: the int32 kernels take the pair
            const int32_t last_rel = __shfl((int32_t)half_of(last_acc, h), g * LP + LP - 1);
            if (act[h] && last_rel == 0) P[h].bad = true;
            P[h].base = last_rel + shift[h];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next strip reads bnd / gSn of this one
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        int bad = P[h].bad ? 1 : 0;
#pragma unroll
        for (int o = LP / 2; o; o >>= 1) bad |= __shfl_xor(bad, o);
        if (bad && ll == 0) {
            P[h].aux[5] = 1;
            if (a.redo_count) atomicAdd(a.redo_count, 1u);
        }
    }
}

}  // namespace

void launch_fill2p(const BandArgs& a, hipStream_t st) {
    // 16 lanes x 2 rows per pair couple: 8 pairs per wavefront, K3i's grid (two wavefronts per SIMD at 16 384 pairs)
    if (a.p_block512) {
        constexpr uint32_t per_block = 8 * 2 * (64 / 16);
        banded_fill2p_kernel<2, 16, 8><<<dim3((a.n_pairs + per_block - 1) / per_block), dim3(512), 0, st>>>(a);
    } else {
        constexpr uint32_t per_block = 4 * 2 * (64 / 16);
        banded_fill2p_kernel<2, 16, 4><<<dim3((a.n_pairs + per_block - 1) / per_block), dim3(256), 0, st>>>(a);
    }
}

}  // namespace bgband_dev
