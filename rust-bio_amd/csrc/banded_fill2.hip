// K3v2 — banded_fill2_kernel<R, LP> + banded_epilogue_kernel.  Same recurrence and outputs as K3
// (banded_fill.hip, design notes in banded_kernels.h; reference: banded.rs:406-723), different geometry:
//
//   * LP = 8 lanes own one pair — eight pairs per wavefront — with R = 4 rows per lane, i.e. strips of 32 rows.  A strip
//     walks (columns its rows touch) + LP - 1 skew steps; with one pair per wavefront and 128-row strips (K3) 60 % of
//     the lane-steps fall outside a 129-wide band, with 16 lanes x 4 rows 36 %, here 21 %, and the per-step overhead
//     (lane shifts, loop control) is shared by eight pairs.  (Measured fill times of the other geometries: BF2_LP below.)
//   * the last-column epilogue (banded.rs:683-723) needs scans over all rows of a pair; it runs afterwards
//     in banded_epilogue_kernel (one wavefront per pair) from what the fill stored per row.
//
//   * traceback bytes (one per band cell, 16-cell groups of a row, eight rows to a 128-byte line: banded_kernels.h) are staged in LDS: every lane owns a
//     ring of RING bytes per row, a cell is one ds_write_b8, and every FLUSH steps — a wave-uniform moment — all lanes
//     hand the 16-byte groups that have become complete (plus the last, partial group of a finished row) to HBM as
//     dwordx4 stores.  Every group is written exactly once, whole: WRITE_SIZE = the traceback bytes (+ the padding of
//     rows to 16 bytes), where per-row dword streams used to be evicted from L2 half-filled (11x, round 1).
//     Byte layout: bits 0-2 the S candidate that won (C_*), bit 3 / bit 4 "the I / D value opened a gap" — the inverse of
//     K3's "extended" flags: K4 XORs kTbFlip onto what it reads (BandArgs::tb_flip); bits 5-7 are not read.
//
// MatchParams scoring only (Scoring::from_scores); tabulated match functions keep using K3.
#include <type_traits>

#include "banded_kernels.h"

namespace bgband_dev {

#ifndef BF2_LP
#define BF2_LP 8  // measured on 16 384 x 10 kb pairs (fill ms): LP x R = 8x4 58.6, 8x5 58.8, 16x4 67.7, 16x2 76.6, 4x6 79.4, 8x6 85.8, 4x4 86.7, 8x8 88.2, 32x4 88.8
#define BF2_R 4
#endif
#define BF2_LP_DEFAULT BF2_LP

namespace {

enum : uint32_t { IC_OPEN = 0, IC_EXT = 1, IC_YS = 2 };

// (MASK & a) | (~MASK & b) as one v_bfi_b32 (the compiler splits the C expression into and / and / or)
template <int MASK>
__device__ __forceinline__ uint32_t bfi(uint32_t a, uint32_t b) {
    static_assert(MASK >= 0 && MASK <= 64, "inline constant");
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "n"(MASK), "v"(a), "v"(b));
    return r;
}

// NARROW values back to the reference's integers (the inverse of the fill's to_s; identity for !NARROW)
template <bool NARROW>
__device__ __forceinline__ int32_t from_scaled(int32_t v) {
    if (!NARROW) return v;
    constexpr int32_t NEGS = kNarrowFloor * 16;
    return v < -(1 << 29) ? NEG + ((v - NEGS) >> 4) : (v >> 4);
}

// first-maximum scan over the 64 lanes: combine(earlier, later) = later.v > earlier.v ? later : earlier
__device__ __forceinline__ void wave_scan_first_max(int lane, int64_t& v, uint32_t& idx) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t ov = __shfl_up(v, off);
        const uint32_t oi = (uint32_t)__shfl_up((int)idx, off);
        if (lane >= off && !(v > ov)) {
            v = ov;
            idx = oi;
        }
    }
}

// NARROW (every reachable score fits 24 bits; checked by the host): K1's key trick — DP values are kept
// scaled by 16 with the candidate's priority in the low bits, so the reference's first-maximum selection
// (banded.rs:609-642, strict '>') is one integer max, "open" gap candidates carry bit 3, and the cell
// update is branch-free (cells outside the band are computed and discarded).  MIN_SCORE maps to
// NEGS = -2^30 with offsets preserved, which is all the reference's arithmetic on it needs.
// XP: the scoring has an x-prefix clip (xclip_prefix != MIN_SCORE).  Without one the column's clip candidate is
// MIN_SCORE + (something <= 0), which never beats the x-suffix-clip slot every cell starts from (banded.rs:609-642 test
// with a strict '>'): the NARROW path then neither prepares nor passes it along.
template <int R, int LP, bool NARROW, bool XP>
__global__ __launch_bounds__(256) void banded_fill2_kernel(const BandArgs a) {
    static_assert(NARROW || XP, "the generic path always carries the candidate");
    constexpr int RING = R <= 2 ? 128 : R <= 4 ? 64 : 32;  // bytes of LDS per row: twice the flush interval (+ a group) stays intact
    constexpr int FLUSH = RING / 2;    // steps between two hand-overs of complete 16-byte groups
    static_assert(FLUSH % (2 * LP) == 0, "hand-overs fall on chunk-pair boundaries");
    constexpr int LANE_LDS = R * RING + 4;   // + 4: consecutive lanes start one bank apart — the 64 byte writes of a step hit
                                             // 64 different banks when the lanes sit at the same ring position (78.9 % of the
                                             // LDS cycles were bank conflicts with a 16-byte pad: four lanes per bank)
    __shared__ __align__(16) uint8_t s_tb_all[256 * LANE_LDS];  // 65 KB per block of four wavefronts
    uint8_t* const s_row = s_tb_all + threadIdx.x * LANE_LDS;
    constexpr int32_t NEGS = NARROW ? (kNarrowFloor * 16) : NEG;
    // exact maps between the reference's integers and the scaled domain (identity for !NARROW)
    auto to_s = [](int32_t v) -> int32_t {
        if (!NARROW) return v;
        if (v <= NEG / 2) return NEGS + (int32_t)((uint32_t)(max(v, NEG - (1 << 20)) - NEG) << 4);
        return (int32_t)((uint32_t)v << 4);
    };
    auto from_s = [](int32_t v) -> int32_t { return from_scaled<NARROW>(v); };
    // NARROW keeps Sn[] (banded.rs:655-660) without its constant term: Sn[r] holds max_j S(i, j) over the band cells seen so
    // far, started at NEGS - ys so that "S + ys > Sn" is "S > Sn[r]"; the true value is Sn[r] + ys wherever it is read
    const int32_t sn_bias = NARROW ? to_s(a.sc.ys) : 0;
    constexpr int PW = 64 / LP;
    constexpr int RS = LP * R;  // rows per strip
    const int lane = threadIdx.x & 63;
    const int g = lane / LP, ll = lane % LP;
    if (a.started && threadIdx.x == 0) atomicAdd(a.started, 1u);  // see launch_band_wait_started
    const uint32_t job = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((uint64_t)job * PW >= a.n_pairs) return;  // wave-uniform
    const uint32_t pair = job * PW + g;
    const SwScoring sc = a.sc;
    // a pair takes part if it exists, its band is usable and x is not empty (m == 0: closed forms in K4)
    bool live = pair < a.n_pairs;
    BandPair bp = {};
    uint32_t m = 0, n = 0;
    uint64_t xo = 0, yo = 0;
    if (live) {
        bp = a.pairs[pair];
        xo = a.x_off[a.pair0 + pair];
        yo = a.y_off[a.pair0 + pair];
        m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo);
        n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
        live = bp.flags == BP_OK && m != 0;
    }
    if (!live) m = n = 0;
    const uint8_t* x = a.x + xo;
    const uint8_t* y = a.y + yo;
    const int2* rowc = a.rowc + bp.rowc_off;
    const uint32_t* roff = a.row_off + bp.rowc_off;
    uint8_t* tb = a.tb + bp.tb_off;
    int32_t* aux = a.aux + bp.aux_off;
    const BandAux L(m, n);
    int32_t* gLy = aux + L.off_Ly();
    int32_t* gLx = aux + L.off_Lx();
    int32_t* gV = aux + L.off_V();
    int32_t* gSn = aux + L.off_Sn();
    int4* bnd = (int4*)(aux + L.off_bnd());
    int2* gEndV = (int2*)(aux + L.off_endv());
    uint8_t* gEndC = (uint8_t*)(aux + L.off_endc());

    if (live)
        for (uint32_t j = ll; j <= n; j += LP) gV[j] = NEGS;  // S[curr][m] of a column without band rows (gV keeps the fill's
                                                              // domain: the epilogue converts what it reads)

    // ---- row 0 (banded.rs:501-508, 518-554): Sn[0] / Ly[0] depend on closed forms only
    const int2 rc0 = live ? rowc[0] : make_int2(1, 0);
    int32_t Sn0 = NEG;
    uint32_t Ly0 = 0;
    if (sc.yp > sc.ys) {
        Sn0 = sc.yp;
    } else {
        Sn0 = sc.ys;
        Ly0 = n;
    }
    {
        const int jf = max(1, rc0.x);  // first column >= 1 whose band contains row 0; later ones cannot improve
        if (rc0.y >= rc0.x && rc0.y >= jf) {
            const int32_t S0 = row0_cell(sc, (uint32_t)jf).S;
            if (S0 + sc.ys > Sn0) {
                Sn0 = S0 + sc.ys;
                Ly0 = n - (uint32_t)jf;
            }
        }
    }
    // ---- column 0 (banded.rs:440-499): only the first band row can move the x-suffix-clip fold
    int32_t fold0 = NEG;
    uint32_t lx0 = 0;
    {
        const uint32_t i0 = max(1u, bp.start_0);
        if (i0 < bp.end_0 && i0 < m) {
            const Col0 c = col0_cell(sc, i0, m, NEG);
            if (c.S + sc.xs > NEG) {
                fold0 = c.S + sc.xs;
                lx0 = m - i0;
            }
        }
    }
    if (live && ll == 0) {
        const bool m_in_col0 = bp.start_0 <= m && m < bp.end_0;
        gLx[0] = (int32_t)lx0;
        gV[0] = to_s(m_in_col0 ? col0_cell(sc, m, m, fold0).S : NEG);  // banded.rs:497-499
        gSn[0] = Sn0;
        gLy[0] = (int32_t)Ly0;
    }

    uint32_t nstrips = live ? (m + RS - 1) / RS : 0;
    uint32_t nstrips_w = nstrips;
#pragma unroll
    for (int o = 32; o; o >>= 1) nstrips_w = max(nstrips_w, (uint32_t)__shfl_xor((int)nstrips_w, o));
    nstrips_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)nstrips_w);

    for (uint32_t strip = 0; strip < nstrips_w; strip++) {
        const uint32_t rb = (strip * LP + ll) * R;
        const int32_t mrow = (int32_t)m - (int32_t)rb - 1;
        int32_t Sl[R], Dl[R], Il[R], Sn[R], SnB[R], cf[R], cl[R], ycl[R];
        uint32_t Ly[R], px[R], celln[R], icase[R];
        uint32_t trow[R];  // offset of the row's bytes (cells cf..cl) in the pair's traceback block
        int jlo = 0x7fffffff, jhi = -1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            px[r] = 0;
            Sl[r] = Dl[r] = Il[r] = NEGS;
            Sn[r] = NEGS - sn_bias;
            SnB[r] = NEGS;
            ycl[r] = NEGS;
            Ly[r] = 0;
            celln[r] = 0;
            icase[r] = IC_OPEN;
            cf[r] = 1;
            cl[r] = 0;
            trow[r] = 0;
            if (live && i <= m) {
                const int2 rc = rowc[i];
                cf[r] = rc.x;
                cl[r] = rc.y;
                if (rc.y >= rc.x) {
                    trow[r] = roff[i];
                    px[r] = x[i - 1];
                    if (NARROW) ycl[r] = (int32_t)((uint32_t)to_s(sc.yp + sc.go + sc.ge * ((int32_t)i - 1)) | C_YP);
                    if (rc.x == 0) {  // (i, 0) is a band cell
                        const Col0 c = col0_cell(sc, i, m, fold0);
                        Sl[r] = to_s(c.S);
                        Il[r] = to_s(c.I);
                        s_row[r * RING] = (uint8_t)((c.sbits | (c.ibits << 4)) ^ kTbFlip);  // column 0 keeps whole nibbles
                    }
                    jlo = min(jlo, max(1, rc.x));
                    jhi = max(jhi, rc.y);
                }
            }
        }
        if (live && strip == 0 && ll == 0 && rc0.y >= rc0.x) {
            jlo = min(jlo, max(1, rc0.x));
            jhi = max(jhi, rc0.y);
        }
#pragma unroll
        for (int o = LP / 2; o; o >>= 1) {  // over the LP lanes of the pair
            jlo = min(jlo, __shfl_xor(jlo, o));
            jhi = max(jhi, __shfl_xor(jhi, o));
        }
        // one extra column on the left so that the diagonal S(i-1, jlo-1) arrives through the pipeline
        if (jlo <= jhi) jlo = max(1, jlo - 1);
        int nsteps = jhi >= jlo ? (jhi - jlo + 1) + (LP - 1) : 0;
        int nsteps_w = nsteps;
#pragma unroll
        for (int o = 32; o; o >>= 1) nsteps_w = max(nsteps_w, __shfl_xor(nsteps_w, o));
        nsteps_w = __builtin_amdgcn_readfirstlane(nsteps_w);  // the same in every lane: loop control on the scalar unit
        // Hand the traceback bytes that became complete 16-byte groups since the previous hand-over to HBM (see the
        // header).  j_now / j_prev: this lane's column after the current / the previous hand-over step.
        auto flush_tb = [&](int j_now, int j_prev, bool final_pass) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int W = cl[r] - cf[r] + 1;  // cells of the row; <= 0: none
                const int cn = final_pass ? W : min(max(j_now - cf[r] + 1, 0), W);
                const int cp = min(max(j_prev - cf[r] + 1, 0), W);
                const int g1 = cn == W ? (W + 15) >> 4 : cn >> 4;  // a finished row also hands over its last, partial group
                const int g0 = cp == W ? g1 : cp >> 4;
#pragma unroll
                for (int k = 0; k < FLUSH / 16 + 2; k++) {
                    const int gk = g0 + k;
                    if (gk < g1) {
                        const uint32_t* src = (const uint32_t*)(s_row + r * RING + ((gk * 16) & (RING - 1)));  // 4-byte aligned only
                        *(uint4*)(tb + trow[r] + (uint32_t)gk * kTbGroupStride) = make_uint4(src[0], src[1], src[2], src[3]);
                    }
                }
            }
        };
        if (nsteps_w == 0) {  // no pair of this wavefront has band rows beyond column 0 in the strip
            flush_tb(0, -0x40000000, true);
            continue;
        }

        // the row above the strip's first row: the pair's first lane takes its upper neighbours from it.  Every lane of
        // the pair holds its extent — each of them prepares one column of a chunk (below)
        int2 rc_above = make_int2(1, 0);
        if (live) {
            if (strip == 0)
                rc_above = rc0;
            else if (strip * RS <= m)
                rc_above = rowc[strip * RS];
        }
        int32_t Sn_above = NEGS;
        if (live && ll == 0) {
            if (strip == 0)
                Sn_above = to_s(Sn0);
            else if (rb <= m)
                Sn_above = to_s(gSn[rb]);
        }
        // S(rb, 0): the diagonal of this lane's first row at column 1
        int32_t diag0 = NEGS;
        if (live) {
            int2 ra = rc_above;
            if (ll != 0 && rb <= m) ra = rowc[rb];
            if (ra.y >= ra.x && ra.x == 0) diag0 = rb == 0 ? 0 : to_s(col0_cell(sc, rb, m, fold0).S);
        }
        // the row below this lane's last one: while it is inside the band of a column, that lane (or the
        // next strip) publishes the column's fold instead of this one
        int2 rc_below = make_int2(1, 0);
        if (live && rb + R + 1 <= m) rc_below = rowc[rb + R + 1];
        uint32_t wn[R];  // cells of the row (0: none): (j - cf) < wn is the band test
#pragma unroll
        for (int r = 0; r < R; r++) wn[r] = (uint32_t)max(cl[r] - cf[r] + 1, 0);

        // What the pair's first lane needs at column j, prepared LP columns at a time: lane ll of the pair prepares column
        // jlo + t0 + ll — the y symbol, the x-prefix-clip candidate of the column (banded.rs:564-572) and the cell above the
        // strip: (S, I, fold, fold row) as the previous strip's last lane left them, or row 0's closed form — and the
        // chunk moves one lane down per step.  Two chunks alternate so that the loads of one are in flight during the
        // LP steps that consume the other.
        struct Chunk {
            int32_t q, xk, S, I, cm, ca;
        };
        auto load_chunk = [&](int t0) -> Chunk {
            Chunk c = {0, NEGS, NEGS, NEGS, NEGS, 0};
            const int jj = jlo + t0 + ll;
            if (jj >= 1 && jj <= jhi) {
                c.q = y[jj - 1];
                if (XP) {
                    const bool last_col = (uint32_t)jj == n;
                    const int32_t xclip_j = sc.xp + max(last_col ? max(sc.yp, Sn0) : sc.yp, sc.go + sc.ge * (jj - 1));
                    c.xk = NARROW ? (int32_t)((uint32_t)to_s(xclip_j) | C_XP) : xclip_j;
                }
                if (rc_above.y >= rc_above.x && jj >= rc_above.x && jj <= rc_above.y) {
                    if (strip) {
                        const int4 b4 = bnd[jj];
                        c.S = b4.x;
                        c.I = b4.y;
                        c.cm = b4.z;
                        c.ca = b4.w;
                    } else {
                        c.S = to_s(row0_cell(sc, (uint32_t)jj).S);  // banded.rs:518-546 (I[curr][0] = MIN)
                    }
                }
            }
            return c;
        };

        int32_t S_out = NEGS, I_out = NEGS, cm_out = NEGS;
        int32_t ca_out = 0, q_out = 0, xk_out = NEGS;
        // NARROW: Sn[i] / Ly[i] ("first maximum of the row", banded.rs:655-660) per block of 16 steps, K1p's way: inside a
        // block the key S | (15 - t % 16) lets ONE max keep the earliest column of the largest value (S values are multiples
        // of 16), the block's winner is folded into the running (Sn, Ly) — strictly greater only — when the block ends
        // (merge_rows, a wave-uniform moment): or + max per cell instead of compare + select + max
        int32_t SnT_out = NEGS - sn_bias;  // true running Sn of this lane's last row (what the lane below reads)
        auto step = [&](const int t, Chunk& c) {
            const int32_t tpri = 15 - (t & 15);
            int32_t S_up = wave_shr1z(S_out), I_up = wave_shr1z(I_out), cm = wave_shr1z(cm_out);
            int32_t ca = wave_shr1z(ca_out), q = wave_shr1z(q_out), xk = XP ? wave_shr1z(xk_out) : NEGS;
            int32_t Sn_prev = wave_shr1z(NARROW ? SnT_out : Sn[R - 1]) + sn_bias;  // Sn of the row above this lane's first one, columns <= j folded in
            if (ll == 0) {
                S_up = c.S;
                I_up = c.I;
                cm = c.cm;
                ca = c.ca;
                q = c.q;
                if (XP) xk = c.xk;
                Sn_prev = Sn_above;
            }
            const int j = jlo + t - ll;
            const bool col_ok = j >= jlo && j <= jhi;
            if (col_ok) {
                const bool last_col = (uint32_t)j == n;
                int32_t diag = diag0;
                diag0 = S_up;
                bool any_in = false;
                int32_t v_best_m = NEGS;
                bool m_here = false;
                if (NARROW) {
                    const int32_t go_s = sc.go * 16, ge_s = sc.ge * 16, go_t = go_s + 8;  // open candidates carry bit 3
                    const int32_t xs_s = to_s(sc.xs), ys_s = to_s(sc.ys);
                    const int32_t match_k = (sc.match * 16) | (int32_t)C_MATCH, mismatch_k = (sc.mismatch * 16) | (int32_t)C_SUBST;
                    const int32_t xkey_j = xk;
                    (void)n;
                    const int32_t ca_in = ca;
                    int32_t cmk = cm | 15;  // fold key: clean running maximum | row priority (15 = an earlier lane)
                    // LAST: some lane of the wavefront is at column n — only then the extra I candidate
                    // Sn[i-1] + go (banded.rs:590-596) and the last-column records exist
                    // HASM: some lane of the wavefront owns row m in this strip (the last strip of a pair) — only then the
                    // x-suffix-clip slot S[curr][m] is a candidate of a cell and row m is kept out of the fold
                    auto rows = [&](auto last_tag, auto m_tag) {
                        constexpr bool LAST = decltype(last_tag)::value;
                        constexpr bool HASM = decltype(m_tag)::value;
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int32_t jc = j - cf[r];
                            const bool inb = (uint32_t)jc < wn[r];
                            const bool is_m = HASM && (r == mrow);
                            const int32_t left_S = Sl[r];
                            const int32_t m_key = diag + (px[r] == (uint32_t)q ? match_k : mismatch_k);
                            // banded.rs:580-607
                            int32_t Iv_t = max(I_up + ge_s, S_up + go_t);
                            uint32_t ic = IC_OPEN;
                            if (LAST) {
                                ic = (Iv_t & 8) ? IC_OPEN : IC_EXT;
                                const int32_t clipk = Sn_prev + go_s;  // (Sn_prev: a true Sn, see below)
                                const bool ys = last_col && clipk > (Iv_t & ~15);
                                Iv_t = ys ? clipk : Iv_t;
                                ic = ys ? (uint32_t)IC_YS : ic;
                            }
                            const int32_t Dv_t = max(Dl[r] + ge_s, left_S + go_t);
                            const int32_t Iv = Iv_t & ~15, Dv = Dv_t & ~15;
                            // banded.rs:609-642: first maximum wins == max over (score | priority)
                            const int32_t k_init = is_m ? (int32_t)(((uint32_t)cmk & ~15u) | C_XS) : (int32_t)((uint32_t)NEGS | C_XS);
                            int32_t kb = max(max(k_init, m_key), (int32_t)((uint32_t)Iv | C_INS));
                            kb = max(kb, (int32_t)((uint32_t)Dv | C_DEL));
                            if (XP) kb = max(kb, xkey_j);
                            kb = max(kb, ycl[r]);
                            const int32_t best = kb & ~15;
                            Sl[r] = inb ? best : NEGS;
                            Dl[r] = inb ? Dv : NEGS;
                            if (LAST) Il[r] = inb ? Iv : Il[r];  // only I(i, n) is read again (the epilogue's records)
                            S_up = Sl[r];
                            I_up = inb ? Iv : NEGS;
                            // banded.rs:648-653 (a no-op at i == m).  Without row m in sight the band test is in Sl already:
                            // outside the band the key is NEGS + xs_s + (14 - r) <= NEGS + 14 (clip penalties are <= 0, the
                            // API refuses others) and the running key is >= NEGS + 15
                            if (HASM) {
                                const int32_t fk = (int32_t)((uint32_t)(best + xs_s) | (uint32_t)(14 - r));
                                cmk = max(cmk, (inb && !is_m) ? fk : (int32_t)0x80000000);
                            } else {
                                cmk = max(cmk, Sl[r] + (xs_s + (14 - r)));
                            }
                            // banded.rs:655-660, blockwise (see above; outside the band S is NEGS: never a winner)
                            SnB[r] = max(SnB[r], Sl[r] | tpri);
                            // traceback byte, I/D flags as the keys carry them (1 = opened: kTbFlip turns them into K4's
                            // "1 = extended").  Unconditional: outside the band the byte lands on a ring slot that is
                            // rewritten before its group is handed over (left of the band) or never handed over (right of it)
                            // (bits 5-7 of the byte carry score bits: K4 reads bits 0-4 only)
                            const uint32_t cell = bfi<16>((uint32_t)Dv_t << 1, bfi<8>((uint32_t)Iv_t, (uint32_t)kb));
                            s_row[r * RING + ((uint32_t)jc & (uint32_t)(RING - 1))] = (uint8_t)cell;
                            any_in = any_in || inb;
                            m_here = m_here || (inb && is_m);
                            v_best_m = (inb && is_m) ? best : v_best_m;
                            if (LAST) {
                                celln[r] = (inb && last_col) ? ((cell & 31u) ^ kTbFlip) : celln[r];
                                icase[r] = (inb && last_col) ? ic : icase[r];
                            }
                            diag = left_S;
                            if (LAST) Sn_prev = max(Sn[r], SnB[r] & ~15) + ys_s;  // the true running value, this column included
                        }
                    };
                    if (__any(last_col) || __any(mrow >= 0 && mrow < R))
                        rows(std::true_type{}, std::true_type{});
                    else
                        rows(std::false_type{}, std::false_type{});
                    const int32_t lo = cmk & 15;
                    ca = lo == 15 ? ca_in : (mrow - 14 + lo);
                    cm = cmk & ~15;
                } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool inb = j >= cf[r] && j <= cl[r];
                    const int32_t left_S = Sl[r];
                    if (inb) {
                        any_in = true;
                        const bool is_m = (r == mrow);
                        const uint32_t i = rb + r + 1;
                        const bool eq = px[r] == (uint32_t)q;
                        const int32_t m_sc = diag + (eq ? sc.match : sc.mismatch);
                        // banded.rs:580-596
                        const int32_t ie = I_up + sc.ge, io = S_up + sc.go;
                        const bool iext = ie > io;
                        int32_t Iv = iext ? ie : io;
                        uint32_t ic = iext ? IC_EXT : IC_OPEN;
                        if (last_col) {
                            const int32_t clip = Sn_prev + sc.go;
                            if (clip > Iv) {
                                Iv = clip;
                                ic = IC_YS;
                            }
                        }
                        // banded.rs:598-607
                        const int32_t de = Dl[r] + sc.ge, dop = left_S + sc.go;
                        const bool dext = de > dop;
                        const int32_t Dv = dext ? de : dop;
                        // banded.rs:609-642
                        int32_t best = is_m ? cm : NEG;
                        uint32_t code = C_XS;
                        if (m_sc > best) { best = m_sc; code = eq ? C_MATCH : C_SUBST; }
                        if (Iv > best) { best = Iv; code = C_INS; }
                        if (Dv > best) { best = Dv; code = C_DEL; }
                        if (xk > best) { best = xk; code = C_XP; }
                        const int32_t yclip_i = sc.yp + sc.go + sc.ge * ((int32_t)i - 1);
                        if (yclip_i > best) { best = yclip_i; code = C_YP; }
                        Sl[r] = best;
                        Dl[r] = Dv;
                        Il[r] = Iv;
                        S_up = best;
                        I_up = Iv;
                        // banded.rs:648-653 (a no-op at i == m)
                        if (!is_m && best + sc.xs > cm) { cm = best + sc.xs; ca = (int32_t)(m - i); }
                        // banded.rs:655-660
                        if (best + sc.ys > Sn[r]) { Sn[r] = best + sc.ys; Ly[r] = n - (uint32_t)j; }
                        const uint32_t cell = code | (iext ? 8u : 0u) | (dext ? 16u : 0u);
                        s_row[r * RING + ((uint32_t)(j - cf[r]) & (uint32_t)(RING - 1))] = (uint8_t)(cell ^ kTbFlip);
                        if (last_col) {
                            celln[r] = cell;
                            icase[r] = ic;
                        }
                        if (is_m) {
                            m_here = true;
                            v_best_m = best;
                        }
                    } else {  // outside the band: MIN_SCORE towards every neighbour
                        Sl[r] = NEG;
                        Dl[r] = NEG;
                        S_up = NEG;
                        I_up = NEG;
                    }
                    diag = left_S;
                    Sn_prev = Sn[r];
                }
                }
                // only the last band row of the column publishes (rows of a column's band are contiguous)
                if (any_in && !(j >= rc_below.x && j <= rc_below.y)) {
                    gV[j] = m_here ? v_best_m : cm;
                    gLx[j] = ca;
                }
                S_out = S_up;
                I_out = I_up;
                if (NARROW) SnT_out = max(Sn[R - 1], SnB[R - 1] & ~15);
                cm_out = cm;
                ca_out = ca;
                q_out = q;
                if (XP) xk_out = xk;
                if (ll == LP - 1 && strip + 1 < nstrips) bnd[j] = make_int4(S_up, I_up, cm, ca);
            }
            // the chunk moves on (all lanes active again; its old values are dead: the moves are in place)
            c.q = wave_shl1z(c.q);
            if (XP) c.xk = wave_shl1z(c.xk);
            c.S = wave_shl1z(c.S);
            c.I = wave_shl1z(c.I);
            c.cm = wave_shl1z(c.cm);
            c.ca = wave_shl1z(c.ca);
        };
        // fold the block that ends at step t_end (t_end % 16 == 15) into (Sn, Ly): the winner sat at step t_end - priority
        auto merge_rows = [&](const int t_end) {
            const int32_t nmj_end = (int32_t)n - (jlo + t_end - ll);  // n - j at step t_end
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int32_t nb = SnB[r] & ~15;
                const bool up = nb > Sn[r];
                Ly[r] = up ? (uint32_t)(nmj_end + (SnB[r] & 15)) : Ly[r];
                Sn[r] = max(Sn[r], nb);
                SnB[r] = NEGS;
            }
        };
        static_assert(2 * LP == 16 || !NARROW, "the Sn blocks are the chunk pairs");
        Chunk c_even = load_chunk(0), c_odd;
        for (int t0 = 0; t0 < nsteps_w; t0 += 2 * LP) {
            c_odd = load_chunk(t0 + LP);
#pragma unroll 1
            for (int t = t0; t < min(t0 + LP, nsteps_w); t++) step(t, c_even);
            c_even = load_chunk(t0 + 2 * LP);
#pragma unroll 1
            for (int t = t0 + LP; t < min(t0 + 2 * LP, nsteps_w); t++) step(t, c_odd);
            if (NARROW) merge_rows(t0 + 2 * LP - 1);  // also the last, partial block: its missing steps added nothing
            const int t_done = min(t0 + 2 * LP, nsteps_w);  // wave-uniform; FLUSH is a multiple of 2 * LP
            if ((t_done & (FLUSH - 1)) == 0) flush_tb(jlo + t_done - 1 - ll, jlo + t_done - 1 - ll - FLUSH, false);
        }
        {
            const int t_last = (nsteps_w & ~(FLUSH - 1)) - 1;  // step of the last hand-over inside the loop (-1: none)
            flush_tb(0, t_last < 0 ? -0x40000000 : jlo + t_last - ll, true);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            if (live && i <= m && cl[r] >= cf[r]) {
                gSn[i] = from_s(Sn[r] + sn_bias);
                gLy[i] = (int32_t)Ly[r];
                if (cl[r] == (int)n && cf[r] <= (int)n) {  // inside the band of the last column
                    gEndV[i] = make_int2(from_s(Sl[r]), from_s(Il[r]));
                    gEndC[i] = (uint8_t)(celln[r] | (icase[r] << 5));
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next strip reads bnd / gSn of this one
    }
}

// Last-column epilogue (banded.rs:683-723) + Sn[m] / Ly[m] (665-670) for one pair per wavefront, from the
// per-row values the fill left in aux.  The arithmetic is K3's (banded_fill.hip), fed from memory.
template <int R, bool NARROW>
__global__ __launch_bounds__(256) void banded_epilogue_kernel(const BandArgs a) {
    constexpr int RS = 64 * R;
    const int lane = threadIdx.x & 63;
    const uint32_t pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= a.n_pairs) return;  // wave-uniform
    const BandPair bp = a.pairs[pair];
    if (bp.flags != BP_OK) return;
    const SwScoring sc = a.sc;
    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo);
    const uint32_t n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
    if (m == 0) return;
    const int2* rowc = a.rowc + bp.rowc_off;
    int32_t* aux = a.aux + bp.aux_off;
    const BandAux L(m, n);
    int32_t* gLy = aux + L.off_Ly();
    const int32_t* gLx = aux + L.off_Lx();
    const int32_t* gV = aux + L.off_V();
    const int32_t* gSn = aux + L.off_Sn();
    uint8_t* gBits = (uint8_t*)(aux + L.off_bits());
    const int2* gEndV = (const int2*)(aux + L.off_endv());
    const uint8_t* gEndC = (const uint8_t*)(aux + L.off_endc());
    const int32_t start_n = (int32_t)bp.start_n, end_n = (int32_t)bp.end_n;
    const int32_t Sn0 = gSn[0];

    int64_t e_carry = INT64_MIN;
    uint32_t sbf_carry = TB_START, sb2_carry = TB_START;
    int64_t c1v = INT64_MIN, c2v = INT64_MIN;
    uint32_t c1i = 0, c2i = 0;
    int32_t Sm_fill = NEG, Ilm = NEG;
    uint32_t sbm_fill = TB_XCLIP_SUFFIX, ibm_fill = TB_START;
    int64_t ssm = INT64_MIN;
    uint32_t sb_above_m = TB_START;

    const uint32_t nstrips = (m + RS - 1) / RS;
    for (uint32_t strip = 0; strip < nstrips; strip++) {
        const uint32_t rb = (strip * 64 + lane) * R;
        const int32_t mrow = (int32_t)m - (int32_t)rb - 1;
        int32_t Sl[R], Il[R], Sn[R], cf[R], cl[R];
        uint32_t celln[R], icase[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            Sl[r] = Il[r] = Sn[r] = NEG;
            celln[r] = 0;
            icase[r] = IC_OPEN;
            cf[r] = 1;
            cl[r] = 0;
            if (i <= m) {
                const int2 rc = rowc[i];
                cf[r] = rc.x;
                cl[r] = rc.y;
                if (rc.y >= rc.x) {
                    Sn[r] = gSn[i];
                    if (rc.y == (int)n && rc.x <= (int)n) {
                        const int2 v = gEndV[i];
                        const uint32_t c = gEndC[i];
                        Sl[r] = v.x;
                        Il[r] = v.y;
                        celln[r] = c & 31u;
                        icase[r] = c >> 5;
                    }
                }
            }
        }
        auto getn = [](uint64_t v, int r) -> uint32_t { return (uint32_t)(v >> (4 * r)) & 15u; };
        auto setn = [](uint64_t& v, int r, uint32_t xv) { v = (v & ~(15ull << (4 * r))) | ((uint64_t)xv << (4 * r)); };
        uint64_t nibF = 0, nibS = 0, nibI = 0;
        int32_t s1[R];
        bool valid[R], inn[R], chain[R], loop2[R];
        int64_t lane_T = INT64_MIN, lane_c1 = INT64_MIN;
        uint32_t lane_c1i = 0;
        const bool has_row0 = strip == 0 && lane == 0;
        int32_t S0fin = NEG;
        uint32_t sb0fill = TB_START;
        bool fired0 = false;
        if (has_row0) {
            if (start_n == 0) {
                const Row0 c = row0_cell(sc, n);
                S0fin = c.S;
                sb0fill = c.sbits;
            } else {
                sb0fill = (sc.yp > sc.ys && Sn0 == sc.yp) ? (uint32_t)TB_YCLIP_PREFIX : (uint32_t)TB_YCLIP_SUFFIX;
            }
            if (Sn0 > S0fin) {
                S0fin = Sn0;
                fired0 = true;
            }
            lane_c1 = (int64_t)(S0fin + sc.xs);
            if (start_n <= 1) lane_T = (int64_t)S0fin;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int32_t i = (int32_t)(rb + r + 1);
            valid[r] = i <= (int32_t)m && cl[r] >= cf[r];
            inn[r] = valid[r] && cl[r] == (int)n && cf[r] <= (int)n;
            loop2[r] = inn[r] && i >= max(1, start_n) && i < end_n;
            chain[r] = valid[r] && i >= max(1, start_n) - 1 && i < end_n;
            const uint32_t f = inn[r] ? s_nibble_of_code(celln[r] & 7u)
                                      : ((valid[r] && Sn[r] > NEG) ? (uint32_t)TB_YCLIP_SUFFIX : (uint32_t)TB_START);
            setn(nibF, r, f);
            uint32_t sb = f;
            s1[r] = inn[r] ? Sl[r] : NEG;
            if (valid[r] && r != mrow) {
                if (Sn[r] > s1[r]) {
                    s1[r] = Sn[r];
                    sb = TB_YCLIP_SUFFIX;
                }
                const int64_t c = (int64_t)(s1[r] + sc.xs);
                if (s1[r] > NEG && c > lane_c1) {
                    lane_c1 = c;
                    lane_c1i = (uint32_t)i;
                }
                if (chain[r]) lane_T = max(lane_T, (int64_t)s1[r] - (int64_t)sc.go * (int64_t)i);
            }
            setn(nibS, r, sb);
        }
        if (has_row0 && !(S0fin > NEG)) lane_c1 = INT64_MIN;
        int64_t incl_T = lane_T;
        {
            uint32_t dummy = 0;
            wave_scan_first_max(lane, incl_T, dummy);
        }
        int64_t excl_T = __shfl_up(incl_T, 1);
        if (lane == 0) excl_T = INT64_MIN;
        excl_T = max(excl_T, e_carry);
        if (has_row0 && start_n <= 1) excl_T = max(excl_T, (int64_t)S0fin);
        uint32_t prev_sbf = (uint32_t)__shfl_up((int)getn(nibF, R - 1), 1);
        if (lane == 0) prev_sbf = strip == 0 ? sb0fill : sbf_carry;

        uint32_t irepair = 0;
        int64_t lane_c2 = INT64_MIN, ss_m = INT64_MIN;
        uint32_t lane_c2i = 0;
        {
            int64_t run_T = excl_T;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int32_t i = (int32_t)(rb + r + 1);
                if (loop2[r]) {
                    const int64_t ss = run_T == INT64_MIN ? INT64_MIN : run_T + (int64_t)sc.go * (int64_t)i;
                    if (ss > (int64_t)Il[r]) irepair |= 1u << r;
                    if (r == mrow) {
                        ss_m = ss;
                    } else if (ss > (int64_t)s1[r]) {
                        setn(nibS, r, TB_INS);
                        const int64_t c = ss + (int64_t)sc.xs;
                        if (c > lane_c2) {
                            lane_c2 = c;
                            lane_c2i = (uint32_t)i;
                        }
                    }
                }
                if (chain[r] && r != mrow) run_T = max(run_T, (int64_t)s1[r] - (int64_t)sc.go * (int64_t)i);
            }
        }
        uint32_t prev_sb2 = (uint32_t)__shfl_up((int)getn(nibS, R - 1), 1);
        if (lane == 0) prev_sb2 = strip == 0 ? (fired0 ? (uint32_t)TB_YCLIP_SUFFIX : sb0fill) : sb2_carry;
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t ib = TB_START;
            if (inn[r]) {
                ib = icase[r] == IC_EXT ? (uint32_t)TB_INS
                                        : (icase[r] == IC_YS ? (uint32_t)TB_YCLIP_SUFFIX : (r ? getn(nibF, r ? r - 1 : 0) : prev_sbf));
                if (irepair & (1u << r)) ib = r ? getn(nibS, r ? r - 1 : 0) : prev_sb2;
            }
            setn(nibI, r, ib);
        }
        int64_t g1 = lane_c1, g2 = lane_c2;
        uint32_t g1i = lane_c1i, g2i = lane_c2i;
        wave_scan_first_max(lane, g1, g1i);
        wave_scan_first_max(lane, g2, g2i);
        const int64_t t1v = __shfl(g1, 63), t2v = __shfl(g2, 63);
        const uint32_t t1i = (uint32_t)__shfl((int)g1i, 63), t2i = (uint32_t)__shfl((int)g2i, 63);
        if (t1v > c1v) { c1v = t1v; c1i = t1i; }
        if (t2v > c2v) { c2v = t2v; c2i = t2i; }
        e_carry = max(e_carry, __shfl(incl_T, 63));
        sbf_carry = (uint32_t)__shfl((int)getn(nibF, R - 1), 63);
        sb2_carry = (uint32_t)__shfl((int)getn(nibS, R - 1), 63);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            if (valid[r] && r != mrow) gBits[i] = (uint8_t)(getn(nibS, r) | (getn(nibI, r) << 4));
        }
        const bool own = mrow >= 0 && mrow < R;
        int32_t t_Sm = NEG, t_Il = NEG;
        uint32_t t_sb = TB_XCLIP_SUFFIX, t_ib = TB_START, t_above = TB_START;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (r == mrow) {
                t_Sm = inn[r] ? Sl[r] : NEG;
                t_Il = Il[r];
                t_sb = inn[r] ? getn(nibF, r) : (uint32_t)TB_XCLIP_SUFFIX;
                t_ib = getn(nibI, r);
                t_above = r ? getn(nibS, r ? r - 1 : 0) : prev_sb2;
            }
        const uint64_t ownmask = __ballot(own);
        if (ownmask) {
            const int src = __ffsll((unsigned long long)ownmask) - 1;
            Sm_fill = __shfl(t_Sm, src);
            Ilm = __shfl(t_Il, src);
            sbm_fill = (uint32_t)__shfl((int)t_sb, src);
            ibm_fill = (uint32_t)__shfl((int)t_ib, src);
            ssm = __shfl(ss_m, src);
            sb_above_m = (uint32_t)__shfl((int)t_above, src);
        }
    }

    // ---- Sn[m] / Ly[m]: banded.rs:665-670 folded over all columns (first maximum wins)
    int64_t bestv = INT64_MIN;
    uint32_t bestj = 0;
    for (uint32_t j0 = 1; j0 <= n; j0 += 64) {
        const uint32_t j = j0 + lane;
        int64_t v = INT64_MIN;
        uint32_t vj = j;
        if (j <= n) {
            const int32_t V = from_scaled<NARROW>(gV[j]);
            if (V + sc.ys > NEG) v = (int64_t)(V + sc.ys);
        }
        wave_scan_first_max(lane, v, vj);
        const int64_t cv = __shfl(v, 63);
        const uint32_t cj = (uint32_t)__shfl((int)vj, 63);
        if (cv > bestv) {
            bestv = cv;
            bestj = cj;
        }
    }
    if (lane == 0) {
        int32_t Snm = NEG;
        uint32_t Lym = 0;
        if (bestv != INT64_MIN) {
            Snm = (int32_t)bestv;
            Lym = n - bestj;
        }
        int32_t Sm = Sm_fill;
        uint32_t sbm = sbm_fill, lxn = (uint32_t)gLx[n];
        const bool m_in_n = start_n <= (int32_t)m && (int32_t)m < end_n;
        if (!m_in_n) sbm = TB_XCLIP_SUFFIX;
        uint32_t ibm = ibm_fill;
        if (c1v > (int64_t)Sm) {
            Sm = (int32_t)c1v;
            lxn = m - c1i;
            sbm = TB_XCLIP_SUFFIX;
        }
        if (Snm > Sm) {
            Sm = Snm;
            sbm = TB_YCLIP_SUFFIX;
        }
        if (c2v > (int64_t)Sm) {
            Sm = (int32_t)c2v;
            lxn = m - c2i;
            sbm = TB_XCLIP_SUFFIX;
        }
        if (m_in_n && (int32_t)m >= max(1, start_n)) {
            if (ssm > (int64_t)Ilm) ibm = sb_above_m;
            if (ssm > (int64_t)Sm) {
                Sm = (int32_t)ssm;
                sbm = TB_INS;
            }
        }
        aux[0] = Sm;
        aux[1] = (int32_t)sbm;
        aux[2] = (int32_t)lxn;
        aux[3] = (int32_t)Lym;
        aux[4] = (int32_t)ibm;
        gLy[m] = (int32_t)Lym;
    }
}

}  // namespace

// The fill's grid is ONE round of blocks (16 384 pairs = 512 blocks, two per CU, resident for the whole kernel), so
// whatever holds LDS or registers on a CU at the moment the fill is dispatched delays some of its blocks by the full
// co-runner's duration and with them the whole kernel.  A kernel that should run NEXT to the fill (the k-mer join of the
// following sub-batch) therefore waits on its own stream until every fill block has counted itself in.
__global__ void band_wait_started_kernel(const uint32_t* counter, uint32_t target) {
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__builtin_readcyclecounter() - t0 > 50000000ull) break;  // ~20 ms: give up waiting, never hang the device
        __builtin_amdgcn_s_sleep(64);
    }
}
void launch_band_wait_started(const uint32_t* counter, uint32_t target, hipStream_t st) {
    band_wait_started_kernel<<<dim3(1), dim3(1), 0, st>>>(counter, target);
}
uint32_t band_fill2_blocks(uint32_t n_pairs) {
    constexpr int PW = 64 / BF2_LP_DEFAULT;
    const uint32_t jobs = (n_pairs + PW - 1) / PW;
    return (jobs + 3) / 4;
}

bool launch_band_fill2(const BandArgs& a, bool narrow, hipStream_t st, hipEvent_t after_fill) {

    constexpr int LP = BF2_LP, R = BF2_R, PW = 64 / LP;
    const uint32_t jobs = (a.n_pairs + PW - 1) / PW;
    if (narrow) {
        if (a.sc.xp > NEG / 2)
            banded_fill2_kernel<R, LP, true, true><<<dim3((jobs + 3) / 4), dim3(256), 0, st>>>(a);
        else
            banded_fill2_kernel<R, LP, true, false><<<dim3((jobs + 3) / 4), dim3(256), 0, st>>>(a);
        if (after_fill) (void)hipEventRecord(after_fill, st);
        banded_epilogue_kernel<2, true><<<dim3((a.n_pairs + 3) / 4), dim3(256), 0, st>>>(a);
    } else {
        banded_fill2_kernel<R, LP, false, true><<<dim3((jobs + 3) / 4), dim3(256), 0, st>>>(a);
        if (after_fill) (void)hipEventRecord(after_fill, st);
        banded_epilogue_kernel<2, false><<<dim3((a.n_pairs + 3) / 4), dim3(256), 0, st>>>(a);
    }
    return true;
}

}  // namespace bgband_dev
