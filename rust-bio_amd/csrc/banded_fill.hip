// K3 — banded_fill_kernel<R, SM>.  Design notes: banded_kernels.h.
// Reference: /root/reference/src/alignment/pairwise/banded.rs:406-723.
#include "banded_kernels.h"

namespace bgband_dev {

template <int SM>
__device__ __forceinline__ int32_t bsub_score(const SwScoring& sc, const int32_t* s_table, const int32_t* g_table,
                                              int alpha, uint32_t p, uint32_t q) {
    if (SM == SCORE_PARAMS) return p == q ? sc.match : sc.mismatch;
    if (SM == SCORE_LDS) return s_table[(p & 255u) * alpha + (q & 255u)];
    return g_table[(p & 255u) * alpha + (q & 255u)];
}

// inclusive scan over the 64 lanes with combine(earlier, later) = later.v > earlier.v ? later : earlier
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

// I nibble cases of a last-column cell
enum : uint32_t { IC_OPEN = 0, IC_EXT = 1, IC_YS = 2 };

template <int R, int SM>
__global__ __launch_bounds__(256) void banded_fill_kernel(const BandArgs a) {
    constexpr int RS = 64 * R;  // rows per strip
    __shared__ int32_t s_table[SM == SCORE_LDS ? kMaxLdsAlphabet * kMaxLdsAlphabet : 1];
    __shared__ uint8_t s_map[SM != SCORE_PARAMS ? 256 : 1];
    if (SM != SCORE_PARAMS) {
        for (uint32_t t = threadIdx.x; t < 256; t += blockDim.x) s_map[t] = a.code_map[t];
        if (SM == SCORE_LDS)
            for (int t = threadIdx.x; t < a.alpha * a.alpha; t += blockDim.x) s_table[t] = a.table[t];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint32_t pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= a.n_pairs) return;  // wave-uniform
    const BandPair bp = a.pairs[pair];
    if (bp.flags != BP_OK) return;
    const SwScoring sc = a.sc;
    if (a.x_off[a.pair0 + pair + 1] == a.x_off[a.pair0 + pair]) return;  // m == 0: closed forms only (K4)

    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo);
    const uint32_t n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
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
    uint8_t* gBits = (uint8_t*)(aux + L.off_bits());
    int4* bnd = (int4*)(aux + L.off_bnd());
    const int32_t start_n = (int32_t)bp.start_n, end_n = (int32_t)bp.end_n;

    for (uint32_t j = lane; j <= n; j += 64) gV[j] = NEG;  // S[curr][m] of a column without band rows

    // ---- row 0 (banded.rs:501-508, 518-554): Sn[0] / Ly[0] depend on closed forms only
    const int2 rc0 = rowc[0];
    int32_t Sn0;
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
    // ---- column 0 (banded.rs:440-499): rows [max(1,start_0), end_0); S(i,0) never increases with i,
    // so only the first of them can move the x-suffix-clip fold
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
    const bool m_in_col0 = bp.start_0 <= m && m < bp.end_0;
    if (lane == 0) {
        gLx[0] = (int32_t)lx0;
        gV[0] = m_in_col0 ? col0_cell(sc, m, m, fold0).S : NEG;  // banded.rs:497-499
        gSn[0] = Sn0;
        gLy[0] = (int32_t)Ly0;
    }

    // ---- carries of the last-column epilogue across strips (wave-uniform)
    int64_t e_carry = INT64_MIN;
    uint32_t sbf_carry = TB_START, sb2_carry = TB_START;
    int64_t c1v = INT64_MIN, c2v = INT64_MIN;
    uint32_t c1i = 0, c2i = 0;
    // row m, as its owner lane saw it (broadcast after the strips)
    int32_t Sm_fill = NEG, Ilm = NEG;
    uint32_t sbm_fill = TB_XCLIP_SUFFIX, ibm_fill = TB_START;
    int64_t ssm = INT64_MIN;
    uint32_t sb_above_m = TB_START;  // final S nibble of (m-1, n)

    const uint32_t nstrips = (m + RS - 1) / RS;
    for (uint32_t strip = 0; strip < nstrips; strip++) {
        const uint32_t rb = (strip * 64 + lane) * R;
        const int32_t mrow = (int32_t)m - (int32_t)rb - 1;
        int32_t Sl[R], Dl[R], Il[R], Sn[R], cf[R], cl[R];
        uint32_t Ly[R], px[R], celln[R], icase[R], acc[R];
        uint32_t* tbr[R];  // dword stream of the row: cells cf..cl, four per word
        bool any_row = false;
        int jlo = 0x7fffffff, jhi = -1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            px[r] = 0;
            Sl[r] = Dl[r] = Il[r] = Sn[r] = NEG;
            Ly[r] = 0;
            celln[r] = 0;
            icase[r] = IC_OPEN;
            cf[r] = 1;
            cl[r] = 0;
            tbr[r] = (uint32_t*)tb;
            acc[r] = 0;
            if (i <= m) {
                const int2 rc = rowc[i];
                cf[r] = rc.x;
                cl[r] = rc.y;
                if (rc.y >= rc.x) {
                    any_row = true;
                    tbr[r] = (uint32_t*)(tb + roff[i]);
                    const uint32_t ch = x[i - 1];
                    px[r] = (SM == SCORE_PARAMS) ? ch : ((uint32_t)s_map[ch] | (ch << 8));
                    if (rc.x == 0) {  // (i, 0) is a band cell
                        const Col0 c = col0_cell(sc, i, m, fold0);
                        Sl[r] = c.S;
                        Il[r] = c.I;
                        acc[r] = c.sbits | (c.ibits << 4);  // column 0 keeps whole nibbles
                        if (rc.y == 0) tbr[r][0] = acc[r];
                    }
                    jlo = min(jlo, max(1, rc.x));
                    jhi = max(jhi, rc.y);
                }
            }
        }
        if (strip == 0 && lane == 0 && rc0.y >= rc0.x) {
            jlo = min(jlo, max(1, rc0.x));
            jhi = max(jhi, rc0.y);
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            jlo = min(jlo, __shfl_xor(jlo, o));
            jhi = max(jhi, __shfl_xor(jhi, o));
        }
        const bool strip_has_rows = __any(any_row);
        if (!strip_has_rows && strip != 0) continue;  // rows never inside the band: aux stays zero, Sn stays MIN
        // one extra column on the left so that the diagonal S(i-1, jlo-1) arrives through the pipeline
        if (jlo <= jhi) jlo = max(1, jlo - 1);

        // row above this lane's first row, for lane 0 of a later strip
        int2 rc_above = make_int2(1, 0);
        int32_t Sn_above = NEG;
        if (lane == 0) {
            if (strip == 0) {
                rc_above = rc0;
                Sn_above = Sn0;
            } else {
                rc_above = rowc[rb];
                Sn_above = gSn[rb];
            }
        }
        // the row below this lane's last one: while it is inside the band of a column, that lane (or the next
        // strip) publishes the column's fold instead of this one
        int2 rc_below = make_int2(1, 0);
        if (rb + R + 1 <= m) rc_below = rowc[rb + R + 1];
        // S(rb, 0): the diagonal of this lane's first row at column 1
        int32_t diag0 = NEG;
        {
            int2 ra = rc_above;
            if (lane != 0 && rb <= m) ra = rowc[rb];
            if (ra.y >= ra.x && ra.x == 0) diag0 = rb == 0 ? 0 : col0_cell(sc, rb, m, fold0).S;
        }

        int32_t S_out = NEG, I_out = NEG, cm_out = NEG, Snl_out = NEG;
        int32_t ca_out = 0, q_out = 0;
        int32_t ychunk = 0, ychunk_nx = 0;
        int4 bch = make_int4(NEG, NEG, NEG, 0), bch_nx = bch;
        const int nsteps = jhi >= jlo ? (jhi - jlo + 1) + 63 : 0;
        {
            const int jj = jlo + lane;  // column of chunk 0 for this lane
            if (jj <= jhi && jj >= 1) {
                const uint32_t ch = y[jj - 1];
                ychunk_nx = (SM == SCORE_PARAMS) ? ch : ((uint32_t)s_map[ch] | (ch << 8));
                if (strip) bch_nx = bnd[jj];
            }
        }
        for (int t = 0; t < nsteps; t++) {
            if ((t & 63) == 0) {
                ychunk = ychunk_nx;
                bch = bch_nx;
                const int jj = jlo + t + 64 + lane;
                if (jj <= jhi) {
                    const uint32_t ch = y[jj - 1];
                    ychunk_nx = (SM == SCORE_PARAMS) ? ch : ((uint32_t)s_map[ch] | (ch << 8));
                    if (strip) bch_nx = bnd[jj];
                }
            }
            int32_t S_up = wave_shr1(S_out), I_up = wave_shr1(I_out), cm = wave_shr1(cm_out);
            int32_t ca = wave_shr1(ca_out), q = wave_shr1(q_out), Sn_prev = wave_shr1(Snl_out);
            const int j = jlo + t - lane;
            const bool col_ok = j >= jlo && j <= jhi;
            if (lane == 0) {
                q = ychunk;
                Sn_prev = Sn_above;
                const bool above_in = rc_above.y >= rc_above.x && j >= rc_above.x && j <= rc_above.y;
                S_up = I_up = cm = NEG;
                ca = 0;
                if (above_in) {
                    if (strip) {
                        S_up = bch.x;
                        I_up = bch.y;
                        cm = bch.z;
                        ca = bch.w;
                    } else {
                        S_up = row0_cell(sc, (uint32_t)j).S;  // banded.rs:518-546 (I[curr][0] = MIN)
                    }
                }
            }
            ychunk = wave_shl1(ychunk);
            if (strip) {
                bch.x = wave_shl1(bch.x);
                bch.y = wave_shl1(bch.y);
                bch.z = wave_shl1(bch.z);
                bch.w = wave_shl1(bch.w);
            }
            if (col_ok) {
                const bool last_col = (uint32_t)j == n;
                // banded.rs:564-572
                const int32_t xclip_j = sc.xp + max(last_col ? max(sc.yp, Sn0) : sc.yp, sc.go + sc.ge * (j - 1));
                int32_t diag = diag0;
                diag0 = S_up;
                bool any_in = false;
                int32_t v_best_m = NEG;
                bool m_here = false;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool inb = j >= cf[r] && j <= cl[r];
                    const int32_t left_S = Sl[r];
                    if (inb) {
                        any_in = true;
                        const bool is_m = (r == mrow);
                        const uint32_t i = rb + r + 1;
                        const bool eq = (SM == SCORE_PARAMS) ? px[r] == (uint32_t)q : ((px[r] ^ (uint32_t)q) >> 8) == 0;
                        const int32_t m_sc = diag + bsub_score<SM>(sc, s_table, a.table, a.alpha, px[r], (uint32_t)q);
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
                        if (xclip_j > best) { best = xclip_j; code = C_XP; }
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
                        {  // four cells per store: single-byte stores cost 6x their size in HBM write traffic
                            const uint32_t cj = (uint32_t)(j - cf[r]);
                            acc[r] = (cj & 3u) ? (acc[r] | (cell << (8 * (cj & 3u)))) : cell;
                            if ((cj & 3u) == 3u || j == cl[r]) tbr[r][tb_cell_off(cj & ~3u) >> 2] = acc[r];
                        }
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
                // only the last band row of the column publishes (rows of a column's band are contiguous)
                if (any_in && !(j >= rc_below.x && j <= rc_below.y)) {
                    gV[j] = m_here ? v_best_m : cm;
                    gLx[j] = ca;
                }
                S_out = S_up;
                I_out = I_up;
                cm_out = cm;
                ca_out = ca;
                q_out = q;
                Snl_out = Sn_prev;
                if (lane == 63 && strip + 1 < nstrips) bnd[j] = make_int4(S_up, I_up, cm, ca);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            if (i <= m && cl[r] >= cf[r]) {
                gSn[i] = Sn[r];
                gLy[i] = (int32_t)Ly[r];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // =========== epilogue of the last column for this strip's rows (banded.rs:683-723) ===========
        {
            auto getn = [](uint64_t v, int r) -> uint32_t { return (uint32_t)(v >> (4 * r)) & 15u; };
            auto setn = [](uint64_t& v, int r, uint32_t xv) { v = (v & ~(15ull << (4 * r))) | ((uint64_t)xv << (4 * r)); };
            // the value S[n%2][i] holds when the first loop starts (689-691) and the fill-time S nibble
            uint64_t nibF = 0, nibS = 0, nibI = 0;
            int32_t s1[R];
            bool valid[R], inn[R], chain[R], loop2[R];
            int64_t lane_T = INT64_MIN, lane_c1 = INT64_MIN;
            uint32_t lane_c1i = 0;
            const bool has_row0 = strip == 0 && lane == 0;
            // row 0: S is the closed form if (0,n) is a band cell, else MIN; then Sn[0] (684-701, i = 0)
            int32_t S0fin = NEG;
            uint32_t sb0fill = TB_START;  // S nibble of (0,n) while column n is filled
            bool fired0 = false;          // banded.rs:692-695 at i = 0
            if (has_row0) {
                if (start_n == 0) {
                    const Row0 c = row0_cell(sc, n);
                    S0fin = c.S;
                    sb0fill = c.sbits;
                } else {
                    // banded.rs:503/507/551: the cell only ever carries the y-clip marks
                    sb0fill = (sc.yp > sc.ys && Sn0 == sc.yp) ? (uint32_t)TB_YCLIP_PREFIX : (uint32_t)TB_YCLIP_SUFFIX;
                }
                if (Sn0 > S0fin) {
                    S0fin = Sn0;
                    fired0 = true;
                }
                if (m != 0) {
                    lane_c1 = (int64_t)(S0fin + sc.xs);
                    if (start_n <= 1) lane_T = (int64_t)S0fin;  // row 0 feeds the second loop only if row 1 is in it
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int32_t i = (int32_t)(rb + r + 1);
                valid[r] = i <= (int32_t)m && cl[r] >= cf[r];
                inn[r] = valid[r] && cl[r] == (int)n && cf[r] <= (int)n;
                loop2[r] = inn[r] && i >= max(1, start_n) && i < end_n;  // banded.rs:705
                chain[r] = valid[r] && i >= max(1, start_n) - 1 && i < end_n;  // its S can be read as S[i-1] there
                const uint32_t f = inn[r] ? s_nibble_of_code(celln[r] & 7u)
                                          : ((valid[r] && Sn[r] > NEG) ? (uint32_t)TB_YCLIP_SUFFIX : (uint32_t)TB_START);
                setn(nibF, r, f);
                uint32_t sb = f;
                s1[r] = inn[r] ? Sl[r] : NEG;
                if (valid[r] && r != mrow) {
                    if (Sn[r] > s1[r]) {  // banded.rs:692-695
                        s1[r] = Sn[r];
                        sb = TB_YCLIP_SUFFIX;
                    }
                    const int64_t c = (int64_t)(s1[r] + sc.xs);  // banded.rs:696-700
                    if (s1[r] > NEG && c > lane_c1) {
                        lane_c1 = c;
                        lane_c1i = (uint32_t)i;
                    }
                    if (chain[r]) lane_T = max(lane_T, (int64_t)s1[r] - (int64_t)sc.go * (int64_t)i);
                }
                setn(nibS, r, sb);
            }
            if (has_row0 && m != 0 && !(S0fin > NEG)) lane_c1 = INT64_MIN;
            int64_t incl_T = lane_T;
            {
                uint32_t dummy = 0;
                wave_scan_first_max(lane, incl_T, dummy);
            }
            int64_t excl_T = __shfl_up(incl_T, 1);
            if (lane == 0) excl_T = INT64_MIN;
            excl_T = max(excl_T, e_carry);
            if (has_row0 && m != 0 && start_n <= 1) excl_T = max(excl_T, (int64_t)S0fin);
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
                        if (ss > (int64_t)Il[r]) irepair |= 1u << r;  // banded.rs:709-713
                        if (r == mrow) {
                            ss_m = ss;
                        } else if (ss > (int64_t)s1[r]) {  // banded.rs:714-716
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
            // row m: remember what its owner saw; resolved after the column scan below
            const bool own = mrow >= 0 && mrow < R;
            int32_t t_Sm = NEG, t_Il = NEG;
            uint32_t t_sb = TB_XCLIP_SUFFIX, t_ib = TB_START, t_above = TB_START;
#pragma unroll
            for (int r = 0; r < R; r++)
                if (r == mrow) {
                    t_Sm = inn[r] ? Sl[r] : NEG;
                    t_Il = Il[r];
                    t_sb = inn[r] ? getn(nibF, r) : (uint32_t)TB_XCLIP_SUFFIX;  // banded.rs:671-673
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
    }

    // ---- Sn[m] / Ly[m]: banded.rs:665-670 folded over all columns (first maximum wins)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int64_t bestv = INT64_MIN;
    uint32_t bestj = 0;
    for (uint32_t j0 = 1; j0 <= n; j0 += 64) {
        const uint32_t j = j0 + lane;
        int64_t v = INT64_MIN;
        uint32_t vj = j;
        if (j <= n) {
            const int32_t V = gV[j];
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
        // (m, n) after column n: banded.rs:662, 665-674
        int32_t Sm = Sm_fill;
        uint32_t sbm = sbm_fill, lxn = (uint32_t)gLx[n];
        const bool m_in_n = start_n <= (int32_t)m && (int32_t)m < end_n;
        // banded.rs:665-674 at j = n: with row m inside the band the cell loop already raised Sn[m]
        // (656-658) so 666 cannot fire; outside the band 671-673 stamps XCLIP_SUFFIX last
        if (!m_in_n) sbm = TB_XCLIP_SUFFIX;
        uint32_t ibm = ibm_fill;
        // first loop (684-701): rows < m, then row m
        if (c1v > (int64_t)Sm) {
            Sm = (int32_t)c1v;
            lxn = m - c1i;
            sbm = TB_XCLIP_SUFFIX;
        }
        if (Snm > Sm) {
            Sm = Snm;
            sbm = TB_YCLIP_SUFFIX;
        }
        // second loop (705-723): rows < m, then row m if it is in the band of column n
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

band_fill_fn get_band_fill(int sm) {
    if (sm == SCORE_PARAMS) return banded_fill_kernel<2, SCORE_PARAMS>;
    if (sm == SCORE_LDS) return banded_fill_kernel<2, SCORE_LDS>;
    return banded_fill_kernel<2, SCORE_GLOBAL>;
}

}  // namespace bgband_dev
