// K2 — sw_traceback_kernel: one lane per pair.  Epilogue of the last column
// (/root/reference/src/alignment/pairwise/mod.rs:808-843) and traceback (845-921) over the
// packed anti-diagonal traceback words K1 wrote.  Design notes: sw_kernels.h.
#include <type_traits>

#include "sw_kernels.h"

namespace bgsw {

template <typename TBW>
__global__ __launch_bounds__(256) void sw_traceback_kernel(const SwArgs a) {
    const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= a.n_pairs) return;
    const SwScoring sc = a.sc;
    const SwGeom geo = a.g;
    const uint32_t R = geo.r, LP = geo.lp, PW = 64 / geo.lp;

    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo);
    const uint32_t n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
    const uint8_t* x = a.x + xo;
    const uint8_t* y = a.y + yo;

    int32_t* aux = a.aux + (size_t)pair * geo.aux_stride;
    int32_t* colS = aux + geo.off_colS();
    const int32_t* colI = aux + geo.off_colI();
    const int32_t* gSn = aux + geo.off_Sn();
    const int32_t* gLy = aux + geo.off_Ly();
    int32_t* gLx = aux + geo.off_Lx();
    int32_t* bits = aux + geo.off_bits();  // (S nibble | I nibble << 4) of column n after the epilogue

    const uint32_t job = pair / PW, grp = pair % PW;
    const TBW* tbj = (const TBW*)a.tb + (size_t)job * geo.nstrips * geo.nsteps * 64 + grp * LP;
    // packed 5 bits of inner cell (1 <= i <= m, 1 <= j <= n)
    auto cell = [&](uint32_t i, uint32_t j) -> uint32_t {
        const uint32_t i1 = i - 1, lrow = i1 / R, rr = i1 - lrow * R;
        const uint32_t st = lrow / LP, llc = lrow - st * LP;
        const TBW w = tbj[((size_t)st * geo.nsteps + (j - 1 + llc)) * 64 + llc];
        return (uint32_t)(w >> (5 * rr)) & 31u;
    };
    const uint32_t sbits_m0 = (uint32_t)aux[0], sbits_0n = (uint32_t)aux[1];
    // S nibble of cell (i,j) as the fill left it (final for every column < n)
    auto s_fill = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == 0) {
            if (i == 0) return TB_START;
            if (i == m) return sbits_m0;
            return col0_cell(sc, i, m, NEG).sbits;
        }
        if (i == 0) return j == n ? sbits_0n : row0_cell(sc, j).sbits;
        switch (cell(i, j) & 7u) {
            case C_DIAG: return x[i - 1] == y[j - 1] ? TB_MATCH : TB_SUBST;  // mod.rs:762
            case C_INS: return TB_INS;
            case C_DEL: return TB_DEL;
            case C_XP: return TB_XCLIP_PREFIX;
            case C_YP: return TB_YCLIP_PREFIX;
            default: return TB_XCLIP_SUFFIX;  // mod.rs:757
        }
    };

    // ---- epilogue, first loop (mod.rs:809-821)
    int32_t Sm = colS[m];
    uint32_t sbm = s_fill(m, n);  // current S nibble of (m, n)
    uint32_t lxn = (uint32_t)gLx[n];
    uint32_t pre_prev = 0;
    for (uint32_t i = 0; i <= m; i++) {
        const uint32_t pre = s_fill(i, n);
        uint32_t ib;  // I nibble the fill gave (i, n)
        if (i == 0)
            ib = TB_START;
        else if (n == 0)
            ib = col0_cell(sc, i, m, NEG).ibits;
        else
            ib = (cell(i, n) & 8u) ? TB_INS : pre_prev;  // mod.rs:740-743
        pre_prev = pre;
        int32_t s_i = (i == m) ? Sm : colS[i];
        uint32_t sb = (i == m) ? sbm : pre;
        if (gSn[i] > s_i) {
            s_i = gSn[i];
            sb = TB_YCLIP_SUFFIX;
            if (i != m) colS[i] = s_i;
        }
        if (i == m) {
            Sm = s_i;
            sbm = sb;
        }
        bits[i] = (int32_t)(sb | (ib << 4));
        if (s_i + sc.xs > Sm) {  // never true at i == m (xclip_suffix <= 0)
            Sm = s_i + sc.xs;
            lxn = m - i;
            sbm = TB_XCLIP_SUFFIX;
        }
    }
    // ---- epilogue, second loop (mod.rs:825-843)
    if (m >= 1) {
        int32_t s_prev = (m == 0) ? Sm : colS[0];
        uint32_t sb_prev = (uint32_t)bits[0] & 15u;
        if (m == 0) sb_prev = sbm;
        for (uint32_t i = 1; i <= m; i++) {
            const int32_t s_score = s_prev + sc.go;
            uint32_t b = (uint32_t)bits[i];
            uint32_t sb = (i == m) ? sbm : (b & 15u);
            uint32_t ib = b >> 4;
            int32_t s_i = (i == m) ? Sm : colS[i];
            if (s_score > colI[i]) ib = sb_prev;
            if (s_score > s_i) {
                s_i = s_score;
                sb = TB_INS;
                if (i == m) {
                    Sm = s_i;
                } else {
                    colS[i] = s_i;
                    if (s_i + sc.xs > Sm) {
                        Sm = s_i + sc.xs;
                        lxn = m - i;
                        sbm = TB_XCLIP_SUFFIX;
                    }
                }
            }
            if (i == m) sbm = sb;
            bits[i] = (int32_t)(sb | (ib << 4));
            s_prev = s_i;
            sb_prev = sb;
        }
    }
    // (m, n) may have been re-marked by a later fold of the first loop: keep `sbm` authoritative
    const uint32_t ib_m = (uint32_t)bits[m] >> 4;

    // ---- traceback (mod.rs:845-908)
    auto s_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == n) return i == m ? sbm : ((uint32_t)bits[i] & 15u);
        return s_fill(i, j);
    };
    auto i_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == n) return i == m ? ib_m : ((uint32_t)bits[i] >> 4);
        if (i == 0) return TB_START;
        if (j == 0) return col0_cell(sc, i, m, NEG).ibits;
        return (cell(i, j) & 8u) ? TB_INS : s_fill(i - 1, j);
    };
    auto d_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == 0) return TB_START;
        if (i == 0) return row0_cell(sc, j).dbits;
        return (cell(i, j) & 16u) ? TB_DEL : s_fill(i, j - 1);  // mod.rs:751-754
    };

    uint8_t* ops_end = a.ops ? a.ops + (a.pair0 + pair + 1) * a.ops_stride : nullptr;
    uint32_t n_ops = 0;
    uint32_t clip_len[4] = {0, 0, 0, 0};
    uint32_t n_clips = 0;
    int status = BG_OK;
    auto push = [&](uint32_t op) {
        if (ops_end) ops_end[-(int64_t)(++n_ops)] = (uint8_t)op;
        else ++n_ops;
    };
    auto push_clip = [&](uint32_t op, uint32_t len) {
        if (a.filter_clips) return;
        if (n_clips < 4) clip_len[n_clips] = len;
        else status = BG_ERR_TRACEBACK;
        n_clips++;
        push(op);
    };

    uint32_t i = m, j = n;
    uint32_t xstart = 0, ystart = 0, xend = m, yend = n;
    uint32_t layer = sbm;
    const uint32_t guard = 2 * (m + n) + 16;
    for (uint32_t steps = 0; layer != TB_START; steps++) {
        if (steps > guard || n_ops + 1 > a.ops_stride) {
            status = BG_ERR_TRACEBACK;
            break;
        }
        uint32_t next;
        switch (layer) {
            case TB_INS:
                push(BG_OP_INS);
                next = i_nib(i, j);
                i -= 1;
                break;
            case TB_DEL:
                push(BG_OP_DEL);
                next = d_nib(i, j);
                j -= 1;
                break;
            case TB_MATCH:
            case TB_SUBST:
                push(layer == TB_MATCH ? BG_OP_MATCH : BG_OP_SUBST);
                next = s_nib(i - 1, j - 1);
                i -= 1;
                j -= 1;
                break;
            case TB_XCLIP_PREFIX:
                push_clip(BG_OP_XCLIP, i);
                xstart = i;
                i = 0;
                next = s_nib(0, j);
                break;
            case TB_XCLIP_SUFFIX: {
                const uint32_t lx = (j == n) ? lxn : (uint32_t)gLx[j];
                push_clip(BG_OP_XCLIP, lx);
                i -= lx;
                xend = i;
                next = s_nib(i, j);
                break;
            }
            case TB_YCLIP_PREFIX:
                push_clip(BG_OP_YCLIP, j);
                ystart = j;
                j = 0;
                next = s_nib(i, 0);
                break;
            default: {  // TB_YCLIP_SUFFIX
                const uint32_t ly = (uint32_t)gLy[i];
                push_clip(BG_OP_YCLIP, ly);
                j -= ly;
                yend = j;
                next = s_nib(i, j);
                break;
            }
        }
        if (i > m || j > n) {  // underflow: the reference would panic on the index
            status = BG_ERR_TRACEBACK;
            break;
        }
        layer = next;
    }

    bg_alignment_t rec;
    rec.score = Sm;  // S[n % 2][m], mod.rs:912
    rec.xstart = xstart;
    rec.xend = xend;
    rec.ystart = ystart;
    rec.yend = yend;
    rec.xlen = m;
    rec.ylen = n;
    rec.n_ops = n_ops;
    rec.ops_off = (a.pair0 + pair + 1) * a.ops_stride - n_ops;
    const uint32_t nc = n_clips < 4 ? n_clips : 4;
    for (uint32_t c = 0; c < 4; c++) rec.clip_len[c] = c < nc ? clip_len[nc - 1 - c] : 0;  // forward order
    rec.n_clips = (uint8_t)nc;
    rec.mode = (uint8_t)a.mode;
    rec.status = (int8_t)status;
    rec._pad = 0;
    a.out[a.pair0 + pair] = rec;
}

void launch_traceback(const SwArgs& a, bool wide, hipStream_t st) {
    const uint32_t blocks = (a.n_pairs + 255) / 256;
    if (wide)
        sw_traceback_kernel<uint64_t><<<dim3(blocks), dim3(256), 0, st>>>(a);
    else
        sw_traceback_kernel<uint32_t><<<dim3(blocks), dim3(256), 0, st>>>(a);
}

}  // namespace bgsw
