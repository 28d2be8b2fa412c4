// K4 — banded_traceback_kernel: one wavefront per pair.  The walk is serial, so all 64 lanes follow it
// redundantly (uniform control flow, broadcast loads) — except along diagonals: whenever the current
// move is MATCH/SUBST, lane t looks at cell (i-t, j-t), a ballot finds how far the path stays on the
// diagonal and the whole run is emitted at once (one pair of dependent loads per run instead of per
// cell; runs are ~25 cells long on 10 % divergent reads).  Border passes
// (/root/reference/src/alignment/pairwise/banded.rs:725-765), traceback (767-831) and the
// "ended outside the band" fix-up (833-855).  Design notes: banded_kernels.h.
#include "banded_kernels.h"

namespace bgband_dev {

__global__ __launch_bounds__(256) void banded_traceback_kernel(const BandArgs a) {
    __builtin_amdgcn_s_setprio(3);  // a chain of dependent loads that runs next to the following sub-batch's fill
    const uint32_t pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (pair >= a.n_pairs) return;  // wave-uniform
    const BandPair bp = a.pairs[pair];
    const SwScoring sc = a.sc;
    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo);
    const uint32_t n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);

    bg_alignment_t rec;
    memset(&rec, 0, sizeof(rec));
    rec.mode = (uint8_t)a.mode;
    rec.ops_off = (a.pair0 + pair + 1) * a.ops_stride;
    if (bp.flags != BP_OK) {
        // banded.rs:407-420: band above MAX_CELLS -> {score: MIN_SCORE, everything else zero, no ops}
        rec.score = BG_MIN_SCORE;
        rec.status = bp.flags == BP_TOO_MANY_CELLS ? (int8_t)BG_OK : (int8_t)BG_ERR_UNSUPPORTED;
        if (lane == 0) a.out[a.pair0 + pair] = rec;
        return;
    }
    const int2* rowc = a.rowc + bp.rowc_off;
    const uint32_t* roff = a.row_off + bp.rowc_off;
    const uint8_t* tb = a.tb + bp.tb_off;
    const int32_t* aux = a.aux + bp.aux_off;
    const BandAux L(m, n);
    const int32_t* gLy = aux + L.off_Ly();
    const int32_t* gLx = aux + L.off_Lx();
    const uint8_t* bits = (const uint8_t*)(aux + L.off_bits());

    int32_t Sm = aux[0];
    uint32_t sbm = (uint32_t)aux[1], lxn = (uint32_t)aux[2], Lym = (uint32_t)aux[3];
    const uint32_t ibm = (uint32_t)aux[4];
    if (m == 0) {
        // x is empty: the band is the single row 0.  S[n%2][0] is wiped by the `S[curr][m]` reset of
        // the last column (banded.rs:561), the first epilogue loop puts Sn[0] back (692-695)
        int32_t Sn0;
        uint32_t Ly0 = 0;
        if (sc.yp > sc.ys) {
            Sn0 = sc.yp;
        } else {
            Sn0 = sc.ys;
            Ly0 = n;
        }
        const int32_t S01 = row0_cell(sc, 1).S;  // later columns cannot improve Sn[0]
        if (S01 + sc.ys > Sn0) {
            Sn0 = S01 + sc.ys;
            Ly0 = n - 1;
        }
        Sm = Sn0 > NEG ? Sn0 : NEG;
        Lym = Ly0;
        lxn = 0;
    }

    // ---- border passes.  Row 0 (banded.rs:725-744)
    uint32_t sb0n;
    {
        const int32_t d_score = sc.go + sc.ge * ((int32_t)n - 1);
        sb0n = d_score > sc.yp ? TB_DEL : TB_YCLIP_PREFIX;
        int32_t best = max(d_score, sc.yp);
        if (sc.ys > best) {
            best = sc.ys;
            sb0n = TB_YCLIP_SUFFIX;
        }
        if (m == 0) sbm = sb0n;  // (m, n) is (0, n)
        if (sc.xs + best > Sm) {
            Sm = sc.xs + best;
            lxn = m;
            sbm = TB_XCLIP_SUFFIX;
        }
    }
    // column 0 (banded.rs:746-765; the loop is empty for m == 0)
    uint32_t sbm0 = TB_START;
    if (m != 0) {
        const int32_t c_score = sc.go + sc.ge * ((int32_t)m - 1);
        sbm0 = c_score > sc.xp ? TB_INS : TB_XCLIP_PREFIX;
        int32_t best = max(c_score, sc.xp);
        if (sc.xs > best) {
            best = sc.xs;
            sbm0 = TB_XCLIP_SUFFIX;
        }
        if (sc.ys + best > Sm) {
            Sm = sc.ys + best;
            Lym = n;
            sbm = TB_YCLIP_SUFFIX;
        }
    }
    // column-0 fold (banded.rs:482-487): only the first band row can fire it
    int32_t fold0 = NEG;
    {
        const uint32_t i0 = max(1u, bp.start_0);
        if (i0 < bp.end_0 && i0 < m) {
            const Col0 c = col0_cell(sc, i0, m, NEG);
            if (c.S + sc.xs > NEG) fold0 = c.S + sc.xs;
        }
    }
    const int2 rc0 = rowc[0];
    auto in_band = [&](uint32_t i, uint32_t j) -> bool {
        const int2 rc = rowc[i];
        return rc.y >= rc.x && (int)j >= rc.x && (int)j <= rc.y;
    };
    // rows of the pair's interior run (band_split; filled by banded_fill2i.hip) keep their bytes in that kernel's layout
    uint32_t in_lo = 1, in_hi = 0;  // rows in_lo .. in_hi
    {
        uint32_t s_a = 0, s_b = 0;
        if (a.split && m != 0 && band_split(sc, bp, m, rowc, kSplitStripRows, s_a, s_b)) {
            in_lo = s_a * kSplitStripRows + 1;
            in_hi = s_b * kSplitStripRows;
        }
    }
    auto raw_cell = [&](uint32_t i, uint32_t off) -> uint32_t {
        const uint32_t b = tb[off];
        return (i >= in_lo && i <= in_hi) ? tb_cell_norm(b) : b;
    };
    // K3v2 leaves the I/D flags as its keys carry them (1 = opened): a.tb_flip turns them into 1 = extended
    auto cellb = [&](uint32_t i, uint32_t j) -> uint32_t { return raw_cell(i, roff[i] + tb_cell_off(j - (uint32_t)rowc[i].x)) ^ a.tb_flip; };
    // S nibble a cell carried while the matrix was being filled (what "open" I/D moves copied)
    auto s_fill = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (i == 0) {
            if (j == 0) return TB_START;
            return (rc0.y >= rc0.x && (int)j >= rc0.x && (int)j <= rc0.y) ? row0_cell(sc, j).sbits : (uint32_t)TB_START;
        }
        if (j == 0) {
            if (in_band(i, 0)) return col0_cell(sc, i, m, fold0).sbits;
            return (i == m && fold0 > NEG) ? (uint32_t)TB_XCLIP_SUFFIX : (uint32_t)TB_START;  // banded.rs:486
        }
        if (in_band(i, j)) return s_nibble_of_code(cellb(i, j) & 7u);
        return i == m ? (uint32_t)TB_XCLIP_SUFFIX : (uint32_t)TB_START;  // banded.rs:652, 671-673
    };
    // nibbles as the traceback finds them
    auto s_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (i == 0) {
            if (j == 0) return TB_START;
            if (j == n) return m == 0 ? sbm : sb0n;  // with an empty x, (0,n) is also (m,n)
            return (sc.go + sc.ge * ((int32_t)j - 1)) > sc.yp ? (uint32_t)TB_DEL : (uint32_t)TB_YCLIP_PREFIX;
        }
        if (j == 0) {
            if (i == m) return sbm0;
            return (sc.go + sc.ge * ((int32_t)i - 1)) > sc.xp ? (uint32_t)TB_INS : (uint32_t)TB_XCLIP_PREFIX;
        }
        if (j == n) return i == m ? sbm : ((uint32_t)bits[i] & 15u);
        return s_fill(i, j);
    };
    auto i_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (i == 0) return TB_START;
        if (j == n) return i == m ? ibm : ((uint32_t)bits[i] >> 4);
        if (!in_band(i, j)) return TB_START;
        if (j == 0) return col0_cell(sc, i, m, fold0).ibits;
        return (cellb(i, j) & 8u) ? (uint32_t)TB_INS : s_fill(i - 1, j);  // banded.rs:583-589
    };
    auto d_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == 0) return TB_START;
        if (i == 0) return (rc0.y >= rc0.x && (int)j >= rc0.x && (int)j <= rc0.y) ? row0_cell(sc, j).dbits : (uint32_t)TB_START;
        if (!in_band(i, j)) return TB_START;
        return (cellb(i, j) & 16u) ? (uint32_t)TB_DEL : s_fill(i, j - 1);  // banded.rs:601-607
    };

    uint8_t* ops_end = a.ops ? a.ops + (a.pair0 + pair + 1) * a.ops_stride : nullptr;
    uint32_t n_ops = 0, n_clips = 0;
    uint32_t clip_len[4] = {0, 0, 0, 0};
    int status = BG_OK;
    auto push = [&](uint32_t op) {
        ++n_ops;
        if (lane == 0 && ops_end && n_ops <= a.ops_stride) ops_end[-(int64_t)n_ops] = (uint8_t)op;
    };
    auto push_many = [&](uint32_t op, uint32_t count) {  // all lanes share the stores
        if (ops_end)
            for (uint32_t t = lane; t < count; t += 64)
                if (n_ops + t + 1 <= a.ops_stride) ops_end[-(int64_t)(n_ops + t + 1)] = (uint8_t)op;
        n_ops += count;
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
        if (steps > guard) {
            status = BG_ERR_TRACEBACK;
            break;
        }
        if ((layer == TB_MATCH || layer == TB_SUBST) && i >= 1 && j >= 1) {
            // ---- diagonal run: lane t inspects (i-t, j-t); interior cells only (row < m, 1 <= column < n)
            bool ok = false;
            uint32_t st = layer;
            if (lane >= 1 && i > lane && j > lane) {
                const uint32_t ii = i - lane, jj = j - lane;
                const int2 rc = rowc[ii];
                if (rc.y >= rc.x && (int)jj >= rc.x && (int)jj <= rc.y) {
                    st = s_nibble_of_code(raw_cell(ii, roff[ii] + tb_cell_off(jj - (uint32_t)rc.x)) & 7u);
                    ok = st == TB_MATCH || st == TB_SUBST;
                }
            }
            const uint64_t mask = __ballot(ok) >> 1;          // bit t-1: lane t continues the run
            const uint32_t r = mask == ~0ull >> 1 ? 63u : (uint32_t)__ffsll((unsigned long long)~mask) - 1u;
            if (lane <= r && ops_end && n_ops + lane + 1 <= a.ops_stride)
                ops_end[-(int64_t)(n_ops + lane + 1)] = (uint8_t)(st == TB_MATCH ? BG_OP_MATCH : BG_OP_SUBST);
            n_ops += r + 1;
            i -= r + 1;
            j -= r + 1;
            steps += r;
            layer = s_nib(i, j);
            continue;
        }
        uint32_t next = TB_START;
        bool bad = false;
        switch (layer) {
            case TB_INS:
                if (i == 0) { bad = true; break; }
                push(BG_OP_INS);
                next = i_nib(i, j);
                i -= 1;
                break;
            case TB_DEL:
                if (j == 0) { bad = true; break; }
                push(BG_OP_DEL);
                next = d_nib(i, j);
                j -= 1;
                break;
            case TB_MATCH:
            case TB_SUBST:
                if (i == 0 || j == 0) { bad = true; break; }
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
                if (lx > i) { bad = true; break; }
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
            case TB_YCLIP_SUFFIX: {
                const uint32_t ly = i == m ? Lym : (uint32_t)gLy[i];
                if (ly > j) { bad = true; break; }
                push_clip(BG_OP_YCLIP, ly);
                j -= ly;
                yend = j;
                next = s_nib(i, j);
                break;
            }
            default:
                bad = true;
                break;
        }
        if (bad) {  // the reference would panic (index underflow / unknown layer)
            status = BG_ERR_TRACEBACK;
            break;
        }
        layer = next;
    }
    // banded.rs:833-855: the traceback stopped on a TB_START cell that is not (0, 0)
    if (status == BG_OK) {
        if (i != 0) {
            const int32_t i_score = sc.go + sc.ge * ((int32_t)i - 1);
            if (i_score > sc.xp) {
                push_many(BG_OP_INS, i);
                xstart = 0;
            } else {
                push_clip(BG_OP_XCLIP, i);
                xstart = i;
            }
        }
        if (j != 0) {
            const int32_t d_score = sc.go + sc.ge * ((int32_t)j - 1);
            if (d_score > sc.yp) {
                push_many(BG_OP_DEL, j);
                ystart = 0;
            } else {
                push_clip(BG_OP_YCLIP, j);
                ystart = j;
            }
        }
    }
    if (ops_end && n_ops > a.ops_stride) status = BG_ERR_OPS_CAP;

    rec.score = Sm;  // S[n % 2][m], banded.rs:859
    rec.xstart = xstart;
    rec.xend = xend;
    rec.ystart = ystart;
    rec.yend = yend;
    rec.xlen = m;
    rec.ylen = n;
    rec.n_ops = n_ops;
    rec.ops_off = (a.pair0 + pair + 1) * a.ops_stride - n_ops;
    const uint32_t nc = n_clips < 4 ? n_clips : 4;
    for (uint32_t c = 0; c < 4; c++) rec.clip_len[c] = c < nc ? clip_len[nc - 1 - c] : 0;
    rec.n_clips = (uint8_t)nc;
    rec.mode = (uint8_t)a.mode;
    rec.status = (int8_t)status;
    if (lane == 0) a.out[a.pair0 + pair] = rec;
}

void launch_band_traceback(const BandArgs& a, hipStream_t st) {
    banded_traceback_kernel<<<dim3((a.n_pairs + 3) / 4), dim3(256), 0, st>>>(a);
}

}  // namespace bgband_dev
