// K3i — banded_fill2i_kernel<R, LP>: the INTERIOR run of every pair of a K3v2 sub-batch (band_split, banded_kernels.h) in
// int32.  Since K3p (banded_fill2p.hip: the same run, two pairs per lane group in 16-bit keys) this kernel runs the pairs K3p
// flags (BandArgs::redo) and the scorings / lengths K3p does not take.
//
// Same geometry, same domain (scores scaled by 16, candidate priority in the low bits), same memory formats as K3v2
// (banded_fill2.inc: LP = 8 lanes own a pair, R = 4 rows per lane, strips of 32 rows; traceback bytes staged in per-row LDS
// rings and handed over in whole 16-byte groups; strips talk through bnd / gSn / gLy) — so K3v2 runs the strips before the
// run (phase 1), this kernel the run, K3v2 the strips behind it (phase 2), and nothing in between has to be converted.
// The traceback rings are indexed by step here (K3v2: by column): conflict-free byte writes, hand-overs realign.
// What an interior strip does not have (reference: banded.rs:556-680):
//   * x clips: no x-prefix-clip candidate (564-572, 625-631), and the x-suffix-clip fold S[curr][m] / Lx[j] (648-653) is
//     MIN_SCORE + something that never wins (band_split's conditions) — not computed, not published;
//   * column n: no Sn[i-1] + go candidate (590-596), no last-column records;
//   * row m, row 0, column 0: no closed forms, no S[curr][m] slot.
// That leaves S / I / D, the y-prefix-clip candidate, Sn[i] / Ly[i] (655-660) and the traceback byte: three lane-to-lane
// moves per step instead of seven, three chunk fields instead of six, a cell of 29 instructions, and — the kernel being
// this one loop — 2 x ~110 VGPRs per SIMD instead of 2 x 218: the builder kernels of the next sub-batch (the chaining's event
// loop above all) find room next to it.
#include <type_traits>

#include "banded_kernels.h"

namespace bgband_dev {

#ifndef BF2_LP
#define BF2_LP 8
#define BF2_R 4
#endif

namespace {

template <int MASK>
__device__ __forceinline__ uint32_t bfi(uint32_t a, uint32_t b) {  // (MASK & a) | (~MASK & b)
    static_assert(MASK >= 0 && MASK <= 64, "inline constant");
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "n"(MASK), "v"(a), "v"(b));
    return r;
}

template <int R, int LP, int RING>  // RING: bytes of LDS per row (64: banded_fill2.inc's; 32: half the LDS, twice the hand-overs)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void banded_fill2i_kernel(const BandArgs a) {
    constexpr int FLUSH = RING / 2;   // steps between two hand-overs of complete 16-byte groups
    static_assert(FLUSH % (2 * LP) == 0 && 2 * LP == 16, "hand-overs fall on chunk-pair boundaries; the Sn blocks are the chunk pairs");
    constexpr int LANE_LDS = R * RING + 4;  // lanes one bank apart
    __shared__ __align__(16) uint8_t s_tb_all[256 * LANE_LDS];
    uint8_t* const s_row = s_tb_all + threadIdx.x * LANE_LDS;
    constexpr int32_t NEGS = kNarrowFloor * 16;
    auto to_s = [](int32_t v) -> int32_t {  // the reference's integers -> the scaled domain (K3v2's map)
        if (v <= NEG / 2) return NEGS + (int32_t)((uint32_t)(max(v, NEG - (1 << 20)) - NEG) << 4);
        return (int32_t)((uint32_t)v << 4);
    };
    auto from_s = [](int32_t v) -> int32_t { return v < -(1 << 29) ? NEG + ((v - NEGS) >> 4) : (v >> 4); };
    const int32_t sn_bias = to_s(a.sc.ys);  // Sn[] is kept without its constant term (banded_fill2.inc)
    constexpr int PW = 64 / LP;
    constexpr int RS = LP * R;
    static_assert(RS == (int)kSplitStripRows, "K4 tells the rows of an interior run by this");
    const int lane = threadIdx.x & 63;
    const int g = lane / LP, ll = lane % LP;
    if (a.started && threadIdx.x == 0) atomicAdd(a.started, 1u);  // see launch_band_wait_started
    const uint32_t job = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((uint64_t)job * PW >= a.n_pairs) return;  // wave-uniform
    const uint32_t pair = job * PW + g;
    const SwScoring sc = a.sc;
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
        if (a.redo && live) live = (a.aux + bp.aux_off)[5] != 0;  // only what K3p flagged (banded_fill2p.hip)
    }
    const uint8_t* x = a.x + xo;
    const uint8_t* y = a.y + yo;
    const int2* rowc = a.rowc + bp.rowc_off;
    const uint32_t* roff = a.row_off + bp.rowc_off;
    uint8_t* tb = a.tb + bp.tb_off;
    int32_t* aux = a.aux + bp.aux_off;
    const BandAux L(m, n);
    int32_t* gLy = aux + L.off_Ly();
    int32_t* gSn = aux + L.off_Sn();
    int4* bnd = (int4*)(aux + L.off_bnd());

    uint32_t s_lo = 0, s_hi = 0;
    if (live && !(a.split && band_split(sc, bp, m, rowc, (uint32_t)RS, s_lo, s_hi))) s_lo = s_hi = 0;
    uint32_t s_lo_w = s_lo < s_hi ? s_lo : 0xffffffffu, s_hi_w = s_lo < s_hi ? s_hi : 0u;
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        s_lo_w = min(s_lo_w, (uint32_t)__shfl_xor((int)s_lo_w, o));
        s_hi_w = max(s_hi_w, (uint32_t)__shfl_xor((int)s_hi_w, o));
    }
    s_lo_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_lo_w);
    s_hi_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_hi_w);

    // Keys of this kernel: score << 4 | candidate priority << 1 | "the gap was opened here".  I and D values CARRY their
    // priority (C_INS, C_DEL) in bits 1-3 wherever they go, so they enter the cell's maximum as they are (K3v2 clears the
    // low bits of both and ors the priority in: two instructions per cell each); the open flag sits below the priority,
    // where it cannot reorder candidates of different kinds, and makes "open" win a tie against "extend" as the
    // reference's strict '>' does (banded.rs:583-589, 601-607).  The traceback byte of an interior row is therefore
    // bit 0 = I opened, bits 1-3 = the S move, bit 4 = D opened (tb_cell_norm, banded_kernels.h, for K4).
    constexpr int32_t kI = (int32_t)(C_INS << 1), kD = (int32_t)(C_DEL << 1);
    const int32_t ge_s = sc.ge * 16, go_ti = sc.go * 16 + (kI | 1), go_td = sc.go * 16 + (kD | 1);
    const int32_t match_k = (sc.match * 16) | (int32_t)(C_MATCH << 1), mismatch_k = (sc.mismatch * 16) | (int32_t)(C_SUBST << 1);

    for (uint32_t strip = s_lo_w; strip < s_hi_w; strip++) {
        const bool act = live && strip >= s_lo && strip < s_hi;
        const uint32_t rb = (strip * LP + ll) * R;  // rows rb + 1 .. rb + R, all of them in [2, m - 1] with columns >= 1
        int32_t Sl[R], Dl[R], Sn[R], SnB[R], cf[R], cl[R], ycl[R];
        uint32_t Ly[R], px[R], wn[R];
        uint32_t trow[R];
        int jlo = 0x7fffffff, jhi = -1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            px[r] = 0;
            Sl[r] = NEGS;
            Dl[r] = NEGS | kD;
            Sn[r] = NEGS - sn_bias;
            SnB[r] = NEGS;
            ycl[r] = NEGS;
            Ly[r] = 0;
            cf[r] = 1;
            cl[r] = 0;
            trow[r] = 0;
            if (act) {
                const int2 rc = rowc[i];
                cf[r] = rc.x;
                cl[r] = rc.y;
                if (rc.y >= rc.x) {
                    trow[r] = roff[i];
                    px[r] = x[i - 1];
                    ycl[r] = (int32_t)((uint32_t)to_s(sc.yp + sc.go + sc.ge * ((int32_t)i - 1)) | (C_YP << 1));
                    jlo = min(jlo, rc.x);
                    jhi = max(jhi, rc.y);
                }
            }
            wn[r] = (uint32_t)max(cl[r] - cf[r] + 1, 0);
        }
#pragma unroll
        for (int o = LP / 2; o; o >>= 1) {  // over the LP lanes of the pair
            jlo = min(jlo, __shfl_xor(jlo, o));
            jhi = max(jhi, __shfl_xor(jhi, o));
        }
        if (jlo <= jhi) jlo = max(1, jlo - 1);  // one extra column on the left: the diagonal arrives through the pipeline
        const int nsteps = jhi >= jlo ? (jhi - jlo + 1) + (LP - 1) : 0;
        int nsteps_w = nsteps;
#pragma unroll
        for (int o = 32; o; o >>= 1) nsteps_w = max(nsteps_w, __shfl_xor(nsteps_w, o));
        nsteps_w = __builtin_amdgcn_readfirstlane(nsteps_w);

        auto flush_tb = [&](int j_now, int j_prev, bool final_pass) {  // (banded_fill2.inc)
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int W = cl[r] - cf[r] + 1;
                const int cn = final_pass ? W : min(max(j_now - cf[r] + 1, 0), W);
                const int cp = min(max(j_prev - cf[r] + 1, 0), W);
                const int g1 = cn == W ? (W + 15) >> 4 : cn >> 4;
                const int g0 = cp == W ? g1 : cp >> 4;
#pragma unroll
                for (int k = 0; k < FLUSH / 16 + 2; k++) {
                    const int gk = g0 + k;
                    if (gk < g1) {
                        // the ring is indexed by STEP (below): cell c of the row was computed at step c + off, so the
                        // group's 16 bytes start at ring byte (16 gk + off) % RING — any alignment, possibly wrapping:
                        // five aligned dwords, funnel-shifted into four
                        const uint32_t sbyte = ((uint32_t)gk * 16u + (uint32_t)(cf[r] - jlo + ll)) & (uint32_t)(RING - 1);
                        const uint8_t* ring = s_row + r * RING;
                        const uint32_t d0 = sbyte & ~3u, sh = sbyte & 3u;
                        uint32_t w[5];
#pragma unroll
                        for (int q = 0; q < 5; q++) w[q] = *(const uint32_t*)(ring + ((d0 + 4u * q) & (uint32_t)(RING - 1)));
                        *(uint4*)(tb + trow[r] + (uint32_t)gk * kTbGroupStride) =
                            make_uint4(__builtin_amdgcn_alignbyte(w[1], w[0], sh), __builtin_amdgcn_alignbyte(w[2], w[1], sh),
                                       __builtin_amdgcn_alignbyte(w[3], w[2], sh), __builtin_amdgcn_alignbyte(w[4], w[3], sh));
                    }
                }
            }
        };
        if (nsteps_w == 0) continue;  // (no ring holds anything: every strip ends with a final hand-over)

        const int2 rc_above = act ? rowc[strip * RS] : make_int2(1, 0);  // strip >= 1
        int32_t diag0 = NEGS;  // S(rb, jlo - 1): outside the band, or it arrives through the pipeline

        struct Chunk {
            int32_t q, S, I;
        };
        // what the pair's first lane needs at column j (the y symbol and the cell above the strip, as the strip above left
        // it in bnd), prepared LP columns at a time: lane ll of the pair prepares column jlo + t0 + ll
        auto load_chunk = [&](int t0) -> Chunk {
            Chunk c = {0, NEGS, NEGS | kI};
            const int jj = jlo + t0 + ll;
            if (jj >= 1 && jj <= jhi) {
                c.q = y[jj - 1];
                if (rc_above.y >= rc_above.x && jj >= rc_above.x && jj <= rc_above.y) {
                    const int2 b2 = *(const int2*)&bnd[jj];
                    c.S = b2.x;
                    c.I = b2.y | kI;  // (bnd holds K3v2's clean values)
                }
            }
            return c;
        };
        int32_t S_out = NEGS, I_out = NEGS | kI, q_out = 0;
        // all_in: every row of every lane that has a column at this step is inside its band (the 16-step blocks between T1
        // and T2 below) — no band test, nothing forced to MIN_SCORE: four instructions per cell less
        auto step = [&](const int t, Chunk& c, auto all_in_tag) {
            constexpr bool ALL_IN = decltype(all_in_tag)::value;
            const int32_t tpri = 15 - (t & 15);
            int32_t S_up = wave_shr1z(S_out), I_up = wave_shr1z(I_out), q = wave_shr1z(q_out);
            if (ll == 0) {
                S_up = c.S;
                I_up = c.I;
                q = c.q;
            }
            const int j = jlo + t - ll;
            if (j >= jlo && j <= jhi) {
                int32_t diag = diag0;
                diag0 = S_up;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int32_t jc = j - cf[r];
                    const bool inb = ALL_IN || (uint32_t)jc < wn[r];
                    const int32_t left_S = Sl[r];
                    const int32_t m_key = diag + (px[r] == (uint32_t)q ? match_k : mismatch_k);
                    const int32_t Iv_t = max(I_up + ge_s, S_up + go_ti);    // banded.rs:580-588
                    const int32_t Dv_t = max(Dl[r] + ge_s, left_S + go_td);  // banded.rs:598-607
                    // banded.rs:609-642 (i != m: S[curr][i] = MIN_SCORE first): first maximum == max over the keys
                    int32_t kb = max(m_key, Iv_t);
                    kb = max(kb, Dv_t);
                    kb = max(kb, ycl[r]);
                    const int32_t best = kb & ~15;
                    Sl[r] = inb ? best : NEGS;  // outside the band: MIN_SCORE towards every neighbour
                    Dl[r] = inb ? (Dv_t & ~1) : (NEGS | kD);
                    S_up = Sl[r];
                    I_up = inb ? (Iv_t & ~1) : (NEGS | kI);
                    SnB[r] = max(SnB[r], Sl[r] | tpri);  // banded.rs:655-660, per block of 16 steps
                    const uint32_t cell = bfi<16>((uint32_t)Dv_t << 4, bfi<1>((uint32_t)Iv_t, (uint32_t)kb));
                    // ring slot = the STEP (not the column): the 64 byte writes of a step hit 64 different banks (lanes are
                    // one bank apart), and no cell needs an address of its own
                    s_row[r * RING + ((uint32_t)t & (uint32_t)(RING - 1))] = (uint8_t)cell;
                    diag = left_S;
                }
                S_out = S_up;
                I_out = I_up;
                q_out = q;
                if (ll == LP - 1) bnd[j] = make_int4(S_up, I_up & ~15, NEGS, 0);  // (K3v2's clean I; fold fields: what a strip without a fold hands on)
            }
            c.q = wave_shl1z(c.q);
            c.S = wave_shl1z(c.S);
            c.I = wave_shl1z(c.I);
        };
        auto merge_rows = [&](const int t_end) {  // the block that ends at step t_end into (Sn, Ly)
            const int32_t nmj_end = (int32_t)n - (jlo + t_end - ll);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int32_t nb = SnB[r] & ~15;
                const bool up = nb > Sn[r];
                Ly[r] = up ? (uint32_t)(nmj_end + (SnB[r] & 15)) : Ly[r];
                Sn[r] = max(Sn[r], nb);
                SnB[r] = NEGS;
            }
        };
        // steps T1 .. T2: every lane of every pair that takes part in this strip has all its R rows inside their bands
        int T1 = -0x40000000, T2 = 0x40000000;  // (a pair that sits this strip out has no say)
        if (jhi >= jlo) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                T1 = wn[r] ? max(T1, cf[r] - jlo + ll) : 0x40000000;
                T2 = min(T2, cl[r] - jlo + ll);
            }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            T1 = max(T1, __shfl_xor(T1, o));
            T2 = min(T2, __shfl_xor(T2, o));
        }
        T1 = __builtin_amdgcn_readfirstlane(T1);
        T2 = __builtin_amdgcn_readfirstlane(T2);
        Chunk c_even = load_chunk(0), c_odd;
        for (int t0 = 0; t0 < nsteps_w; t0 += 2 * LP) {
            c_odd = load_chunk(t0 + LP);
            if (t0 >= T1 && t0 + 2 * LP - 1 <= T2) {  // (then t0 + 2 * LP <= nsteps_w as well)
#pragma unroll 2
                for (int t = t0; t < t0 + LP; t++) step(t, c_even, std::true_type{});
                c_even = load_chunk(t0 + 2 * LP);
#pragma unroll 2
                for (int t = t0 + LP; t < t0 + 2 * LP; t++) step(t, c_odd, std::true_type{});
            } else {
#pragma unroll 1
                for (int t = t0; t < min(t0 + LP, nsteps_w); t++) step(t, c_even, std::false_type{});
                c_even = load_chunk(t0 + 2 * LP);
#pragma unroll 1
                for (int t = t0 + LP; t < min(t0 + 2 * LP, nsteps_w); t++) step(t, c_odd, std::false_type{});
            }
            merge_rows(t0 + 2 * LP - 1);
            const int t_done = min(t0 + 2 * LP, nsteps_w);
            if ((t_done & (FLUSH - 1)) == 0) flush_tb(jlo + t_done - 1 - ll, jlo + t_done - 1 - ll - FLUSH, false);
        }
        {
            const int t_last = (nsteps_w & ~(FLUSH - 1)) - 1;
            flush_tb(0, t_last < 0 ? -0x40000000 : jlo + t_last - ll, true);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            if (act && cl[r] >= cf[r]) {
                gSn[i] = from_s(Sn[r] + sn_bias);
                gLy[i] = (int32_t)Ly[r];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next strip reads bnd / gSn of this one
    }
}

}  // namespace

void launch_fill2i(const BandArgs& a, dim3 grid, hipStream_t st) {
    if (a.ring32)
        banded_fill2i_kernel<BF2_R, BF2_LP, 32><<<grid, dim3(256), 0, st>>>(a);
    else
        banded_fill2i_kernel<BF2_R, BF2_LP, 64><<<grid, dim3(256), 0, st>>>(a);
}

}  // namespace bgband_dev
