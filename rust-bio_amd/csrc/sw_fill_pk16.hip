// K1p — sw_fill_pk16_kernel<R, LP>: the short-read variant of K1 (sw_fill.inc) for Aligner::local with
// MatchParams scoring (pairwise/mod.rs:597-843, 986-1009) when every reachable score fits 12 bits.
//
// Same anti-diagonal wavefront, same keys (score << 4 | priority, "first maximum wins" == one integer
// max), but every VGPR holds TWO pairs: the DP values of pair A in bits 0-15 and of pair B in bits 16-31,
// combined with the packed-math VALU ops of CDNA (v_pk_add_i16 clamp, v_pk_max_i16, v_pk_mad_u16): one
// instruction advances a cell of both pairs.  K1 is bound by VALU issue (about 46 lane-instructions per
// cell), so halving the instructions per cell is the lever; HBM traffic per cell is unchanged.
// The two pairs of a lane group ("couple") are pairs g of the wavefront jobs 2w and 2w+1; they share the
// lane-level control (column validity, the row that owns row m), so they must have equal lengths — a
// couple with different lengths is simply processed in two passes, each pair against itself.
// Traceback words: 3 cells x 5 bits (I extends | move << 1 | D extends << 4) per 16-bit half
// (SwGeom::tb_fmt == 1), same tiles as K1.
#include <type_traits>

#include "sw_kernels.h"

namespace bgsw {
namespace pk16 {

typedef uint32_t pk;  // two int16: [15:0] = first pair of the couple, [31:16] = second

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s(pk v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ u16x2 as_u(pk v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ pk bits(s16x2 v) { return __builtin_bit_cast(pk, v); }
__device__ __forceinline__ pk bits(u16x2 v) { return __builtin_bit_cast(pk, v); }
// v_pk_add_i16 clamp / v_pk_max_i16 / v_pk_add_u16 / v_pk_sub_u16 [clamp] / v_pk_mad_u16
__device__ __forceinline__ pk pk_adds(pk a, pk b) { return bits(__builtin_elementwise_add_sat(as_s(a), as_s(b))); }
__device__ __forceinline__ pk pk_max(pk a, pk b) { return bits(__builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ pk pk_add_u16(pk a, pk b) { return bits((u16x2)(as_u(a) + as_u(b))); }
__device__ __forceinline__ pk pk_sub_u16(pk a, pk b) { return bits((u16x2)(as_u(a) - as_u(b))); }
__device__ __forceinline__ pk pk_subs_u16(pk a, pk b) { return bits(__builtin_elementwise_sub_sat(as_u(a), as_u(b))); }  // max(a - b, 0)
__device__ __forceinline__ pk pk_mad_u16(pk a, pk b, pk c) { return bits((u16x2)(as_u(a) * as_u(b) + as_u(c))); }
__device__ __forceinline__ pk dup16(int32_t v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }
__device__ __forceinline__ pk bfi(pk mask, pk a, pk b) { return (a & mask) | (b & ~mask); }
// 0xffff in every half that is non-zero
__device__ __forceinline__ pk nz_mask(pk v, pk one) { return pk_sub_u16(pk_subs_u16(one, v), one); }

template <int LP>
__device__ __forceinline__ void scan_first_max(int ll, int64_t& v, uint32_t& idx) {
#pragma unroll
    for (int off = 1; off < LP; off <<= 1) {
        const int64_t ov = __shfl_up(v, off, LP);
        const uint32_t oi = (uint32_t)__shfl_up((int)idx, off, LP);
        if (ll >= off && !(v > ov)) {
            v = ov;
            idx = oi;
        }
    }
}

constexpr int32_t kFloor16 = -2048;  // 'minus infinity' of the 12-bit score range (scaled: 0x8000)

// FAST: the wavefronts whose pairs all end on the last row of a lane (m % R == 0, the usual equal-length
// short reads), first pass only.  !FAST: launched right after it on the same grid, picks up what FAST left —
// the other wavefronts, and the second pair of every couple whose lengths differ.  Both derive the split
// from the lengths alone, so no flags travel between the two launches.
template <int R, int LP, bool FAST>
__device__ __forceinline__ void sw_fill_pk16_body(const SwArgs& a) {
    constexpr int PW = 64 / LP;
    constexpr int NW = tb_words(R);
    constexpr int NH = (R + 2) / 3;  // 16-bit halves of traceback cells per pair and step
    constexpr bool NARROW = true, LOCAL = true;
    constexpr int SH = 4;
    constexpr int32_t NEGS = kFloor16 * 16;
    constexpr pk CLEAN = 0xfff0fff0u, FLOORK = 0x80008000u, ONE = 0x00010001u;
    static_assert(LP == 16 || LP == 32, "lanes per pair");
    static_assert(R >= 1 && R <= 12, "rows per lane");
    (void)NARROW;
    (void)LOCAL;
    // LDS, two uses that never overlap in time: during the fill, the traceback words of the current tile (a
    // lane's 64 bytes per pair, kTileStride dwords apart: 8-byte writes, 16-byte aligned reads); after it, the
    // packed rows parked for the epilogue
    constexpr int kTileStride = 20;
    constexpr int kParkDw = 64 * (4 * R + NH + 1), kTileDw = 2 * 64 * kTileStride;  // per wavefront
    constexpr int kWaveDw = kParkDw > kTileDw ? kParkDw : kTileDw;
    __shared__ __align__(16) pk s_lds[4 * kWaveDw];
    pk* const wave_lds = s_lds + (threadIdx.x >> 6) * kWaveDw;  // wavefronts of a block run unsynchronised

    const int lane = threadIdx.x & 63;
    const uint32_t wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((uint64_t)wv * 2 * PW >= a.n_pairs) return;  // wave-uniform
    const int g = lane / LP, ll = lane % LP;
    const SwScoring sc = a.sc;
    const SwGeom geo = a.g;
    auto scl = [](int32_t v) -> int32_t { return max(v, kFloor16) * 16; };
    const int32_t go_s = sc.go * 16, xs_s = scl(sc.xs);
    (void)xs_s;
    // Keys: score << 4 | move code << 1 | "opens".  The I and D values keep their move code in the low bits
    // (bit 0, the open/extend decision, is cleared when they are stored), so they enter the cell's maximum
    // as they are; bit 0 sits below the code and only ever decides between the two candidates of one value
    // (open wins ties, like the reference's strict '>' for extend, mod.rs:738,749).
    constexpr int32_t K_XS = C_XS << 1, K_MATCH = C_MATCH << 1, K_SUBST = C_SUBST << 1, K_INS = C_INS << 1,
                      K_DEL = C_DEL << 1, K_XP = C_XP << 1;
    constexpr pk NOFLAG = 0xfffefffeu;
    const pk GE = dup16(sc.ge * 16);
    const pk GOT_I = dup16(sc.go * 16 + K_INS + 1), GOT_D = dup16(sc.go * 16 + K_DEL + 1);
    const pk MISK = dup16((sc.mismatch * 16) | K_SUBST);
    const pk DELTA = dup16(((sc.match * 16) | K_MATCH) - ((sc.mismatch * 16) | K_SUBST));
    const pk XKEY = dup16(K_XP);  // xclip_j == 0 for local
    const pk C16 = dup16(16), C32 = dup16(32), C1024 = dup16(1024);

    // ---- the couple of this lane group
    const uint32_t pA = (2 * wv) * PW + g, pB = (2 * wv + 1) * PW + g;
    const bool okA = pA < a.n_pairs, okB = pB < a.n_pairs;
    uint32_t mA = 0, nA = 0, mB = 0, nB = 0;
    uint64_t xoA = 0, yoA = 0, xoB = 0, yoB = 0;
    if (okA) {
        xoA = a.x_off[a.pair0 + pA];
        yoA = a.y_off[a.pair0 + pA];
        mA = (uint32_t)(a.x_off[a.pair0 + pA + 1] - xoA);
        nA = (uint32_t)(a.y_off[a.pair0 + pA + 1] - yoA);
    }
    if (okB) {
        xoB = a.x_off[a.pair0 + pB];
        yoB = a.y_off[a.pair0 + pB];
        mB = (uint32_t)(a.x_off[a.pair0 + pB + 1] - xoB);
        nB = (uint32_t)(a.y_off[a.pair0 + pB + 1] - yoB);
    }
    const bool same = okA && okB && mA == mB && nA == nB;
    const bool two = okA && okB && !same;
    const bool m_last0 = __all(!okA || mA % R == 0);
    if (FAST && !m_last0) return;                  // wave-uniform
    if (!FAST && m_last0 && !__any(two)) return;   // FAST did it all
    const int pass_begin = (!FAST && m_last0) ? 1 : 0;
    const int pass_end = FAST ? 1 : (__any(two) ? 2 : 1);
    const uint64_t job_words = tb_job_words(geo.nstrips, geo.nsteps, NW);

    auto do_pass = [&](const int pass) {
        // pass 0: (A, B) if they agree, else (A, A); pass 1: (B, B) for the couples that did not
        const bool pair_ok = pass == 0 ? okA : two;
        const uint32_t P0 = pass == 0 ? pA : pB, P1 = (pass == 0 && !same) ? pA : pB;
        const uint32_t m = !pair_ok ? 0 : (pass == 0 ? mA : mB), n = !pair_ok ? 0 : (pass == 0 ? nA : nB);
        const uint8_t* x0 = a.x + (pass == 0 ? xoA : xoB);
        const uint8_t* y0 = a.y + (pass == 0 ? yoA : yoB);
        const uint8_t* x1 = a.x + ((pass == 0 && !same) ? xoA : xoB);
        const uint8_t* y1 = a.y + ((pass == 0 && !same) ? yoA : yoB);
        const uint32_t Q0 = pair_ok ? P0 : 0, Q1 = pair_ok ? P1 : 0;
        uint32_t* tb0 = (uint32_t*)a.tb + (size_t)(Q0 / PW) * job_words + ((Q0 % PW) * LP + ll) * 16u;
        uint32_t* tb1 = (uint32_t*)a.tb + (size_t)(Q1 / PW) * job_words + ((Q1 % PW) * LP + ll) * 16u;
        int32_t* aux0 = a.aux + (size_t)Q0 * geo.aux_stride;
        int32_t* aux1 = a.aux + (size_t)Q1 * geo.aux_stride;
        int32_t* gLx0 = aux0 + geo.off_Lx();
        int32_t* gLx1 = aux1 + geo.off_Lx();

        uint32_t m_w = m, n_w = n;  // wave-uniform loop bounds
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            m_w = max(m_w, (uint32_t)__shfl_xor((int)m_w, o));
            n_w = max(n_w, (uint32_t)__shfl_xor((int)n_w, o));
        }
        (void)m_w;
        n_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_w);  // tell the compiler it is wave-uniform
        const uint32_t nsteps_w = n_w ? n_w + LP - 1 : 0;

        int32_t fold0;
        uint32_t lx0;
        col0_fold(sc, m, fold0, lx0);
        int32_t Sn0 = sc.ys;  // row 0 of a local alignment: S(0,j) = 0 by YCLIP_PREFIX, Sn[0] stays 0, Ly[0] = n
        uint32_t Ly0 = n;
        const int32_t S0n = 0;
        const uint32_t sb0n = n ? row0_cell(sc, n).sbits : (uint32_t)TB_START;
        if (pair_ok && ll == 0) {
            const int32_t v = m ? (int32_t)col0_cell(sc, m, m, fold0).sbits : (int32_t)TB_START;
            aux0[0] = v;
            aux1[0] = v;
            if (m == 0) {  // otherwise Lx[0] travels with the row-m lane's first packed store
                *(uint8_t*)gLx0 = (uint8_t)lx0;
                *(uint8_t*)gLx1 = (uint8_t)lx0;
            }
        }

        const uint32_t rb = (uint32_t)ll * R;  // this lane owns rows rb+1 .. rb+R (single strip)
        const int32_t mrow = (int32_t)m - (int32_t)rb - 1;
        pk Sl[R], Dl[R], Il[R], SnR[R], SnB[R], Ly[R], px[R];
        uint64_t nib0S = 0, nib0I = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = rb + r + 1;
            px[r] = 0;
            Sl[r] = Dl[r] = Il[r] = SnR[r] = SnB[r] = FLOORK;
            Ly[r] = 0;
            if (pair_ok && i <= m) {
                px[r] = (uint32_t)x0[i - 1] | ((uint32_t)x1[i - 1] << 16);
                const Col0 c = col0_cell(sc, i, m, fold0);
                Sl[r] = dup16(scl(c.S));
                Il[r] = dup16(scl(c.I) | K_INS);
                nib0S |= (uint64_t)c.sbits << (4 * r);
                nib0I |= (uint64_t)c.ibits << (4 * r);
                SnR[r] = dup16(scl(c.S) + scl(sc.ys));  // mod.rs:667-670 (always taken: S(i,0) >= 0, ys == 0)
                Ly[r] = dup16((int32_t)n);
            }
        }
        pk diag0 = dup16((pair_ok && rb < m) ? scl(col0_S(sc, rb, m, fold0)) : NEGS);  // S(rb, 0)

        pk S_out = FLOORK, I_out = FLOORK, cm_out = FLOORK, ca_out = 0, q_out = 0;
        pk ychunk = 0, ychunk_nx = 0;
        pk hwlast[NH];  // traceback halves of this lane's last column
#pragma unroll
        for (int k = 0; k < NH; k++) hwlast[k] = 0;
        pk lx_n = dup16((int32_t)lx0);
        // VMEM store instructions are the scarce resource of this kernel (four per step cost 5 ms of 17 on
        // 1M x 150 bp), and 64-byte slots filled 8 or 16 bytes at a time get evicted half-written (1.5x HBM
        // write traffic): a lane's traceback words collect in LDS and leave as its complete 64-byte slot of the
        // tile, four back-to-back 16-byte stores per pair every 16 / NW steps; Lx[j] (<= m < 256) leaves as bytes,
        // four columns per dword store
        pk* tileA = wave_lds + (0 * 64 + lane) * kTileStride;
        pk* tileB = wave_lds + (1 * 64 + lane) * kTileStride;
        uint32_t lxaccA = lx0 << 24, lxaccB = lx0 << 24;  // Lx[0] ends up in byte 0 of the first dword
        if (pair_ok && (uint32_t)ll < n) ychunk_nx = (uint32_t)y0[ll] | ((uint32_t)y1[ll] << 16);

        // Sn[i]/Ly[i] (mod.rs:799-802, "first maximum of the row") per block of 16 steps: inside a block the
        // key best | (15 - step % 16) lets one packed max keep the earliest column; the block's winner is
        // folded into the running maximum (strictly greater only) when the block ends.
        auto merge_rows = [&](uint32_t blockbase) {
            const pk lybase = dup16((int32_t)(n - blockbase - 16u + (uint32_t)ll));
#pragma unroll
            for (int r = 0; r < R; r++) {
                const pk nr = pk_max(SnR[r], SnB[r] & CLEAN);
                const pk msk = nz_mask(nr ^ SnR[r], ONE);
                Ly[r] = bfi(msk, pk_add_u16(lybase, SnB[r] & 0x000f000fu), Ly[r]);
                SnR[r] = nr;
                SnB[r] = FLOORK;
            }
        };

        constexpr uint32_t tsteps = tb_tile_steps(NW);
        auto store_tile = [&](uint32_t s_any) {  // the tile that holds step s_any, see tb_word_off()
            if (!pair_ok) return;
            const uint32_t off = (s_any / tsteps) * 1024u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                *(uint4*)&tb0[off + 4 * k] = *(const uint4*)&tileA[4 * k];
                *(uint4*)&tb1[off + 4 * k] = *(const uint4*)&tileB[4 * k];
            }
        };
        // MRk >= 0: every pair of this wavefront has its row m at r == MRk of its owner lane (equal read
        // lengths, the usual short-read case): the x-suffix-clip candidate drops out of the other rows
        auto run_steps = [&](auto mr_tag) {
        constexpr int MRk = decltype(mr_tag)::value;
        for (uint32_t s = 0; s < nsteps_w; s++) {
            if ((s & (LP - 1)) == 0) {  // wave-uniform: next LP columns of y
                ychunk = ychunk_nx;
                const uint32_t nb = s + LP + ll;
                if (pair_ok && nb < n) ychunk_nx = (uint32_t)y0[nb] | ((uint32_t)y1[nb] << 16);
            }
            // lane 0 of the wavefront keeps its own value ("old" == source): it is a row-0 lane and overrides it
            pk S_up = (pk)wave_shr1((int)S_out, (int)S_out), I_up = (pk)wave_shr1((int)I_out, (int)I_out);
            pk cm = (pk)wave_shr1((int)cm_out, (int)cm_out), ca = (pk)wave_shr1((int)ca_out, (int)ca_out);
            pk q = (pk)wave_shr1((int)q_out, (int)q_out);
            const uint32_t j = s + 1 - (uint32_t)ll;
            const bool col_ok = pair_ok && (j - 1) < n;  // 1 <= j <= n
            if (ll == 0) {  // row 0 of the matrix (mod.rs:678-721) for a local alignment
                q = ychunk;
                S_up = 0;
                I_up = FLOORK;
                cm = FLOORK;
                ca = 0;
            }
            ychunk = (pk)wave_shl1((int)ychunk, (int)ychunk);

            if (col_ok) {
                const pk sprio = dup16((int32_t)(15u - (s & 15u)));
                pk diag = diag0;
                diag0 = S_up;
                pk hw[NH];
                const pk ca_in = ca;
                pk cmk = cm | 0x000f000fu;  // fold key: clean running maximum | row priority (15 = an earlier lane)
                pk snap = cmk;               // the fold as row m sees it
#pragma unroll
                    for (int r = 0; r < R; r++) {  // rows past m compute garbage nobody reads
                        const bool maybe_m = MRk < 0 || r == MRk;
                        const bool is_m = MRk < 0 ? (r == mrow) : (r == MRk && mrow == MRk);
                        // mod.rs:733-755: substitution score with the MATCH/SUBST move code in its low bits
                        const pk e = pk_subs_u16(ONE, px[r] ^ q);  // 1 where the characters agree
                        const pk m_key = pk_adds(diag, pk_mad_u16(e, DELTA, MISK));
                        const pk Iv_t = pk_max(pk_adds(I_up, GE), pk_adds(S_up, GOT_I));
                        const pk Dv_t = pk_max(pk_adds(Dl[r], GE), pk_adds(Sl[r], GOT_D));
                        const pk Iv = Iv_t & NOFLAG, Dv = Dv_t & NOFLAG;
                        // mod.rs:757-786: first maximum wins == max over (score | priority)
                        pk kb = pk_max(pk_max(m_key, Dv_t), XKEY);
                        kb = pk_max(kb, Iv_t);  // the value that waits for the row above comes last
                        if (maybe_m) kb = pk_max(kb, is_m ? ((cmk & CLEAN) | (0x00010001u * K_XS)) : FLOORK);
                        const pk best = kb & CLEAN;
                        diag = Sl[r];
                        Sl[r] = best;
                        Dl[r] = Dv;
                        Il[r] = Iv;
                        S_up = best;
                        I_up = Iv;
                        if (maybe_m) snap = is_m ? cmk : snap;
                        cmk = pk_max(cmk, best | (0x00010001u * (uint32_t)(14 - r)));  // mod.rs:793-796
                        SnB[r] = pk_max(SnB[r], best | sprio);                           // mod.rs:799-802
                        // packed cell: I opens | 3-bit move << 1 | D opens << 4 (flipped to "extends" below)
                        const pk c5 = pk_mad_u16(Dv_t & ONE, C16, (kb & 0x000e000eu) | (Iv_t & ONE));
                        if (r % 3 == 0)
                            hw[r / 3] = c5;
                        else
                            hw[r / 3] = pk_mad_u16(c5, r % 3 == 1 ? C32 : C1024, hw[r / 3]);
                    }
                // Lx[j] = rows between the fold's winner and row m (mod.rs:793-796): priority 15 = a lane above
                const pk mbase = dup16(mrow - 14);
                const pk lo = cmk & 0x000f000fu, slo = snap & 0x000f000fu;
                if (mrow >= 0 && mrow < R) {  // the lane that owns row m publishes Lx[j]
                    const pk is15 = nz_mask(slo ^ 0x000f000fu, ONE);  // 0xffff where slo != 15
                    lx_n = bfi(is15, pk_add_u16(slo, mbase), ca_in);
                    lxaccA = __builtin_amdgcn_perm(lx_n, lxaccA, 0x04030201u);  // acc >> 8 | Lx_A << 24
                    lxaccB = __builtin_amdgcn_perm(lx_n, lxaccB, 0x06030201u);  // acc >> 8 | Lx_B << 24
                    if ((j & 3u) == 3u || j == n) {
                        const uint32_t sh = 8u * (3u - (j & 3u));  // the last dword of the row may be partial
                        ((uint32_t*)gLx0)[j >> 2] = lxaccA >> sh;
                        ((uint32_t*)gLx1)[j >> 2] = lxaccB >> sh;
                    }
                }
                {
                    const pk is15 = nz_mask(lo ^ 0x000f000fu, ONE);
                    ca = bfi(is15, pk_add_u16(lo, mbase), ca_in);
                }
                cm = cmk & CLEAN;
                // 17 = both "opens" bits: stored as "extends" like K1's cells
                constexpr pk INV = 0x00010001u * (17u | (17u << 5) | (17u << 10));
                constexpr pk INV_LAST = 0x00010001u * ((R % 3 == 0) ? (17u | (17u << 5) | (17u << 10))
                                                      : (R % 3 == 1) ? 17u : (17u | (17u << 5)));
#pragma unroll
                for (int k = 0; k < NH; k++) {
                    hw[k] ^= (k == NH - 1) ? INV_LAST : INV;
                    hwlast[k] = hw[k];
                }
                const pk h0 = hw[0], h1 = NH > 1 ? hw[NH > 1 ? 1 : 0] : 0u;
                const uint32_t slot = (s % tsteps) * NW;
                if (NW == 1) {
                    tileA[slot] = __builtin_amdgcn_perm(h1, h0, 0x05040100u);
                    tileB[slot] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
                } else {
                    const pk h2 = hw[NH > 2 ? 2 : 0], h3 = NH > 3 ? hw[NH > 3 ? 3 : 0] : 0u;
                    *(uint2*)&tileA[slot] = make_uint2(__builtin_amdgcn_perm(h1, h0, 0x05040100u),
                                                       __builtin_amdgcn_perm(h3, h2, 0x05040100u));
                    *(uint2*)&tileB[slot] = make_uint2(__builtin_amdgcn_perm(h1, h0, 0x07060302u),
                                                       __builtin_amdgcn_perm(h3, h2, 0x07060302u));
                }
                S_out = S_up;
                I_out = I_up;
                cm_out = cm;
                ca_out = ca;
                q_out = q;
            }
            if (s % tsteps == tsteps - 1) store_tile(s);  // wave-uniform
            if ((s & 15u) == 15u) merge_rows(s & ~15u);  // wave-uniform
        }
        };
        run_steps(std::integral_constant<int, FAST ? R - 1 : -1>{});
        if (nsteps_w % tsteps) store_tile(nsteps_w - 1);  // the last, partial tile
        if (nsteps_w) merge_rows((nsteps_w - 1) & ~15u);

        // =========== epilogue of the last column (mod.rs:808-843), one pair of the couple at a time ===========
        // The packed rows are parked in LDS (thread-private slots, stride 64: conflict-free) so that the
        // epilogue's 64-bit scans do not have to share the register file with them.
        pk* park = wave_lds + lane;
#pragma unroll
        for (int r = 0; r < R; r++) {
            park[(0 * R + r) * 64] = Sl[r];
            park[(1 * R + r) * 64] = Il[r];
            park[(2 * R + r) * 64] = SnR[r];
            park[(3 * R + r) * 64] = Ly[r];
        }
#pragma unroll
        for (int k = 0; k < NH; k++) park[(4 * R + k) * 64] = hwlast[k];
        park[(4 * R + NH) * 64] = lx_n;
        const int nhalf = (pair_ok && P1 != P0) ? 2 : 1;
        const int nhalf_w = __any(nhalf == 2) ? 2 : 1;
#pragma unroll 1
        for (int h = 0; h < nhalf_w; h++) {
            if (h >= nhalf) continue;  // the shuffles below only pair lanes of one group, whose nhalf agree
            const uint32_t hs = 16u * (uint32_t)h;
            int32_t* aux = h ? aux1 : aux0;
            int32_t* gLy = aux + geo.off_Ly();
            uint8_t* gBits = (uint8_t*)(aux + geo.off_bits());
            const uint32_t strip = 0;
            // unpacked (sign-extended, still scaled by 16) views of this pair's half
            struct HalfS {
                const pk* p;
                uint32_t hs;
                __device__ int32_t operator[](int r) const { return (int32_t)(int16_t)(p[r * 64] >> hs); }
            };
            struct HalfU {
                const pk* p;
                uint32_t hs;
                __device__ uint32_t operator[](int r) const { return (p[r * 64] >> hs) & 0xffffu; }
            };
            int32_t Sl_u[R];  // the epilogue overwrites S(i, n)
#pragma unroll
            for (int r = 0; r < R; r++) Sl_u[r] = (int32_t)(int16_t)(park[r * 64] >> hs);
            const HalfS Il_u{park + 1 * R * 64, hs}, Sn_u{park + 2 * R * 64, hs};
            const HalfU Ly_u{park + 3 * R * 64, hs};
            const uint32_t lx_n_u = (park[(4 * R + NH) * 64] >> hs) & 0xffffu;
            int64_t e_carry = INT64_MIN;
            uint32_t sbf_carry = TB_START, sb2_carry = TB_START;
            int64_t c1v = INT64_MIN, c2v = INT64_MIN;
            uint32_t c1i = 0, c2i = 0;
            auto cell_of = [&](int r) -> uint32_t {  // K1's bit order: move | I extends << 3 | D extends << 4
                const uint32_t c = (park[(4 * R + r / 3) * 64] >> (hs + 5 * (r % 3))) & 31u;
                return ((c >> 1) & 7u) | ((c & 1u) << 3) | (c & 16u);
            };
            {
                // the names the shared body expects
                int32_t(&Sl)[R] = Sl_u;
                const HalfS& Il = Il_u;
                const HalfS& Sn = Sn_u;
                const HalfU& Ly = Ly_u;
                const uint32_t lx_n = lx_n_u;
#include "sw_epilogue.inc"
            }
        }
    };
    if (FAST) {
        do_pass(0);
    } else {
#pragma unroll 1
        for (int pass = pass_begin; pass < pass_end; pass++) do_pass(pass);
    }
}

// the fast launch is tuned for three wavefronts per SIMD up to R = 10 (168 VGPRs, 46 KB of LDS per block)
template <int R, int LP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R <= 10 ? 3 : 2, R <= 10 ? 3 : 2))) void sw_fill_pk16_kernel(const SwArgs a) {
    sw_fill_pk16_body<R, LP, true>(a);
}
template <int R, int LP>
__global__ __launch_bounds__(256) void sw_fill_pk16_rest_kernel(const SwArgs a) {
    sw_fill_pk16_body<R, LP, false>(a);
}

}  // namespace pk16

sw_fill_fn get_fill_pk16(int lp, int r, bool fast) {
#define CASE(LP, R) \
    if (lp == LP && r == R) return fast ? pk16::sw_fill_pk16_kernel<R, LP> : pk16::sw_fill_pk16_rest_kernel<R, LP>;
    CASE(16, 2) CASE(16, 3) CASE(16, 4) CASE(16, 5) CASE(16, 6) CASE(16, 7) CASE(16, 8) CASE(16, 9) CASE(16, 10)
    CASE(16, 11) CASE(16, 12)
    CASE(32, 7) CASE(32, 8) CASE(32, 9) CASE(32, 10) CASE(32, 11) CASE(32, 12)
#undef CASE
    return nullptr;
}

}  // namespace bgsw
