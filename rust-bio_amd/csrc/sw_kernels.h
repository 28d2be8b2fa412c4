// Kernels K1 (sw_fill) and K2 (sw_traceback): batched `Aligner::custom` on gfx950.
//
// Reference semantics: /root/reference/src/alignment/pairwise/mod.rs
//   prologue 597-672, fill 674-806, epilogue 808-843, traceback 845-921.
//
// K1 — anti-diagonal wavefront.  A group of LP lanes (16/32/64) owns one pair; lane ll of the
//   group owns R consecutive rows i = strip*LP*R + ll*R + 1 .. +R of the DP matrix and walks
//   the columns skewed by one step per lane (lane ll is at column j = s - ll + 1 at step s),
//   so the value a lane needs from the row above — S(i-1,j), I(i-1,j), the running
//   x-suffix-clip maximum of the column (`S[curr][m]`/`Lx[j]` in the reference) and the
//   column's y character — is exactly what its neighbour lane produced one step earlier:
//   five DPP `wave_shr:1` moves per step, no LDS round trip.  S(i,j-1), D(i,j-1), Sn[i],
//   Ly[i] stay in the lane's registers.  Sequences longer than LP*R rows are processed in
//   strips; the last row of a strip is handed to the next strip through a small global
//   row buffer that is read back in LP-column chunks and walked with `wave_shl:1`.
//   The traceback matrix is written as one packed word per lane per step — R cells x 5 bits
//   (3-bit S move, 1 bit "I extends", 1 bit "D extends") — i.e. a fully coalesced
//   64-lane store in anti-diagonal order, 5 bits/cell instead of the reference's 16.
//   The I/D nibbles of the reference cell are either the constant INS/DEL or a copy of a
//   neighbour's S nibble (mod.rs:743,754), so one bit each reproduces them at traceback time.
// K2 — one lane per pair: the serial epilogue of the last column (mod.rs:808-843), then the
//   pointer-chasing traceback (mod.rs:845-921), writing one byte per operation backwards
//   into the pair's slot of the ops buffer.
// MFMA is unused: the recurrence is integer max-plus with data-dependent tie-breaking.
#ifndef BG_SW_KERNELS_H
#define BG_SW_KERNELS_H
#include <type_traits>

#include "bg_common.h"

namespace bgsw {

constexpr int32_t NEG = BG_MIN_SCORE;

// reference traceback codes (mod.rs:1036-1045)
enum : uint32_t {
    TB_START = 0, TB_INS = 1, TB_DEL = 2, TB_SUBST = 3, TB_MATCH = 4,
    TB_XCLIP_PREFIX = 5, TB_XCLIP_SUFFIX = 6, TB_YCLIP_PREFIX = 7, TB_YCLIP_SUFFIX = 8
};
// 3-bit S-move code of a packed cell; bit 3 = I extends, bit 4 = D extends
// (numbered by priority: on equal scores the reference keeps the earlier candidate, mod.rs:757-786)
// MATCH and SUBST are told apart in the cell (only one of them is a candidate of a given cell), so
// that K2 never has to re-read the sequences.
enum : uint32_t { C_XS = 7, C_MATCH = 6, C_SUBST = 5, C_INS = 3, C_DEL = 2, C_XP = 1, C_YP = 0 };
// NARROW kernels keep scores scaled by 16 and order candidates with one integer max over
// (score << 4 | priority) keys; real scores must stay inside +-2^25, 'minus infinity' style values
// are clamped to this floor (scaled: -2^30, so that floor + floor still fits an int32).
constexpr int32_t kNarrowFloor = -(1 << 26);
constexpr int32_t kNarrowKeyFloor = (int32_t)0x80000000;

enum { SCORE_PARAMS = 0, SCORE_LDS = 1, SCORE_GLOBAL = 2 };
constexpr int kMaxLdsAlphabet = 64;

struct SwScoring {
    int32_t go, ge, xp, xs, yp, ys;
    int32_t match, mismatch;
};

// aux record of one pair (int32 words):
//   [0] S nibble of (m,0)   [1] score = S[n%2][m] after the epilogue   [2] Lx[n] after the epilogue
//   Ly[m_cap+1]  Lx[n_cap+1]  colBits[m_cap+1 bytes]: (S nibble | I nibble << 4) of column n
struct SwGeom {
    uint32_t lp, r, nsteps, nstrips, m_cap, n_cap, aux_stride;
    uint32_t tb_fmt;  // 0: six 5-bit cells per word (K1); 1: three per 16-bit half (K1p, sw_fill_pk16.inc); 2: as 1, move code 0 == C_XP (K1p LF)
    uint32_t r_inv;     // ceil(2^32 / r): (row * r_inv) >> 32 == row / r for row < 2^24 (K2 divides per traceback step)
    uint32_t lp_shift;  // log2(lp)
    __host__ __device__ uint32_t off_Ly() const { return 4; }
    __host__ __device__ uint32_t off_Lx() const { return 4 + (m_cap + 1); }
    __host__ __device__ uint32_t off_bits() const { return 4 + (m_cap + 1) + (n_cap + 1); }
    __host__ __device__ static uint32_t stride_for(uint32_t m_cap, uint32_t n_cap) {
        return 4 + (m_cap + 1) + (n_cap + 1) + (m_cap + 1 + 3) / 4;
    }
};

// A pair longer than the bounds the caller stated (max_xlen / max_ylen) has no scratch: the fill kernels treat
// it as an empty pair, K2 writes a record with status BG_ERR_INVALID_ARG for it.
__device__ __forceinline__ bool len_over(const SwGeom& g, uint32_t m, uint32_t n) { return m > g.m_cap || n > g.n_cap; }

// packed traceback: R cells x 5 bits per lane per step, 6 cells per 32-bit word
__host__ __device__ constexpr int tb_words(int r) { return (r + 5) / 6; }
// Traceback words of one wavefront job are stored in tiles of kTbTile(NW) steps: tile t holds, for each
// of the 64 lanes, the lane's 64 bytes for those steps (tile = 4 KB).  A lane's cells of one diagonal
// run stay contiguous for K2, and every 128-byte line is completed within one tile's steps by two
// lanes, so L2 write-combines it (a pure lane-major layout was measured at 2.9x HBM write traffic,
// a step-major one slows K2's dependent loads down by 1.75x).
__host__ __device__ constexpr uint32_t tb_tile_steps(int nw) { return 16u / (uint32_t)nw; }
__host__ __device__ inline uint64_t tb_job_words(uint32_t nstrips, uint32_t nsteps, int nw) {
    const uint64_t g = (uint64_t)nstrips * nsteps, t = tb_tile_steps(nw);
    return (g + t - 1) / t * 1024ull;
}
// word offset of (linear step g, lane) inside the job's block
__host__ __device__ inline uint64_t tb_word_off(uint64_t g, uint32_t lane, int nw) {
    const uint32_t t = tb_tile_steps(nw);
    return (g / t) * 1024ull + lane * 16u + (uint32_t)(g % t) * (uint32_t)nw;
}

struct SwArgs {
    const uint8_t* x;
    const uint64_t* x_off;
    const uint8_t* y;
    const uint64_t* y_off;
    uint64_t pair0;    // first pair of this sub-batch in the offset arrays
    uint32_t n_pairs;  // pairs in this sub-batch
    SwScoring sc;
    const int32_t* table;     // compacted scoring table A x A (SCORE_LDS / SCORE_GLOBAL)
    const uint8_t* code_map;  // byte -> code
    int32_t alpha;
    void* tb;
    int32_t* aux;
    int4* bnd;  // strip hand-over rows: per pair (n_cap+1) x {S, I, cmax, carg}
    // K1p couples pairs of equal lengths: with ragged batches the pairs are visited in (m, n) order — slot s of the
    // sub-batch holds pair perm[s]; traceback words and aux records are per slot, results per pair.  NULL: identity.
    const uint32_t* perm;
    const uint32_t* len_stats;  // {min m, min n, max m, max n} of the sub-batch: perm applies only if they differ (device-side decision)
    // K2 is launched in both flavours; n_eff[0] / n_eff[1] = pairs the identity / the permuted one has to do (one of
    // them 0), written on the device once the lengths are known.  NULL: identity, n_pairs.
    const uint32_t* n_eff;
    __device__ const uint32_t* slot_perm() const {
        if (!perm) return nullptr;
        if (len_stats && len_stats[0] == len_stats[2] && len_stats[1] == len_stats[3]) return nullptr;
        return perm;
    }
    SwGeom g;
    // K2 outputs
    bg_alignment_t* out;
    uint8_t* ops;
    uint64_t ops_stride;
    int32_t mode;         // BG_MODE_* recorded in the result
    int32_t filter_clips; // semiglobal/local drop Xclip/Yclip ops (mod.rs:974,1006)
    // x / y are 2-bit streams (pack2.hip: 16 symbols per dword, offsets in symbols) instead of bytes — K1p only
    int32_t packed;
};
// symbol s of a 2-bit stream
__device__ __forceinline__ uint32_t sym2(const uint8_t* stream, uint64_t s) {
    return (((const uint32_t*)stream)[s >> 4] >> (2 * ((uint32_t)s & 15u))) & 3u;
}

// ---- closed forms of the matrix borders (mod.rs:622-671 column 0, 678-717 row 0) -------------
struct Col0 {
    int32_t S, I;
    uint32_t sbits, ibits;
};
// cell (i,0), 1 <= i; `fold` = running S[k][m] (only read when i == m)
__device__ __forceinline__ Col0 col0_cell(const SwScoring& sc, uint32_t i, uint32_t m, int32_t fold) {
    Col0 c;
    if (i == 1) {
        c.I = sc.go;
        c.ibits = TB_START;
    } else {
        const int32_t i_score = sc.go + sc.ge * ((int32_t)i - 1);
        const int32_t c_score = sc.xp + sc.go;
        if (i_score > c_score) {
            c.I = i_score;
            c.ibits = TB_INS;
        } else {
            c.I = c_score;
            c.ibits = TB_XCLIP_PREFIX;
        }
    }
    if (i == m) {
        c.S = fold;
        c.sbits = TB_XCLIP_SUFFIX;
    } else {
        c.S = NEG;
        c.sbits = TB_START;
    }
    if (c.I > c.S) {
        c.S = c.I;
        c.sbits = TB_INS;
    }
    if (sc.xp > c.S) {
        c.S = sc.xp;
        c.sbits = TB_XCLIP_PREFIX;
    }
    return c;
}
// running x-suffix-clip fold of column 0 over rows 1..m-1 (mod.rs:658-661).  S(i,0) is
// non-increasing in i (gap_extend <= 0, xclip_prefix <= 0), so only i == 1 can fire.
__device__ __forceinline__ void col0_fold(const SwScoring& sc, uint32_t m, int32_t& fold, uint32_t& lx0) {
    fold = NEG;
    lx0 = 0;
    if (m >= 2) {
        const Col0 c = col0_cell(sc, 1, m, NEG);
        if (c.S + sc.xs > fold) {
            fold = c.S + sc.xs;
            lx0 = m - 1;
        }
    }
}
// S(i,0) for any 0 <= i <= m
__device__ __forceinline__ int32_t col0_S(const SwScoring& sc, uint32_t i, uint32_t m, int32_t fold0) {
    if (i == 0) return 0;
    return col0_cell(sc, i, m, fold0).S;
}

struct Row0 {
    int32_t S, D;
    uint32_t sbits, dbits;
};
// cell (0,j), j >= 1, before the Sn[0] bookkeeping of mod.rs:706-714
__device__ __forceinline__ Row0 row0_cell(const SwScoring& sc, uint32_t j) {
    Row0 c;
    if (j == 1) {
        c.D = sc.go;
        c.dbits = TB_START;
    } else {
        const int32_t d_score = sc.go + sc.ge * ((int32_t)j - 1);
        const int32_t c_score = sc.yp + sc.go;
        if (d_score > c_score) {
            c.D = d_score;
            c.dbits = TB_DEL;
        } else {
            c.D = c_score;
            c.dbits = TB_YCLIP_PREFIX;
        }
    }
    if (c.D > sc.yp) {
        c.S = c.D;
        c.sbits = TB_DEL;
    } else {
        c.S = sc.yp;
        c.sbits = TB_YCLIP_PREFIX;
    }
    return c;
}

// reference S nibble of a packed 3-bit move code
__device__ __forceinline__ uint32_t s_nibble_of_code(uint32_t code) {
    switch (code) {
        case C_MATCH: return TB_MATCH;  // mod.rs:762
        case C_SUBST: return TB_SUBST;
        case C_INS: return TB_INS;
        case C_DEL: return TB_DEL;
        case C_XP: return TB_XCLIP_PREFIX;
        case C_YP: return TB_YCLIP_PREFIX;
        default: return TB_XCLIP_SUFFIX;  // mod.rs:757
    }
}

typedef void (*sw_fill_fn)(const SwArgs);

}  // namespace bgsw
#endif
