// K2 — sw_traceback_kernel: one lane per pair.  Pointer-chasing traceback
// (/root/reference/src/alignment/pairwise/mod.rs:845-921) over the packed anti-diagonal
// traceback words and the last-column nibbles K1 wrote.  Design notes: sw_kernels.h.
#include "sw_kernels.h"

namespace bgsw {

constexpr uint32_t kSegStride = 20;  // dwords per thread slot: 16 of data, padded against bank conflicts

// PERM: the slots of the sub-batch hold the pairs in (m, n) order (SwArgs::perm).  Whether that order is in force may
// be decided on the device, so both instantiations are launched, each with its pair count read from SwArgs::n_eff
// (one of the two is 0) — a single kernel that merely might go through the indirection was measured at 2.0 ms
// against 1.4.
template <int NW, bool PERM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void sw_traceback_kernel(const SwArgs a) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;  // traceback words and aux records are per slot
    if (slot >= (a.n_eff ? a.n_eff[PERM ? 1 : 0] : a.n_pairs)) return;
    const uint32_t pair = PERM ? a.perm[slot] : slot;              // sequences and results per pair (sw_kernels.h)
    const SwScoring sc = a.sc;
    const SwGeom geo = a.g;
    const uint32_t R = geo.r, LP = geo.lp, PW = 64 / geo.lp;

    const uint64_t xo = a.x_off[a.pair0 + pair], yo = a.y_off[a.pair0 + pair];
    const uint32_t m = (uint32_t)(a.x_off[a.pair0 + pair + 1] - xo);
    const uint32_t n = (uint32_t)(a.y_off[a.pair0 + pair + 1] - yo);
    if (len_over(geo, m, n)) {  // longer than the max_xlen / max_ylen the caller stated: its scratch does not exist
        bg_alignment_t bad = {};
        bad.xlen = m;
        bad.ylen = n;
        bad.ops_off = (a.pair0 + pair + 1) * a.ops_stride;
        bad.mode = (uint8_t)a.mode;
        bad.status = (int8_t)BG_ERR_INVALID_ARG;
        a.out[a.pair0 + pair] = bad;
        return;
    }

    const int32_t* aux = a.aux + (size_t)slot * geo.aux_stride;
    const int32_t* gLy = aux + geo.off_Ly();
    const int32_t* gLx = aux + geo.off_Lx();
    const uint8_t* bits = (const uint8_t*)(aux + geo.off_bits());  // column n: S | I << 4
    const uint32_t sbits_m0 = (uint32_t)aux[0];
    const int32_t score = aux[1];
    const uint32_t lxn = (uint32_t)aux[2];

    const uint32_t job = slot / PW, grp = slot % PW;
    const uint32_t* tbj = (const uint32_t*)a.tb + (size_t)job * tb_job_words(geo.nstrips, geo.nsteps, NW);
    // A diagonal step stays in the same lane's stream nine times out of ten (R = 10) and moves one step
    // back in it: the 64 bytes a lane wrote for 16 / NW consecutive steps (one tile row, tb_word_off) are
    // fetched once into this thread's LDS slot and the following cells are served from there — one global
    // round trip per ~8 cells of the path instead of one per cell.
    // (Round 4 built the prefetch DESIGN.md had sketched: every miss also fetched the slot a diagonal walk needs next
    //  — previous tile row of the lane, or the lane above — into 16 registers, and a miss that found its slot there
    //  copied it from registers.  Measured on 1 M x 150 bp: 1.69 ms against 1.39 — the 16 registers cost the fifth
    //  wavefront per SIMD (102 -> 128 VGPRs; with five and spills: 3.85 ms), and the second load per miss doubles the
    //  requests of a kernel that already sits near the gather rate.  Removed.)
    __shared__ __align__(16) uint32_t s_seg[256 * kSegStride];
    uint32_t* seg = s_seg + threadIdx.x * kSegStride;
    uint64_t seg_tag = ~0ull;
    // packed 5 bits of inner cell (1 <= i <= m, 1 <= j <= n)
    auto cell = [&](uint32_t i, uint32_t j) -> uint32_t {
        const uint32_t i1 = i - 1, lrow = __umulhi(i1, geo.r_inv), rr = i1 - lrow * R;  // i1 / R, exact (sw_kernels.h)
        const uint32_t st = lrow >> geo.lp_shift, llc = lrow & (LP - 1);
        const uint64_t g = (uint64_t)st * geo.nsteps + (j - 1 + llc);
        constexpr uint32_t tsteps = 16 / NW;
        const uint64_t base = (g / tsteps) * 1024ull + (grp * LP + llc) * 16u;
        if (base != seg_tag) {
#pragma unroll
            for (int q = 0; q < 4; q++) *(uint4*)&seg[4 * q] = *(const uint4*)&tbj[base + 4 * q];
            seg_tag = base;
        }
        const uint32_t w = seg[(uint32_t)(g % tsteps) * NW + rr / 6];
        const uint32_t c = rr % 6;
        if (!geo.tb_fmt) return (w >> (5 * c)) & 31u;
        if (geo.tb_fmt == 3) {  // K1's LF flavour: tb_fmt 0's packing, each cell I opened | move << 1 | D opened << 4
            const uint32_t u = (w >> (5 * c)) & 31u;
            return ((u >> 1) & 7u) | ((~u & 1u) << 3) | (~u & 16u);
        }
        // tb_fmt 1 (K1p): three cells per 16-bit half, each I extends | move << 1 | D extends << 4
        // tb_fmt 2 (K1p, LF flavour): the same, and move code 0 stands for C_XP (the floor of a local alignment)
        const uint32_t v = (w >> (5 * (c % 3) + 16 * (c / 3))) & 31u;
        const uint32_t mv = (v >> 1) & 7u;
        return ((geo.tb_fmt == 2 && mv == 0) ? (uint32_t)C_XP : mv) | ((v & 1u) << 3) | (v & 16u);
    };
    // S nibble of cell (i,j), j < n (or the fill-time value for j == n, never requested)
    auto s_fill = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == 0) {
            if (i == 0) return TB_START;
            if (i == m) return sbits_m0;
            return col0_cell(sc, i, m, NEG).sbits;
        }
        if (i == 0) return row0_cell(sc, j).sbits;
        return s_nibble_of_code(cell(i, j) & 7u);
    };
    // column n comes from K1's epilogue
    auto s_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == n) return (uint32_t)bits[i] & 15u;
        return s_fill(i, j);
    };
    auto i_nib = [&](uint32_t i, uint32_t j) -> uint32_t {
        if (j == n) return (uint32_t)bits[i] >> 4;  // INS, or a copy of (i-1, n)'s S nibble
        if (i == 0) return TB_START;
        if (j == 0) return col0_cell(sc, i, m, NEG).ibits;
        return (cell(i, j) & 8u) ? TB_INS : s_fill(i - 1, j);  // mod.rs:740-743
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
    // ops are produced back to front; four of them are collected into one aligned dword store
    const bool dword_ok = ops_end && (((uintptr_t)ops_end & 3) == 0);
    uint32_t acc = 0;
    auto push = [&](uint32_t op) {
        ++n_ops;
        if (!ops_end) return;
        if (dword_ok) {
            acc = (acc << 8) | op;
            if ((n_ops & 3) == 0) *(uint32_t*)(ops_end - n_ops) = acc;
        } else {
            ops_end[-(int64_t)n_ops] = (uint8_t)op;
        }
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
    uint32_t layer = s_nib(m, n);
    const uint32_t guard = 2 * (m + n) + 16;
    // Moves between interior cells (1 <= i, 1 <= j < n, target i', j' >= 1) — nearly all of a path — take one common
    // sequence instead of the per-move cases below, which a wavefront whose 64 paths are at different moves walks one
    // after the other: the packed cell of the position is kept from the step that arrived there, the target's cell is
    // looked up once per step whatever the move (mod.rs:857-877: INS / DEL continue in their layer while the cell says
    // "extended", everything else continues with the S move of the cell it steps to).
    uint32_t c_here = 0;
    bool have_c = false;
    for (uint32_t steps = 0; layer != TB_START; steps++) {
        if (steps > guard || (ops_end && n_ops + 1 > a.ops_stride)) {
            status = BG_ERR_TRACEBACK;
            break;
        }
        if (layer >= TB_INS && layer <= TB_MATCH && i >= 1 && i <= m && j >= 1 && j < n) {
            const bool is_ins = layer == TB_INS, is_del = layer == TB_DEL;
            const uint32_t ti = i - (is_del ? 0u : 1u), tj = j - (is_ins ? 0u : 1u);
            if (ti >= 1 && tj >= 1) {
                if (!have_c && (is_ins || is_del)) c_here = cell(i, j);
                const bool ext = (is_ins && (c_here & 8u)) || (is_del && (c_here & 16u));
                push(is_ins ? (uint32_t)BG_OP_INS : is_del ? (uint32_t)BG_OP_DEL : layer == TB_MATCH ? (uint32_t)BG_OP_MATCH : (uint32_t)BG_OP_SUBST);
                c_here = cell(ti, tj);
                have_c = true;
                if (!ext) layer = s_nibble_of_code(c_here & 7u);
                i = ti;
                j = tj;
                continue;
            }
        }
        have_c = false;
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
                next = (i && j) ? s_nib(i - 1, j - 1) : TB_START;
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
                // K1p (tb_fmt 1) publishes Lx[j] <= m packed: bytes with 16 lanes per pair (m <= 192), else 16 bits
                if ((geo.tb_fmt == 2 || geo.tb_fmt == 3) && j != n) {  // the LF fills publish no Lx[j < n]: no local path asks for it
                    status = BG_ERR_TRACEBACK;
                    layer = TB_START;
                    continue;
                }
                const uint32_t lx = (j == n) ? lxn
                                    : !geo.tb_fmt ? (uint32_t)gLx[j]
                                    : geo.lp == 16 ? (uint32_t)((const uint8_t*)gLx)[j] : (uint32_t)((const uint16_t*)gLx)[j];
                push_clip(BG_OP_XCLIP, lx);
                i -= lx;
                xend = i;
                next = i <= m ? s_nib(i, j) : TB_START;
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
                next = j <= n ? s_nib(i, j) : TB_START;
                break;
            }
        }
        if (i > m || j > n) {  // underflow: the reference would panic on the index
            status = BG_ERR_TRACEBACK;
            break;
        }
        layer = next;
    }

    if (dword_ok) {  // the last partial group
        const uint32_t k = n_ops & 3, done = n_ops - k;
        for (uint32_t t = 0; t < k; t++) ops_end[-(int64_t)(done + t + 1)] = (uint8_t)(acc >> (8 * (k - 1 - t)));
    }
    bg_alignment_t rec;
    rec.score = score;  // S[n % 2][m], mod.rs:912
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
    rec._reserved = 0;
    a.out[a.pair0 + pair] = rec;
}

void launch_traceback(const SwArgs& a, int nw, hipStream_t st) {
    const uint32_t blocks = (a.n_pairs + 255) / 256;
    if (nw == 2)
        sw_traceback_kernel<2, false><<<dim3(blocks), dim3(256), 0, st>>>(a);
    else
        sw_traceback_kernel<1, false><<<dim3(blocks), dim3(256), 0, st>>>(a);
    if (a.perm) {
        if (nw == 2)
            sw_traceback_kernel<2, true><<<dim3(blocks), dim3(256), 0, st>>>(a);
        else
            sw_traceback_kernel<1, true><<<dim3(blocks), dim3(256), 0, st>>>(a);
    }
}

}  // namespace bgsw
