// K3v2, second half: banded_epilogue_kernel + the launcher.  The fill kernel itself (banded_fill2.inc, where the design
// notes are) is instantiated in three units of its own — banded_fill2_narrow.hip, banded_fill2_narrow_xp.hip,
// banded_fill2_wide.hip — so that they compile in parallel (one unit with all three took five minutes).
#include "banded_fill2.inc"

namespace bgband_dev {
void launch_fill2_narrow(const BandArgs& a, dim3 grid, hipStream_t st);     // <R, LP, NARROW, no x-prefix clip>
void launch_fill2_narrow_xp(const BandArgs& a, dim3 grid, hipStream_t st);  // <R, LP, NARROW, x-prefix clip>
void launch_fill2_wide(const BandArgs& a, dim3 grid, hipStream_t st);       // <R, LP, plain int32>

namespace {

// Last-column epilogue (banded.rs:683-723) + Sn[m] / Ly[m] (665-670) for one pair per wavefront, from the
// per-row values the fill left in aux.  The arithmetic is K3's (banded_fill.hip), fed from memory.
template <int R, bool NARROW>
__global__ __launch_bounds__(256) void banded_epilogue_kernel(const BandArgs a) {
    __builtin_amdgcn_s_setprio(3);  // (short, latency bound, next to other kernels)
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

bool launch_band_fill2(const BandArgs& a0, bool narrow, hipStream_t st, hipEvent_t after_fill, hipStream_t epi, hipStream_t pre,
                       hipEvent_t pre_done) {
    // epi: the epilogue runs there, behind after_fill (so that `st` can go on with the fill of the next sub-batch); null: on st
    // pre: the stream that prepared this sub-batch (banded_api.hip); phase 1 of a split fill runs there too — under the tail
    //      of the previous sub-batch's long kernel instead of behind it — and `st` takes over behind pre_done; null: all on st
    auto join_pre = [&]() {
        if (pre) {
            (void)hipEventRecord(pre_done, pre);
            (void)hipStreamWaitEvent(st, pre_done, 0);
        }
    };
    constexpr int LP = BF2_LP, PW = 64 / LP;
    BandArgs a = a0;
    const uint32_t jobs = (a.n_pairs + PW - 1) / PW;
    const dim3 grid((jobs + 3) / 4);
    if (narrow) {
        // scorings that admit interior runs (band_split, banded_kernels.h): K3v2 on the strips before them, K3i on the runs,
        // K3v2 on the strips behind them — three launches, the first and the last a few strips per pair
        const bool split = a.split && a.sc.xp <= NEG / 2;
        uint32_t* const started = a.started;
        if (!split) {
            join_pre();
            a.phase = 0;
            a.split = 0;
            if (a.sc.xp > NEG / 2) launch_fill2_narrow_xp(a, grid, st);
            else launch_fill2_narrow(a, grid, st);
        } else {
            a.started = nullptr;  // whoever waits for "the fill is resident" means the long launch
            a.phase = 1;
            launch_fill2_narrow(a, grid, pre ? pre : st);
            join_pre();
            a.started = started;
            if (a.packed) {
                // K3p first (two pairs per lane group, 16-bit keys relative to a per-strip base); the pairs it flags — a
                // band cell below the floor of its strip — go through the int32 kernels again: phase 1 once more (bnd is
                // reused by every strip, so the run's first boundary row has to be rebuilt), then K3i.  Both launches find
                // no flag and leave within microseconds in the usual case.
                launch_fill2p(a, st);
                a.started = nullptr;
                a.redo = 1;
                a.phase = 1;
                launch_fill2_narrow(a, grid, st);
                launch_fill2i(a, grid, st);
                a.redo = 0;
            } else {
                launch_fill2i(a, grid, st);
            }
            a.started = nullptr;
            a.phase = 2;
            launch_fill2_narrow(a, grid, st);
        }
        if (after_fill) (void)hipEventRecord(after_fill, st);
        if (epi && after_fill) (void)hipStreamWaitEvent(epi, after_fill, 0);
        banded_epilogue_kernel<2, true><<<dim3((a.n_pairs + 3) / 4), dim3(256), 0, epi && after_fill ? epi : st>>>(a);
    } else {
        join_pre();
        a.phase = 0;
        a.split = 0;
        launch_fill2_wide(a, grid, st);
        if (after_fill) (void)hipEventRecord(after_fill, st);
        if (epi && after_fill) (void)hipStreamWaitEvent(epi, after_fill, 0);
        banded_epilogue_kernel<2, false><<<dim3((a.n_pairs + 3) / 4), dim3(256), 0, epi && after_fill ? epi : st>>>(a);
    }
    return true;
}

}  // namespace bgband_dev
