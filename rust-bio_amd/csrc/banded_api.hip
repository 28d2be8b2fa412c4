// C-ABI entry points of the banded aligner: bg_align_banded_batch, bg_band_create_batch.
// Host side of `banded::Aligner::{custom,global,semiglobal,local}` (pairwise/banded.rs:282,872,901,972):
// the wrappers' clip overrides, Band::create per pair on host threads (band_host.cpp), the
// MAX_CELLS early-out (banded.rs:104,407-420), then K3 + K4 over sub-batches.
#include <algorithm>
#include <atomic>
#include <thread>

#include "band_host.h"
#include "banded_kernels.h"

using namespace bgband_dev;

int bg_compact_matrix(const int32_t* matrix, std::vector<uint8_t>& code_map, std::vector<int32_t>& table);

namespace {

constexpr uint64_t kMaxCells = 5000000;  // banded.rs:104

struct HostPair {
    uint32_t m = 0, n = 0, flags = BP_OK;
    uint32_t start_0 = 0, end_0 = 0, start_n = 0, end_n = 0;
    uint64_t cells = 0;
    std::vector<int2> rowc;         // per row {cf, cl}
    std::vector<uint32_t> row_off;  // per row traceback byte offset
    uint64_t tb_bytes = 0;
};

bgband::ClipScores clip_scores(const bg_scoring_t* sc, int mode) {
    bgband::ClipScores c;
    c.gap_open = sc->gap_open;
    c.gap_extend = sc->gap_extend;
    c.xclip_prefix = sc->xclip_prefix;
    c.xclip_suffix = sc->xclip_suffix;
    c.yclip_prefix = sc->yclip_prefix;
    c.yclip_suffix = sc->yclip_suffix;
    // banded.rs:872-1004: the wrappers overwrite the clips before Band::create runs
    if (mode == BG_MODE_GLOBAL) c.xclip_prefix = c.xclip_suffix = c.yclip_prefix = c.yclip_suffix = BG_MIN_SCORE;
    if (mode == BG_MODE_SEMIGLOBAL) {
        c.xclip_prefix = c.xclip_suffix = BG_MIN_SCORE;
        c.yclip_prefix = c.yclip_suffix = 0;
    }
    if (mode == BG_MODE_LOCAL) c.xclip_prefix = c.xclip_suffix = c.yclip_prefix = c.yclip_suffix = 0;
    c.match_score = sc->match_score;
    c.match_scores_some = sc->match_scores_some != 0;
    return c;
}

// per-column row ranges -> per-row column ranges (one interval per row for a monotone band)
void rows_from_columns(const bgband::Band& b, HostPair& hp) {
    const uint32_t m = hp.m, n = hp.n;
    hp.rowc.assign((size_t)m + 1, make_int2(1, 0));
    for (uint32_t j = 0; j <= n; j++) {
        if (b.end[j] <= b.start[j]) continue;
        for (uint32_t i = b.start[j]; i < b.end[j] && i <= m; i++) {
            int2& rc = hp.rowc[i];
            if (rc.y < rc.x) {
                rc.x = (int)j;
                rc.y = (int)j;
            } else {
                rc.y = (int)j;
            }
        }
    }
    hp.row_off.assign((size_t)m + 1, 0);
    uint64_t off = 0, covered = 0;
    for (uint32_t i = 0; i <= m; i++) {
        hp.row_off[i] = (uint32_t)off;
        const int2 rc = hp.rowc[i];
        if (rc.y >= rc.x) {
            covered += (uint64_t)(rc.y - rc.x + 1);
            // row 0 is a closed form, not stored; rows start dword-aligned (K3 stores four cells at a time)
            if (i >= 1) off += ((uint64_t)(rc.y - rc.x + 1) + 3) & ~3ull;
        }
    }
    hp.tb_bytes = (off + 15) & ~15ull;
    // every row's band columns must form ONE interval (no holes) for the device layout
    if (covered != hp.cells || off > 0xFFFFFFF0ull) hp.flags = BP_UNSUPPORTED;
}

void build_pair(const bgband::ClipScores& cs, uint32_t k, uint32_t w, const uint8_t* x, uint32_t m, const uint8_t* y,
                uint32_t n, bgband::Band& band, bgband::Workspace& ws, HostPair& hp, bool want_rows) {
    hp.m = m;
    hp.n = n;
    hp.flags = BP_OK;
    band.create(x, m, y, n, k, w, cs, ws);
    hp.cells = band.num_cells();
    hp.start_0 = band.start[0];
    hp.end_0 = band.end[0];
    hp.start_n = band.start[n];
    hp.end_n = band.end[n];
    if (hp.cells > kMaxCells) {
        hp.flags = BP_TOO_MANY_CELLS;
        return;
    }
    if (n == 0 || !band.monotone()) {
        // with an empty y the reference's own traceback does not terminate in most modes;
        // non-monotone bands never come out of Band::create
        hp.flags = BP_UNSUPPORTED;
        return;
    }
    if (want_rows) rows_from_columns(band, hp);
}

template <typename F>
void parallel_for(uint64_t n, F&& fn) {
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, n / 4));
    if (nt <= 1) {
        fn(0, 0, n);
        return;
    }
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            for (;;) {
                const uint64_t lo = next.fetch_add(16);
                if (lo >= n) break;
                fn(t, lo, std::min<uint64_t>(n, lo + 16));
            }
        });
    for (auto& t : th) t.join();
}

int check_scoring(const bg_scoring_t* sc) {
    if (sc->gap_open > 0 || sc->gap_extend > 0 || sc->xclip_prefix > 0 || sc->xclip_suffix > 0 ||
        sc->yclip_prefix > 0 || sc->yclip_suffix > 0)
        return BG_ERR_POSITIVE_PENALTY;
    return BG_OK;
}

}  // namespace

// Band::create for a batch (banded.rs:1278): writes n+1 half-open row ranges per pair at
// band_off[p] (same offsets for start and end); band_cells = Band::num_cells.
extern "C" int bg_band_create_batch(const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w, uint64_t n_pairs,
                                    const uint8_t* x, const uint64_t* x_off, const uint8_t* y, const uint64_t* y_off,
                                    const uint64_t* band_off, uint32_t* start, uint32_t* end, uint64_t* band_cells) {
    if (!sc || !x_off || !y_off || (n_pairs && (!band_off || !start || !end))) return BG_ERR_INVALID_ARG;
    int rc = check_scoring(sc);
    if (rc) return rc;
    const bgband::ClipScores cs = clip_scores(sc, mode);
    parallel_for(n_pairs, [&](unsigned, uint64_t lo, uint64_t hi) {
        bgband::Band band;
        bgband::Workspace ws;
        for (uint64_t p = lo; p < hi; p++) {
            const uint32_t m = (uint32_t)(x_off[p + 1] - x_off[p]), n = (uint32_t)(y_off[p + 1] - y_off[p]);
            band.create(x + x_off[p], m, y + y_off[p], n, k, w, cs, ws);
            memcpy(start + band_off[p], band.start.data(), (size_t)(n + 1) * 4);
            memcpy(end + band_off[p], band.end.data(), (size_t)(n + 1) * 4);
            if (band_cells) band_cells[p] = band.num_cells();
        }
    });
    return BG_OK;
}

extern "C" int bg_align_banded_batch(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w,
                                     uint64_t n_pairs, const uint8_t* x, const uint64_t* x_off, const uint8_t* y,
                                     const uint64_t* y_off, bg_alignment_t* out, uint8_t* ops_buf, uint64_t ops_cap,
                                     uint64_t* ops_used, uint64_t* band_cells) {
    if (!ctx || !sc || mode < BG_MODE_CUSTOM || mode > BG_MODE_LOCAL) return BG_ERR_INVALID_ARG;
    int rc = check_scoring(sc);
    if (rc) return rc;
    if (ops_used) *ops_used = 0;
    if (n_pairs == 0) return BG_OK;
    if (!x_off || !y_off || !out) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    const bgband::ClipScores cs = clip_scores(sc, mode);
    uint64_t max_x = 0, max_y = 0;
    for (uint64_t p = 0; p < n_pairs; p++) {
        max_x = std::max(max_x, x_off[p + 1] - x_off[p]);
        max_y = std::max(max_y, y_off[p + 1] - y_off[p]);
    }
    if (max_x > (1u << 24) || max_y > (1u << 24)) return BG_ERR_TOO_LARGE;
    const uint64_t stride = (max_x + max_y + 4 + 3) & ~3ull;

    BandArgs a = {};
    a.sc = {cs.gap_open, cs.gap_extend, cs.xclip_prefix, cs.xclip_suffix, cs.yclip_prefix, cs.yclip_suffix,
            sc->match_score, sc->mismatch_score};
    a.mode = mode;
    a.filter_clips = (mode == BG_MODE_SEMIGLOBAL || mode == BG_MODE_LOCAL);
    a.ops_stride = stride;
    int sm = SCORE_PARAMS;
    // (the tabulated match function is compacted exactly like in sw_api.hip)
    std::vector<uint8_t> code_map;
    std::vector<int32_t> table;
    if (sc->matrix) {
        const int A = bg_compact_matrix(sc->matrix, code_map, table);
        sm = A <= kMaxLdsAlphabet ? SCORE_LDS : SCORE_GLOBAL;
        if ((rc = bg_reserve(&ctx->table, &ctx->table_bytes, 256 + table.size() * 4))) return rc;
        BG_HIP(hipMemcpy((uint8_t*)ctx->table + 256, table.data(), table.size() * 4, hipMemcpyHostToDevice));
        BG_HIP(hipMemcpy(ctx->table, code_map.data(), 256, hipMemcpyHostToDevice));
        a.code_map = (const uint8_t*)ctx->table;
        a.table = (const int32_t*)((uint8_t*)ctx->table + 256);
        a.alpha = A;
    }
    band_fill_fn fill = get_band_fill(sm);

    // sequences and outputs of the whole batch live on the device; band data goes chunk by chunk
    const uint64_t xb = x_off[n_pairs], yb = y_off[n_pairs];
    uint8_t *d_x = nullptr, *d_y = nullptr, *d_ops = nullptr;
    uint64_t *d_xo = nullptr, *d_yo = nullptr;
    bg_alignment_t* d_out = nullptr;
    void *d_pairs = nullptr, *d_rowc = nullptr, *d_roff = nullptr, *d_tb = nullptr, *d_aux = nullptr;
    size_t cap_pairs = 0, cap_rowc = 0, cap_roff = 0, cap_tb = 0, cap_aux = 0;
    std::vector<uint8_t> h_ops;
    auto cleanup = [&]() {
        hipFree(d_x); hipFree(d_y); hipFree(d_ops); hipFree(d_xo); hipFree(d_yo); hipFree(d_out);
        hipFree(d_pairs); hipFree(d_rowc); hipFree(d_roff); hipFree(d_tb); hipFree(d_aux);
    };
    auto run = [&]() -> int {
        BG_HIP(hipMalloc((void**)&d_x, std::max<uint64_t>(xb, 16)));
        BG_HIP(hipMalloc((void**)&d_y, std::max<uint64_t>(yb, 16)));
        BG_HIP(hipMalloc((void**)&d_xo, (n_pairs + 1) * 8));
        BG_HIP(hipMalloc((void**)&d_yo, (n_pairs + 1) * 8));
        BG_HIP(hipMalloc((void**)&d_out, n_pairs * sizeof(bg_alignment_t)));
        if (ops_buf) BG_HIP(hipMalloc((void**)&d_ops, n_pairs * stride));
        if (xb) BG_HIP(hipMemcpyAsync(d_x, x, xb, hipMemcpyHostToDevice, st));
        if (yb) BG_HIP(hipMemcpyAsync(d_y, y, yb, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_xo, x_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync(d_yo, y_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice, st));
        a.x = d_x;
        a.x_off = d_xo;
        a.y = d_y;
        a.y_off = d_yo;
        a.out = d_out;
        a.ops = d_ops;

        const uint64_t chunk_pairs = ctx->chunk_pairs > 0 ? (uint64_t)ctx->chunk_pairs : 4096;
        const uint64_t budget = 32ull << 30;
        std::vector<HostPair> hp;
        std::vector<BandPair> dp;
        std::vector<int2> rowc_all;
        std::vector<uint32_t> roff_all;
        for (uint64_t p0 = 0; p0 < n_pairs;) {
            const uint64_t want = std::min<uint64_t>(chunk_pairs, n_pairs - p0);
            hp.assign(want, HostPair());
            parallel_for(want, [&](unsigned, uint64_t lo, uint64_t hi) {
                bgband::Band band;
                bgband::Workspace ws;
                for (uint64_t q = lo; q < hi; q++) {
                    const uint64_t p = p0 + q;
                    build_pair(cs, k, w, x + x_off[p], (uint32_t)(x_off[p + 1] - x_off[p]), y + y_off[p],
                               (uint32_t)(y_off[p + 1] - y_off[p]), band, ws, hp[q], true);
                }
            });
            // take as many pairs as fit the scratch budget
            uint64_t take = 0, rows = 0, tbb = 0, auxw = 0;
            for (; take < want; take++) {
                const HostPair& h = hp[take];
                const uint64_t r2 = rows + h.m + 1, t2 = tbb + (h.flags == BP_OK ? h.tb_bytes : 0);
                const uint64_t a2 = auxw + ((BandAux(h.m, h.n).words() + 3) & ~3ull);
                if (take > 0 && t2 + a2 * 4 + r2 * 12 > budget) break;
                rows = r2;
                tbb = t2;
                auxw = a2;
            }
            dp.assign(take, BandPair());
            rowc_all.resize(rows);
            roff_all.resize(rows);
            uint64_t ro = 0, to = 0, ao = 0;
            for (uint64_t q = 0; q < take; q++) {
                const HostPair& h = hp[q];
                BandPair& d = dp[q];
                d.rowc_off = ro;
                d.tb_off = to;
                d.aux_off = ao;
                d.start_0 = h.start_0;
                d.end_0 = h.end_0;
                d.start_n = h.start_n;
                d.end_n = h.end_n;
                d.flags = h.flags;
                if (h.flags == BP_OK) {
                    memcpy(&rowc_all[ro], h.rowc.data(), (size_t)(h.m + 1) * sizeof(int2));
                    memcpy(&roff_all[ro], h.row_off.data(), (size_t)(h.m + 1) * 4);
                    to += h.tb_bytes;
                } else {
                    for (uint32_t i = 0; i <= h.m; i++) {
                        rowc_all[ro + i] = make_int2(1, 0);
                        roff_all[ro + i] = 0;
                    }
                }
                ro += h.m + 1;
                ao += (BandAux(h.m, h.n).words() + 3) & ~3ull;
                if (band_cells) band_cells[p0 + q] = h.cells;
            }
            int r2;
            if ((r2 = bg_reserve(&d_pairs, &cap_pairs, take * sizeof(BandPair)))) return r2;
            if ((r2 = bg_reserve(&d_rowc, &cap_rowc, std::max<size_t>(rows * sizeof(int2), 64)))) return r2;
            if ((r2 = bg_reserve(&d_roff, &cap_roff, std::max<size_t>(rows * 4, 64)))) return r2;
            if ((r2 = bg_reserve(&d_tb, &cap_tb, std::max<size_t>(to, 64)))) return r2;
            if ((r2 = bg_reserve(&d_aux, &cap_aux, std::max<size_t>(ao * 4, 64)))) return r2;
            BG_HIP(hipMemcpyAsync(d_pairs, dp.data(), take * sizeof(BandPair), hipMemcpyHostToDevice, st));
            BG_HIP(hipMemcpyAsync(d_rowc, rowc_all.data(), rows * sizeof(int2), hipMemcpyHostToDevice, st));
            BG_HIP(hipMemcpyAsync(d_roff, roff_all.data(), rows * 4, hipMemcpyHostToDevice, st));
            BG_HIP(hipMemsetAsync(d_aux, 0, ao * 4, st));
            a.pairs = (const BandPair*)d_pairs;
            a.rowc = (const int2*)d_rowc;
            a.row_off = (const uint32_t*)d_roff;
            a.tb = (uint8_t*)d_tb;
            a.aux = (int32_t*)d_aux;
            a.pair0 = p0;
            a.n_pairs = (uint32_t)take;
            if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st));
            fill<<<dim3((unsigned)((take + 3) / 4)), dim3(256), 0, st>>>(a);
            BG_HIP(hipGetLastError());
            if (ctx->timing) {
                BG_HIP(hipEventRecord(ctx->ev[1], st));
                BG_HIP(hipEventSynchronize(ctx->ev[1]));
                float ms = 0;
                BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
                ctx->last.fill_ms += ms;
                ctx->last.fill_launches += 1;
                BG_HIP(hipEventRecord(ctx->ev[0], st));
            }
            launch_band_traceback(a, st);
            BG_HIP(hipGetLastError());
            if (ctx->timing) {
                BG_HIP(hipEventRecord(ctx->ev[1], st));
                BG_HIP(hipEventSynchronize(ctx->ev[1]));
                float ms = 0;
                BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
                ctx->last.traceback_ms += ms;
                ctx->last.traceback_launches += 1;
            }
            BG_HIP(hipStreamSynchronize(st));  // the host vectors of this chunk are reused
            p0 += take;
        }
        BG_HIP(hipMemcpyAsync(out, d_out, n_pairs * sizeof(bg_alignment_t), hipMemcpyDeviceToHost, st));
        if (ops_buf) {
            h_ops.resize(n_pairs * stride);
            BG_HIP(hipMemcpyAsync(h_ops.data(), d_ops, n_pairs * stride, hipMemcpyDeviceToHost, st));
        }
        BG_HIP(hipStreamSynchronize(st));
        return BG_OK;
    };
    rc = run();
    cleanup();
    if (rc) return rc;
    uint64_t used = 0;
    int status = BG_OK;
    for (uint64_t p = 0; p < n_pairs; p++) {
        if (out[p].status && status == BG_OK) status = out[p].status;
        const uint64_t src = out[p].ops_off;
        out[p].ops_off = used;
        if (ops_buf && out[p].status == BG_OK) {
            if (used + out[p].n_ops <= ops_cap)
                memcpy(ops_buf + used, h_ops.data() + src, out[p].n_ops);
            else if (status == BG_OK)
                status = BG_ERR_OPS_CAP;
        }
        used += out[p].n_ops;
    }
    if (ops_used) *ops_used = used;
    return status;
}
