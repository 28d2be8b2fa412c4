// C-ABI entry points of the banded aligner: bg_align_banded_batch, bg_band_create_batch.
// Host side of `banded::Aligner::{custom,global,semiglobal,local}` (pairwise/banded.rs:282,872,901,972):
// the wrappers' clip overrides, Band::create per pair on host threads (band_host.cpp), the
// MAX_CELLS early-out (banded.rs:104,407-420), then K3 + K4 over sub-batches.
#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>

#include "band_host.h"
#include <sched.h>

#include <chrono>

#include "band_device.h"
#include "banded_kernels.h"

using namespace bgband_dev;

int bg_compact_matrix(const int32_t* matrix, std::vector<uint8_t>& code_map, std::vector<int32_t>& table);

namespace bgband { extern std::atomic<uint64_t> g_prof[4]; }
namespace {

constexpr uint64_t kMaxCells = 5000000;  // banded.rs:104

struct HostPair {
    uint32_t m = 0, n = 0, flags = BP_OK;
    uint32_t start_0 = 0, end_0 = 0, start_n = 0, end_n = 0;
    uint64_t cells = 0;
    uint64_t tb_bytes = 0;
};

unsigned host_threads() { return bg_host_threads(); }

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

// Per-column row ranges -> per-row column ranges, written straight into the pinned staging arrays.
// For a monotone band (Band::monotone) the columns containing row i form one interval: its first
// column is the first one whose range reaches down to i, its last the last one starting at or above
// i — two O(m + n) sweeps instead of visiting every band cell.
void rows_from_columns(const bgband::Band& b, HostPair& hp, int2* rowc, uint32_t* row_off) {
    const uint32_t m = hp.m, n = hp.n;
    for (uint32_t i = 0; i <= m; i++) rowc[i] = make_int2(1, 0);
    uint32_t done = 0;  // rows < done were inside an earlier column
    for (uint32_t j = 0; j <= n; j++) {
        if (b.end[j] <= b.start[j]) continue;
        const uint32_t e = std::min<uint32_t>(b.end[j], m + 1);
        for (uint32_t i = std::max(b.start[j], done); i < e; i++) rowc[i].x = (int)j;
        done = std::max(done, e);
    }
    uint32_t lim = m + 1;  // rows >= lim are inside a later column
    for (uint32_t j = n + 1; j-- > 0;) {
        if (b.end[j] <= b.start[j]) continue;
        const uint32_t e = std::min<uint32_t>(std::min<uint32_t>(b.end[j], m + 1), lim);
        for (uint32_t i = b.start[j]; i < e; i++) rowc[i].y = (int)j;
        lim = std::min(lim, b.start[j]);
    }
    uint64_t off = 0, covered = 0;
    row_off[0] = 0;  // row 0 is a closed form, not stored
    if (rowc[0].y >= rowc[0].x) covered += (uint64_t)(rowc[0].y - rowc[0].x + 1);
    // rows 1..m in line groups of eight: 16-cell groups, the groups of the eight rows interleaved (banded_kernels.h)
    for (uint32_t q0 = 0; q0 < m; q0 += kTbLineRows) {
        uint64_t groups = 0;
        for (uint32_t q = q0; q < std::min(m, q0 + kTbLineRows); q++) {
            const int2 rc = rowc[q + 1];
            if (rc.y >= rc.x) {
                covered += (uint64_t)(rc.y - rc.x + 1);
                groups = std::max<uint64_t>(groups, ((uint64_t)(rc.y - rc.x + 1) + (kTbRowAlign - 1)) / kTbRowAlign);
            }
        }
        for (uint32_t q = q0; q < std::min(m, q0 + kTbLineRows); q++) row_off[q + 1] = (uint32_t)(off + (q - q0) * kTbRowAlign);
        off += groups * kTbGroupStride;
    }
    hp.tb_bytes = (off + 15) & ~15ull;
    // every row's band columns must form ONE interval (no holes) for the device layout
    if (covered != hp.cells || off > 0xFFFFFFF0ull) hp.flags = BP_UNSUPPORTED;
}

// everything the device needs to know about one pair's band (the band itself is already in `band`)
void build_pair(uint32_t m, uint32_t n, const bgband::Band& band, HostPair& hp, int2* rowc, uint32_t* row_off) {
    hp.m = m;
    hp.n = n;
    hp.flags = BP_OK;
    hp.tb_bytes = 0;
    hp.cells = band.num_cells();
    hp.start_0 = band.start[0];
    hp.end_0 = band.end[0];
    hp.start_n = band.start[n];
    hp.end_n = band.end[n];
    if (hp.cells > kMaxCells) {
        hp.flags = BP_TOO_MANY_CELLS;
    } else if (n == 0 || !band.monotone()) {
        // with an empty y the reference's own traceback does not terminate in most modes;
        // non-monotone bands never come out of Band::create
        hp.flags = BP_UNSUPPORTED;
    } else {
        rows_from_columns(band, hp, rowc, row_off);
    }
    if (hp.flags != BP_OK) {
        for (uint32_t i = 0; i <= m; i++) {
            rowc[i] = make_int2(1, 0);
            row_off[i] = 0;
        }
        hp.tb_bytes = 0;
    }
}

// grow-only pinned host buffer
int pinned_reserve(void** p, size_t* cur, size_t need) {
    if (need <= *cur) return BG_OK;
    if (*p) {
        hipHostFree(*p);
        *p = nullptr;
        *cur = 0;
    }
    need = (need + 4095) & ~(size_t)4095;
    BG_HIP(hipHostMalloc(p, need, hipHostMallocDefault));
    *cur = need;
    return BG_OK;
}

}  // namespace

// Scratch that survives between calls (bg_ctx::band): two sets, so that the traceback of one
// sub-batch (K4, latency bound, a handful of wavefronts) overlaps the fill of the next one (K3) and
// the host threads that build the following band.
struct bg_band_scratch {
    struct Set {
        // pinned staging
        void *h_pairs = nullptr, *h_rowc = nullptr, *h_roff = nullptr;
        size_t hc_pairs = 0, hc_rowc = 0, hc_roff = 0;
        // device
        void *d_pairs = nullptr, *d_rowc = nullptr, *d_roff = nullptr, *d_tb = nullptr, *d_aux = nullptr;
        size_t dc_pairs = 0, dc_rowc = 0, dc_roff = 0, dc_tb = 0, dc_aux = 0;
        hipEvent_t copied = nullptr, filled = nullptr, traced = nullptr, built = nullptr, matched = nullptr;
        bool built_valid = false;  // `built` has been recorded in this call
        hipEvent_t fill_gone = nullptr;  // the fill kernel itself is off the device (its epilogue may still run)
        hipEvent_t pre_done = nullptr;   // ... and what pre_stream did for it is done
        hipEvent_t cleared = nullptr;    // the aux block has been zeroed (on aux_stream)
        bool busy = false, fill_gone_valid = false;
    } set[2];
    // device band builder (band_device.hip): scratch slices per pair + its per-pair state
    void* db[2][17] = {};  // the builder's own arrays, one set per sub-batch parity (the join of c + 2 runs next to the chaining of c + 1)
    size_t db_cap[2][17] = {};
    hipStream_t build_stream = nullptr;
    hipStream_t join_stream = nullptr;  // k-mer join + chain preparation of the sub-batch after next
    hipStream_t pre_stream = nullptr;   // what a fill needs before its long kernel: the pair table, the waits, K3v2's first strips
    hipEvent_t seq_ready = nullptr;
    hipEvent_t chained = nullptr;  // the chaining's event loop of the sub-batch being built has left the device
    bool chained_valid = false;    // ... recorded in this call, under a running fill
    void* h_state = nullptr;  // pinned copy of the builder's BandDevPair array
    size_t h_state_cap = 0;
    void* io[6] = {};  // x, y, x_off, y_off, out, ops on the device
    size_t io_cap[6] = {};
    void* h_ops = nullptr;  // pinned landing zone of the operations
    size_t h_ops_cap = 0;
    void *d_cmp = nullptr, *d_cscan = nullptr;  // host-buffer flavour: the operations compacted on the device, scan scratch
    size_t d_cmp_cap = 0, d_cscan_cap = 0;
    uint64_t* d_cell = nullptr;                 // ... and their running byte count over the sub-batches of a call
    uint64_t* d_dlslot = nullptr;               // per sub-batch {bytes, end offset} of its compacted operations (ring of kDlSlots)
    hipStream_t dl_stream = nullptr;            // ... which a few blocks on this high-priority stream bring to h_ops meanwhile
    static constexpr uint64_t kDlSlots = 1024;
    hipStream_t tb_stream = nullptr;
    hipStream_t aux_stream = nullptr;   // clears the aux block of the next sub-batch under the running fill
    hipStream_t copy_stream = nullptr;  // host-buffer flavour: sequence slices go up here
    uint32_t* d_started = nullptr;  // blocks of the K3v2 launches of the current call that have started (see banded_fill2.hip)
    uint32_t started_target = 0;    // ... and how many have been launched
};

extern "C" int bg_band_redo_pairs(bg_ctx* ctx, uint64_t* out) {
    if (!ctx || !out) return BG_ERR_INVALID_ARG;
    *out = 0;
    if (!ctx->band || !ctx->band->d_started) return BG_OK;
    BG_HIP(hipSetDevice(ctx->device));  // (the caller's current device may be another one: several contexts, torch elsewhere)
    BG_HIP(hipDeviceSynchronize());
    uint32_t v = 0;
    BG_HIP(hipMemcpy(&v, ctx->band->d_started + 1, 4, hipMemcpyDeviceToHost));
    *out = v;
    return BG_OK;
}

void bg_band_scratch_free(bg_band_scratch* b) {
    if (!b) return;
    for (auto& s : b->set) {
        hipHostFree(s.h_pairs); hipHostFree(s.h_rowc); hipHostFree(s.h_roff);
        hipFree(s.d_pairs); hipFree(s.d_rowc); hipFree(s.d_roff); hipFree(s.d_tb); hipFree(s.d_aux);
        hipFree(b->d_started);
        b->d_started = nullptr;
        if (s.copied) hipEventDestroy(s.copied);
        if (s.filled) hipEventDestroy(s.filled);
        if (s.traced) hipEventDestroy(s.traced);
        if (s.built) hipEventDestroy(s.built);
        if (s.matched) hipEventDestroy(s.matched);
        if (s.fill_gone) hipEventDestroy(s.fill_gone);
        if (s.pre_done) hipEventDestroy(s.pre_done);
        if (s.cleared) hipEventDestroy(s.cleared);
    }
    for (void* p : b->io) hipFree(p);
    hipFree(b->d_cmp);
    hipFree(b->d_cscan);
    hipFree(b->d_cell);
    hipFree(b->d_dlslot);
    if (b->dl_stream) hipStreamDestroy(b->dl_stream);
    for (auto& set : b->db)
        for (void* p : set) hipFree(p);
    if (b->join_stream) hipStreamDestroy(b->join_stream);
    if (b->pre_stream) hipStreamDestroy(b->pre_stream);
    hipHostFree(b->h_state);
    hipHostFree(b->h_ops);
    if (b->tb_stream) hipStreamDestroy(b->tb_stream);
    if (b->aux_stream) hipStreamDestroy(b->aux_stream);
    if (b->copy_stream) hipStreamDestroy(b->copy_stream);
    if (b->build_stream) hipStreamDestroy(b->build_stream);
    if (b->seq_ready) hipEventDestroy(b->seq_ready);
    if (b->chained) hipEventDestroy(b->chained);
    delete b;
}

namespace {

template <typename F>
void parallel_for(uint64_t n, uint64_t grain, F&& fn) {
    unsigned nt = host_threads();
    nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, n / grain));
    if (nt <= 1) {
        fn(0, 0, n);
        return;
    }
    std::atomic<uint64_t> next{0};
    bg_pool_run(nt, [&](unsigned t) {
        for (;;) {
            const uint64_t lo = next.fetch_add(grain);
            if (lo >= n) break;
            fn(t, lo, std::min<uint64_t>(n, lo + grain));
        }
    });
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
    parallel_for(n_pairs, 4, [&](unsigned, uint64_t lo, uint64_t hi) {
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
    if (getenv("BG_TRACE")) fprintf(stderr, "[bg banded] cpu-ms: kmers %.1f sdp %.1f band %.1f (threads %u)\n", bgband::g_prof[0] / 1e6, bgband::g_prof[1] / 1e6, bgband::g_prof[2] / 1e6, host_threads());
    return BG_OK;
}

// `make_band(p, band, ws)` fills the band of pair p (called from host threads); false = invalid input
using BandMaker = std::function<bool(uint64_t, bgband::Band&, bgband::Workspace&)>;

// device-resident flavour: sequences, offsets, records and (strided) operation slots stay in HBM
// Streams of the pipeline by queue priority.  The runtime maps a process's streams onto a few hardware queues PER PRIORITY
// LEVEL, and a hardware queue hands out its packets in order: a short kernel queued behind a long, starved one (the k-mer
// join under a fill) waits for that one's last block to be dispatched, whatever streams the two were launched on
// (profiles/r05_banded_timeline_hostsync.txt: the next fill's preparation started 1.3 ms after the join two sub-batches
// ahead ended, every cycle).  So the join goes to the low-priority queues, a fill's preparation to the high-priority
// ones, and neither can sit in front of the other or of the fill: the gap between two fills drops from 10.8 to 1.9 ms
// (profiles/r05_banded_timeline_prio.txt) — and the call gains nothing, because the join that used to run in that gap now
// starves under two fills in a row and the chaining behind it starts late (profiles/r05_banded_pipeline_experiments.txt).
// BG_BAND_STREAM_PRIO=0: all normal (rounds 2-4).
int band_stream_create(hipStream_t* s, int level) {
    static const int on = [] { const char* e = getenv("BG_BAND_STREAM_PRIO"); return e ? atoi(e) : 1; }();
    int lo = 0, hi = 0;
    if (!on || level == 0 || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo == hi)
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess ? BG_OK : BG_ERR_HIP;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, level > 0 ? hi : lo) == hipSuccess ? BG_OK : BG_ERR_HIP;
}

struct BandDevIO {
    const uint8_t* d_x;
    const uint64_t* d_xo;
    const uint8_t* d_y;
    const uint64_t* d_yo;
    bg_alignment_t* d_out;
    uint8_t* d_ops;       // may be null
    uint64_t ops_stride;  // >= max_x + max_y + 4
};

static int banded_batch_impl(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs, const uint8_t* x,
                             const uint64_t* x_off, const uint8_t* y, const uint64_t* y_off, bg_alignment_t* out,
                             uint8_t* ops_buf, uint64_t ops_cap, uint64_t* ops_used, uint64_t* band_cells,
                             const BandMaker& make_band, const uint32_t* dev_kw = nullptr, const BandDevIO* dio = nullptr) {
    if (!ctx || !sc || mode < BG_MODE_CUSTOM || mode > BG_MODE_LOCAL) return BG_ERR_INVALID_ARG;
    int rc = check_scoring(sc);
    if (rc) return rc;
    if (ops_used) *ops_used = 0;
    if (n_pairs == 0) return BG_OK;
    if (!x_off || !y_off || (!out && !dio)) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (!ctx->band) {
        ctx->band = new bg_band_scratch;
        BG_HIP(hipStreamCreateWithFlags(&ctx->band->tb_stream, hipStreamNonBlocking));
        for (auto& s : ctx->band->set) {
            BG_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
            BG_HIP(hipEventCreateWithFlags(&s.filled, hipEventDisableTiming));
            BG_HIP(hipEventCreateWithFlags(&s.traced, hipEventDisableTiming));
        }
    }
    bg_band_scratch& B = *ctx->band;
    hipStream_t st_tb = B.tb_stream;

    const bgband::ClipScores cs = clip_scores(sc, mode);
    uint64_t max_x = 0, max_y = 0;
    for (uint64_t p = 0; p < n_pairs; p++) {
        max_x = std::max(max_x, x_off[p + 1] - x_off[p]);
        max_y = std::max(max_y, y_off[p + 1] - y_off[p]);
    }
    if (max_x > (1u << 24) || max_y > (1u << 24)) return BG_ERR_TOO_LARGE;
    const uint64_t stride = dio ? dio->ops_stride : ((max_x + max_y + 4 + 3) & ~3ull);
    if (dio && dio->d_ops && stride < max_x + max_y + 4) return BG_ERR_OPS_CAP;

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
    // every reachable score within 24 bits: the scaled-key variant of K3v2 applies (same bound as sw_api.hip)
    const int64_t mag = std::max<int64_t>({std::abs((int64_t)cs.gap_open), std::abs((int64_t)cs.gap_extend),
                                           std::abs((int64_t)sc->match_score), std::abs((int64_t)sc->mismatch_score), 1});
    auto clip_ok = [](int32_t c) { return c <= BG_MIN_SCORE / 2 || c >= -(1 << 22); };  // 'minus infinity' or small
    const bool narrow = !ctx->force_wide && mag * ((int64_t)max_x + (int64_t)max_y + 8) < (1 << 24) && clip_ok(cs.xclip_prefix) &&
                        clip_ok(cs.xclip_suffix) && clip_ok(cs.yclip_prefix) && clip_ok(cs.yclip_suffix);

    const bool trace = getenv("BG_TRACE") != nullptr;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto t_last = now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto t1 = now();
        fprintf(stderr, "[bg banded] %-18s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_last).count());
        t_last = t1;
    };

    // sequences and result records of the whole batch live on the device; band data goes in sub-batches
    const uint64_t xb = x_off[n_pairs], yb = y_off[n_pairs];
    const uint8_t *d_x, *d_y;
    const uint64_t *d_xo, *d_yo;
    uint8_t* d_ops;
    bg_alignment_t* d_out;
    if (dio) {
        d_x = dio->d_x;
        d_y = dio->d_y;
        d_xo = dio->d_xo;
        d_yo = dio->d_yo;
        d_out = dio->d_out;
        d_ops = dio->d_ops;
    } else {
        const size_t io_need[6] = {std::max<uint64_t>(xb, 16), std::max<uint64_t>(yb, 16), (n_pairs + 1) * 8, (n_pairs + 1) * 8,
                                   n_pairs * sizeof(bg_alignment_t), ops_buf ? n_pairs * stride : 16};
        for (int i = 0; i < 6; i++)
            if ((rc = bg_reserve(&B.io[i], &B.io_cap[i], io_need[i]))) return rc;
        d_x = (uint8_t*)B.io[0];
        d_y = (uint8_t*)B.io[1];
        d_ops = ops_buf ? (uint8_t*)B.io[5] : nullptr;
        d_xo = (uint64_t*)B.io[2];
        d_yo = (uint64_t*)B.io[3];
        d_out = (bg_alignment_t*)B.io[4];
        BG_HIP(hipMemcpyAsync((void*)d_xo, x_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice, st));
        BG_HIP(hipMemcpyAsync((void*)d_yo, y_off, (n_pairs + 1) * 8, hipMemcpyHostToDevice, st));
        // the sequences go up in slices of one sub-batch each on a copy stream of their own: the first sub-batch starts
        // after 1 / n-th of the upload, the rest travels under its band construction and fill (upload_slices below)
    }
    a.x = d_x;
    a.x_off = d_xo;
    a.y = d_y;
    a.y_off = d_yo;
    a.out = d_out;
    a.ops = d_ops;
    lap("h2d sequences");
    // host-buffer flavour with operations: they are compacted on the device, sub-batch by sub-batch (an operation list
    // is at most m + n + 4 bytes)
    const bool compact_on_device = !dio && ops_buf != nullptr;
    if (compact_on_device) {
        if ((rc = bg_reserve(&B.d_cmp, &B.d_cmp_cap, xb + yb + 4 * n_pairs + 256))) return rc;
        if ((rc = bg_reserve(&B.d_cscan, &B.d_cscan_cap,
                             bg_compact_ops_scratch(std::min<uint64_t>(n_pairs, ctx->chunk_pairs > 0 ? (uint64_t)ctx->chunk_pairs : 16384)))))
            return rc;
        if (!B.d_cell) BG_HIP(hipMalloc((void**)&B.d_cell, 64));
        BG_HIP(hipMemsetAsync(B.d_cell, 0, 8, st));  // st_tb waits for st's events before every traceback
        // the operations of a sub-batch leave for the pinned buffer as soon as they are compacted, while the next ones are
        // computed (one download of everything after the last sub-batch was 20 ms of a 490 ms call, with the device idle)
        if ((rc = pinned_reserve(&B.h_ops, &B.h_ops_cap, xb + yb + 4 * n_pairs + 256))) return rc;
        if (!B.d_dlslot) BG_HIP(hipMalloc((void**)&B.d_dlslot, bg_band_scratch::kDlSlots * 16));
        if (!B.dl_stream) {
            int lo = 0, hi = 0;
            BG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            BG_HIP(hipStreamCreateWithPriority(&B.dl_stream, hipStreamNonBlocking, hi));
        }
    }

    // sub-batch size: enough wavefronts to fill the chip, small enough that a large batch pipelines
    const uint64_t chunk_pairs = ctx->chunk_pairs > 0 ? (uint64_t)ctx->chunk_pairs : 16384;
    // host-buffer flavour: sequence slices (pairs [k * chunk_pairs, (k + 1) * chunk_pairs)) and their upload events
    struct SliceEvents {
        std::vector<hipEvent_t> ev;
        ~SliceEvents() {
            for (hipEvent_t e : ev)
                if (e) hipEventDestroy(e);
        }
    } slices;
    const uint64_t n_slices = dio ? 0 : (n_pairs + chunk_pairs - 1) / chunk_pairs;
    uint64_t slices_up = 0, waited_fill = 0, waited_build = 0;
    (void)waited_build;
    if (n_slices && !B.copy_stream && (rc = band_stream_create(&B.copy_stream, -1))) return rc;
    auto upload_slices = [&](uint64_t upto) -> int {  // slices [slices_up, upto)
        for (; slices_up < std::min(upto, n_slices); slices_up++) {
            const uint64_t q0 = slices_up * chunk_pairs, q1 = std::min(n_pairs, q0 + chunk_pairs);
            if (slices_up == 0) {  // behind the offsets (and whatever the caller's stream ran before)
                hipEvent_t e0;
                BG_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
                slices.ev.push_back(e0);
                BG_HIP(hipEventRecord(e0, st));
                BG_HIP(hipStreamWaitEvent(B.copy_stream, e0, 0));
            }
            if (x_off[q1] > x_off[q0])
                BG_HIP(bg_copy_pieces((uint8_t*)d_x + x_off[q0], x + x_off[q0], x_off[q1] - x_off[q0], hipMemcpyHostToDevice, B.copy_stream));
            if (y_off[q1] > y_off[q0])
                BG_HIP(bg_copy_pieces((uint8_t*)d_y + y_off[q0], y + y_off[q0], y_off[q1] - y_off[q0], hipMemcpyHostToDevice, B.copy_stream));
            hipEvent_t e;
            BG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            slices.ev.push_back(e);  // ev[k + 1]: slice k is on the device
            BG_HIP(hipEventRecord(e, B.copy_stream));
        }
        return BG_OK;
    };
    // stream `s` may touch the sequences of pairs [0, upto) only behind their slices
    auto need_seq = [&](hipStream_t s, uint64_t& waited, uint64_t upto) -> int {
        if (!n_slices) return BG_OK;
        const uint64_t k1 = std::min(n_slices, (upto + chunk_pairs - 1) / chunk_pairs);
        int rcu = upload_slices(k1);
        if (rcu) return rcu;
        for (; waited < k1; waited++) BG_HIP(hipStreamWaitEvent(s, slices.ev[waited + 1], 0));
        return BG_OK;
    };
    if ((rc = need_seq(st, waited_fill, std::min<uint64_t>(n_pairs, chunk_pairs)))) return rc;  // the first slice
    // traceback + aux per scratch set (two sets): 40 GB each on an otherwise empty 288 GB part — but no more than a third of
    // what the device has free now plus what the sets already hold (a smaller or shared GPU, a torch caching allocator next
    // to the engine): a smaller budget cuts the sub-batches (`take < want` below) instead of failing bg_reserve with OOM
    uint64_t budget = (ctx->band_budget_gb > 0 ? (uint64_t)ctx->band_budget_gb : 40ull) << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            uint64_t held = 0;
            for (auto& s : B.set) held += s.dc_tb + s.dc_aux;
            budget = std::min<uint64_t>(budget, std::max<uint64_t>(((uint64_t)free_b + held) / 3, 1ull << 30));
        }
    }
    const uint64_t grain = std::max<uint64_t>(1, std::min<uint64_t>(64, 65536 / (max_x + max_y + 1)));
    // Two sub-batches are in flight: while K3/K4 of one run, the band of the next one is being built —
    // by band_device.hip on its own stream, or by the host threads.
    struct Plan {
        uint64_t p0 = 0, want = 0;
        bool on_device = false, matched = false;  // matched: the k-mer join of this sub-batch has been launched (issue_match)
        BandDevArgs d = {};
        std::vector<HostPair> hp;
        std::vector<uint64_t> row0;
    } plan[2];
    if (!B.build_stream) {
        BG_HIP(hipStreamCreateWithFlags(&B.build_stream, hipStreamNonBlocking));
        BG_HIP(hipEventCreateWithFlags(&B.seq_ready, hipEventDisableTiming));
        BG_HIP(hipEventCreateWithFlags(&B.chained, hipEventDisableTiming));
        for (auto& s : B.set) {
            BG_HIP(hipEventCreateWithFlags(&s.built, hipEventDisableTiming));
            BG_HIP(hipEventCreateWithFlags(&s.matched, hipEventDisableTiming));
            BG_HIP(hipEventCreateWithFlags(&s.fill_gone, hipEventDisableTiming));
        }
    }
    hipStream_t st_build = B.build_stream;
    // The join (and the chain preparation behind it) of a sub-batch on a stream of its own: the builder's kernels are one
    // chain per sub-batch — join, preparation, chaining, raster, row ranges — and under the fill that chain, not the fill,
    // was the cycle (44 ms per 16 384 pairs, 14 of them the join).  With the builder's arrays twice the join of c + 2 runs
    // next to the chaining of c + 1 (`band_join_serial` = 1: one stream, as before).
    // (no stream of its own: the process maps its streams onto a handful of hardware queues, and two more of them cost the
    //  full bench — a dozen streams by then — 12 % of this leg where the leg alone gained 3 %; the join shares the stream of
    //  the host-buffer flavour's sequence uploads, which it waits for anyway)
    if (!B.copy_stream && (rc = band_stream_create(&B.copy_stream, -1))) return rc;
    hipStream_t st_join = ctx->band_join_serial ? st_build : B.copy_stream;
    for (auto& s : B.set) s.built_valid = false;
    B.chained_valid = false;
    uint64_t waited_join = 0;
    // The preparation of a fill — the pair table's upload, the waits for the band, the cleared aux block and the sequences,
    // and K3v2's phase 1 (the strips before the interior runs) — does not depend on the fill before it, but on the fill
    // stream it queued behind it: 2 ms between two long kernels, every cycle.  On a stream of its own it runs under the
    // previous fill's tail (`band_pre_serial` = 1: on the fill stream as before; event timing keeps one stream).
    // (it shares the stream that clears the aux block: the clear is one of the things it waits for)
    if (!B.aux_stream && (rc = band_stream_create(&B.aux_stream, 1))) return rc;
    for (auto& s : B.set)
        if (!s.pre_done) BG_HIP(hipEventCreateWithFlags(&s.pre_done, hipEventDisableTiming));
    const bool use_pre = dev_kw != nullptr && !ctx->band_on_host && !ctx->timing && !ctx->band_window && !ctx->band_pre_serial;
    uint64_t waited_pre = 0;
    if (!B.d_started) BG_HIP(hipMalloc((void**)&B.d_started, 64));
    BG_HIP(hipMemsetAsync(B.d_started, 0, 8, st));  // [0] blocks started, [1] pairs K3p flagged
    B.started_target = 0;
    BG_HIP(hipEventRecord(B.seq_ready, st));
    BG_HIP(hipStreamWaitEvent(st_build, B.seq_ready, 0));
    if (st_join != st_build) BG_HIP(hipStreamWaitEvent(st_join, B.seq_ready, 0));

    const bool build_on_device = dev_kw != nullptr && !ctx->band_on_host;
    // First half of the device builder for the sub-batch that starts at p0: the k-mer join (B1) and the event
    // preparation of the chaining.  They get along badly with a running fill (65 VGPRs and 20 KB of LDS per block, 16 KB
    // of LDS per wavefront, against the 86 VGPRs per SIMD and 30 KB per CU two fill wavefronts leave: 10.7 + 3.2 ms alone,
    // 40 + 35 ms under the fill), while the chaining's event loop (24 VGPRs, no LDS) loses little there.  So these two
    // run for sub-batch c + 1 on the idle device just before fill c is launched — finish(c) calls this once it knows
    // where c + 1 starts and lets the fill wait for it — and the event loop of c + 1 then has the whole fill to itself.  It touches nothing of the scratch set (K4 of c - 1
    // may still be reading that), only the builder's own arrays, which raster c has finished with.
    // Sub-batch sizes: a batch that is not a whole number of sub-batches takes its REMAINDER FIRST.  A fill is one round
    // of resident wavefronts that lasts as long as its slowest pair whatever the sub-batch size, so a short sub-batch at
    // the end costs a full fill with the device idle around it (100 000 pairs = 6 x 16 384 + 1 696: 30 ms of 345); up
    // front it runs under the band construction of the first full sub-batch, which nothing else would overlap.
    const uint64_t first_want = (n_pairs > chunk_pairs && n_pairs % chunk_pairs && !ctx->band_tail_last) ? n_pairs % chunk_pairs : 0;
    auto want_at = [&](uint64_t p0) -> uint64_t {
        if (p0 == 0 && first_want) return first_want;
        return std::min<uint64_t>(chunk_pairs, n_pairs - p0);
    };
    auto issue_match = [&](uint64_t p0, uint64_t n_chunk) -> int {
        Plan& P = plan[n_chunk & 1];
        int rc = BG_OK;
        P.matched = false;
        if (!build_on_device) return BG_OK;
        const uint64_t want = want_at(p0);
        if ((rc = need_seq(st_join, waited_join, p0 + want))) return rc;
        // this parity's arrays were last read by the raster of two sub-batches ago
        if (st_join != st_build && n_chunk >= 2 && B.set[n_chunk & 1].built_valid) BG_HIP(hipStreamWaitEvent(st_join, B.set[n_chunk & 1].built, 0));
        // `band_join_late` = 1 (A/B, round 5): this join starts when the chaining of the sub-batch before it has left the
        // device.  Under a running fill the registers two fill wavefronts per SIMD leave hold EITHER that chaining OR this
        // join plus a part of it, and whichever is dispatched first keeps the other one in rounds.  Measured: ordering them
        // costs more than the race (279 against 265 ms per 100 000 pairs: the join then starves behind the raster and K4
        // instead, and the next chaining starts late) — the default leaves the dispatch order to the hardware.
        if (st_join != st_build && B.chained_valid && ctx->band_join_late) BG_HIP(hipStreamWaitEvent(st_join, B.chained, 0));
        void** db = B.db[n_chunk & 1];
        size_t* db_cap = B.db_cap[n_chunk & 1];
        uint32_t max_m = 0, max_n = 0;
        for (uint64_t q = 0; q < want; q++) {
            max_m = std::max<uint32_t>(max_m, (uint32_t)(x_off[p0 + q + 1] - x_off[p0 + q]));
            max_n = std::max<uint32_t>(max_n, (uint32_t)(y_off[p0 + q + 1] - y_off[p0 + q]));
        }
        BandDevArgs d = {};
        d.x = d_x;
        d.x_off = d_xo;
        d.y = d_y;
        d.y_off = d_yo;
        d.pair0 = p0;
        d.n_pairs = (uint32_t)want;
        d.k = dev_kw[0];
        d.w = dev_kw[1];
        d.gap_open = cs.gap_open;
        d.gap_extend = cs.gap_extend;
        d.xclip_prefix = cs.xclip_prefix;
        d.xclip_suffix = cs.xclip_suffix;
        d.yclip_prefix = cs.yclip_prefix;
        d.yclip_suffix = cs.yclip_suffix;
        d.match_score = (uint32_t)(cs.match_scores_some ? cs.match_score : 2);  // banded.rs:105,1315-1318
        d.max_m = max_m;
        d.max_n = std::max<uint32_t>(max_n, 1);
        d.table_bits = 4;
        while ((1u << d.table_bits) < 2 * d.max_n) d.table_bits++;
        d.table_size = 1u << d.table_bits;
        d.cap_matches = kMaxChainMatches + 1;
        d.chain_global = ctx->band_chain_global;
        d.chain_rows = ctx->band_chain_rows ? 1 : 0;
        d.join_global = ctx->band_join_global ? 1 : 0;
        const size_t need[17] = {(size_t)want * d.table_size * 4, (size_t)want * d.max_n * 4, (size_t)want * d.max_n * 8,
                                 (size_t)64 /* (unused) */, (size_t)want * d.cap_matches * 4,
                                 (size_t)want * d.cap_matches * 4, (size_t)want * d.cap_matches * 4,
                                 (size_t)want * d.cap_matches * 4, (size_t)want * d.cap_matches * 4,
                                 (size_t)want * d.cap_matches * 4, (size_t)want * (d.max_n + 1) * 4,
                                 (size_t)want * (d.max_n + 1) * 4, (size_t)want * sizeof(BandDevPair), (size_t)(want + 1) * 8,
                                 (size_t)want * (d.cap_matches + 1) * 16, (size_t)want * d.cap_matches * 4, (size_t)want * d.cap_matches * 2};
        for (int i = 0; i < 17; i++)
            if ((rc = bg_reserve(&db[i], &db_cap[i], std::max<size_t>(need[i], 64)))) return rc;
        d.head = (uint32_t*)db[0];
        d.next = (uint32_t*)db[1];
        d.hy = (uint64_t*)db[2];
        d.mx = (uint32_t*)db[4];
        d.my = (uint32_t*)db[5];
        d.path = (uint32_t*)db[6];
        d.qpos = (uint32_t*)db[7];
        d.upos = (uint32_t*)db[8];
        d.cont = (int32_t*)db[9];
        d.col_start = (uint32_t*)db[10];
        d.col_end = (uint32_t*)db[11];
        d.state = (BandDevPair*)db[12];
        d.row0 = (const uint64_t*)db[13];
        d.g_tree = db[14];
        d.g_score = (uint32_t*)db[15];
        d.g_back = (int16_t*)db[16];
        if ((rc = launch_band_match(d, st_join))) return rc;
        if ((rc = launch_band_chain(d, st_join, 1))) return rc;
        BG_HIP(hipEventRecord(B.set[n_chunk & 1].matched, st_join));
        P.d = d;
        P.p0 = p0;
        P.want = want;
        P.matched = true;
        return BG_OK;
    };

    auto issue = [&](uint64_t p0, uint64_t n_chunk) -> int {
        Plan& P = plan[n_chunk & 1];
        std::vector<HostPair>& hp = P.hp;
        std::vector<uint64_t>& row0 = P.row0;
        int rc = BG_OK;
        bg_band_scratch::Set& S = B.set[n_chunk & 1];
        if (build_on_device && !(P.matched && P.p0 == p0))
            if ((rc = issue_match(p0, n_chunk))) return rc;
        P.p0 = p0;
        const uint64_t want = want_at(p0);
        P.want = want;
        hp.assign(want, HostPair());
        row0.resize(want + 1);
        row0[0] = 0;
        for (uint64_t q = 0; q < want; q++) row0[q + 1] = row0[q] + (x_off[p0 + q + 1] - x_off[p0 + q]) + 1;
        // The set's staging and device buffers were last used two sub-batches ago — by K4 of sub-batch c - 2, which ends
        // ~14 ms after ITS fill.  Rounds 2-4 waited for that on the HOST, here, before launching anything of sub-batch c: the
        // chaining of c then started 5 ms into the fill of c - 1 it is meant to run under (profiles/
        // r05_banded_timeline_chain_rows.txt: chain_rows 68.85 behind K4's end at 68.79, the fill at 63.55), and the builder's
        // chain — not the fill — timed the cycle.  The chaining touches none of the set's buffers (the builder's own arrays are
        // double-buffered by parity and guarded by `built`): only the raster, which writes the set's row ranges, has to wait,
        // and it can do so on the device; the host waits in finish(), before it writes the pinned staging.  (A buffer that has
        // to GROW is freed and allocated anew: then, and for host-built bands, the host waits here as before.)
        auto set_idle = [&]() -> int {
            if (S.busy) {
                BG_HIP(hipEventSynchronize(S.traced));
                S.busy = false;
            }
            return BG_OK;
        };
        {
            const size_t need_rc = std::max<size_t>(row0[want] * sizeof(int2), 64), need_ro = std::max<size_t>(row0[want] * 4, 64);
            const bool grows = S.hc_rowc < need_rc || S.hc_roff < need_ro || S.hc_pairs < want * sizeof(BandPair) || S.dc_rowc < need_rc ||
                               S.dc_roff < need_ro || B.h_state_cap < want * sizeof(BandDevPair);
            if (grows || !build_on_device || ctx->band_host_sync)
                if ((rc = set_idle())) return rc;
        }
        if ((rc = pinned_reserve(&S.h_rowc, &S.hc_rowc, std::max<size_t>(row0[want] * sizeof(int2), 64)))) return rc;
        if ((rc = pinned_reserve(&S.h_roff, &S.hc_roff, std::max<size_t>(row0[want] * 4, 64)))) return rc;
        if ((rc = pinned_reserve(&S.h_pairs, &S.hc_pairs, want * sizeof(BandPair)))) return rc;
        int2* h_rowc = (int2*)S.h_rowc;
        uint32_t* h_roff = (uint32_t*)S.h_roff;
        (void)S.h_pairs;
        const bool on_device = build_on_device;
        P.on_device = on_device;
        if (on_device) {
            // ---- Band::create on the device (band_device.hip); the few pairs it hands back are built below.
            // The k-mer join is already on its way (issue_match)
            BandDevArgs d = P.d;
            if ((rc = bg_reserve(&S.d_rowc, &S.dc_rowc, std::max<size_t>(row0[want] * sizeof(int2), 64)))) return rc;
            if ((rc = bg_reserve(&S.d_roff, &S.dc_roff, std::max<size_t>(row0[want] * 4, 64)))) return rc;
            d.rowc = (int2*)S.d_rowc;
            d.row_off = (uint32_t*)S.d_roff;
            if ((rc = pinned_reserve(&B.h_state, &B.h_state_cap, want * sizeof(BandDevPair)))) return rc;
            if (st_join != st_build) BG_HIP(hipStreamWaitEvent(st_build, S.matched, 0));
            BG_HIP(hipMemcpyAsync((void*)d.row0, row0.data(), (want + 1) * 8, hipMemcpyHostToDevice, st_build));
            // The chaining of sub-batch c + 1 runs UNDER the fill of c: it mostly waits on memory and fits the registers
            // the fill leaves free — provided it starts after every block of the fill is resident
            // (launch_band_wait_started: the fill's grid is a single round of blocks, and a co-runner that is on a CU
            // first delays the whole kernel: fill 57 -> 116 ms).  The raster kernels start when that fill is done and
            // overlap K4 of sub-batch c.
            if (B.started_target) launch_band_wait_started(B.d_started, B.started_target, st_build);
            if ((rc = launch_band_chain(d, st_build, 2))) return rc;
            if (B.started_target) {
                BG_HIP(hipEventRecord(B.chained, st_build));
                B.chained_valid = true;
            }
            if (ctx->band_raster_late && n_chunk >= 1 && B.set[(n_chunk - 1) & 1].busy) {  // (the raster does not have to wait for the fill's epilogue)
                bg_band_scratch::Set& prev = B.set[(n_chunk - 1) & 1];
                BG_HIP(hipStreamWaitEvent(st_build, prev.fill_gone_valid ? prev.fill_gone : prev.filled, 0));
            }
            if (S.busy) BG_HIP(hipStreamWaitEvent(st_build, S.traced, 0));  // the raster writes the set's row ranges: K4 of c - 2 has read them
            if ((rc = launch_band_raster(d, st_build))) return rc;
            BG_HIP(hipMemcpyAsync(B.h_state, d.state, want * sizeof(BandDevPair), hipMemcpyDeviceToHost, st_build));
            BG_HIP(hipEventRecord(S.built, st_build));
            S.built_valid = true;
            // The builder's own arrays are free again: the join and the chain preparation of the NEXT sub-batch go right
            // behind, without waiting for the host to learn this one's sizes (finish() used to issue them: 2 ms of round trip
            // on the builder's path, and they landed in the gap between two fills).  Speculative in one respect: the next
            // sub-batch starts at p0 + want only if the scratch budget takes all of this one — otherwise finish() re-issues.
            if (p0 + want < n_pairs && !ctx->band_window)
                if ((rc = issue_match(p0 + want, n_chunk + 1))) return rc;
        } else {
        std::atomic<bool> bad_input{false};
            parallel_for(want, grain, [&](unsigned, uint64_t lo, uint64_t hi) {
                bgband::Band band;
                bgband::Workspace ws;
                for (uint64_t q = lo; q < hi; q++) {
                    const uint64_t p = p0 + q;
                    const uint32_t m = (uint32_t)(x_off[p + 1] - x_off[p]), n = (uint32_t)(y_off[p + 1] - y_off[p]);
                    if (!make_band(p, band, ws) || band.start.size() != (size_t)n + 1) {
                        bad_input = true;
                        band.reset(m, n);
                    }
                    build_pair(m, n, band, hp[q], h_rowc + row0[q], h_roff + row0[q]);
                }
            });
            if (bad_input) return BG_ERR_INVALID_ARG;
            lap("band build");
            if (trace) { fprintf(stderr, "[bg banded] cpu-ms: kmers %.1f sdp %.1f band %.1f (threads %u)\n", bgband::g_prof[0] / 1e6, bgband::g_prof[1] / 1e6, bgband::g_prof[2] / 1e6, host_threads()); }
        }
        return BG_OK;
    };

    auto finish = [&](uint64_t n_chunk, uint64_t* take_out) -> int {
        Plan& P = plan[n_chunk & 1];
        std::vector<HostPair>& hp = P.hp;
        std::vector<uint64_t>& row0 = P.row0;
        bg_band_scratch::Set& S = B.set[n_chunk & 1];
        const uint64_t p0 = P.p0, want = P.want;
        const bool on_device = P.on_device;
        int2* h_rowc = (int2*)S.h_rowc;
        uint32_t* h_roff = (uint32_t*)S.h_roff;
        BandPair* dp = (BandPair*)S.h_pairs;
        int rc = BG_OK;
        hipStream_t sp = use_pre && on_device ? B.aux_stream : st;  // (see use_pre)
        if (S.busy) {  // (issue() left this wait to the device: the host writes the set's pinned staging from here on)
            BG_HIP(hipEventSynchronize(S.traced));
            S.busy = false;
        }
        if (on_device) {
            BG_HIP(hipEventSynchronize(S.built));
            lap("band build (device)");
            const BandDevPair* hs = (const BandDevPair*)B.h_state;
            std::vector<uint64_t> redo;
            for (uint64_t q = 0; q < want; q++) {
                HostPair& h = hp[q];
                h.m = (uint32_t)(x_off[p0 + q + 1] - x_off[p0 + q]);
                h.n = (uint32_t)(y_off[p0 + q + 1] - y_off[p0 + q]);
                h.flags = hs[q].flags;
                h.cells = hs[q].cells;
                h.tb_bytes = hs[q].tb_bytes;
                h.start_0 = hs[q].start_0;
                h.end_0 = hs[q].end_0;
                h.start_n = hs[q].start_n;
                h.end_n = hs[q].end_n;
                if (h.flags == BP_HOST_FALLBACK) redo.push_back(q);
            }
            if (!redo.empty()) {
                parallel_for(redo.size(), 1, [&](unsigned, uint64_t lo, uint64_t hi) {
                    bgband::Band band;
                    bgband::Workspace ws;
                    for (uint64_t t = lo; t < hi; t++) {
                        const uint64_t q = redo[t], p = p0 + q;
                        make_band(p, band, ws);
                        build_pair(hp[q].m, hp[q].n, band, hp[q], h_rowc + row0[q], h_roff + row0[q]);
                    }
                });
                for (uint64_t q : redo) {
                    const size_t nr = (size_t)hp[q].m + 1;
                    BG_HIP(hipMemcpyAsync((int2*)S.d_rowc + row0[q], h_rowc + row0[q], nr * sizeof(int2), hipMemcpyHostToDevice, sp));
                    BG_HIP(hipMemcpyAsync((uint32_t*)S.d_roff + row0[q], h_roff + row0[q], nr * 4, hipMemcpyHostToDevice, sp));
                }
                if (trace) fprintf(stderr, "[bg banded] %zu of %llu pairs rebuilt on the host\n", redo.size(), (unsigned long long)want);
            }
        }
        // take as many pairs as fit the scratch budget (the rest is rebuilt with the next sub-batch)
        uint64_t take = 0, tbb = 0, auxw = 0;
        for (; take < want; take++) {
            const HostPair& h = hp[take];
            const uint64_t t2 = tbb + h.tb_bytes, a2 = auxw + ((BandAux(h.m, h.n).words() + 3) & ~3ull);
            if (take > 0 && t2 + a2 * 4 > budget) break;
            BandPair& d = dp[take];
            d.rowc_off = row0[take];
            d.tb_off = tbb;
            d.aux_off = auxw;
            d.start_0 = h.start_0;
            d.end_0 = h.end_0;
            d.start_n = h.start_n;
            d.end_n = h.end_n;
            d.flags = h.flags;
            d._pad = 0;
            if (band_cells) band_cells[p0 + take] = h.cells;
            tbb = t2;
            auxw = a2;
        }
        const uint64_t rows = row0[take];
        if ((rc = bg_reserve(&S.d_pairs, &S.dc_pairs, take * sizeof(BandPair)))) return rc;
        if ((rc = bg_reserve(&S.d_rowc, &S.dc_rowc, std::max<size_t>(rows * sizeof(int2), 64)))) return rc;
        if ((rc = bg_reserve(&S.d_roff, &S.dc_roff, std::max<size_t>(rows * 4, 64)))) return rc;
        if ((rc = bg_reserve(&S.d_tb, &S.dc_tb, std::max<size_t>(tbb, 64)))) return rc;
        if ((rc = bg_reserve(&S.d_aux, &S.dc_aux, std::max<size_t>(auxw * 4, 64)))) return rc;
        BG_HIP(hipMemcpyAsync(S.d_pairs, dp, take * sizeof(BandPair), hipMemcpyHostToDevice, sp));
        if (!on_device) {
            BG_HIP(hipMemcpyAsync(S.d_rowc, h_rowc, rows * sizeof(int2), hipMemcpyHostToDevice, st));
            BG_HIP(hipMemcpyAsync(S.d_roff, h_roff, rows * 4, hipMemcpyHostToDevice, st));
        }
        BG_HIP(hipEventRecord(S.copied, sp));
        // the aux block is cleared on a stream of its own: 7 GB per sub-batch, 1.8 ms that used to sit between two fills
        // (the set's previous user, K4 of two sub-batches ago, is done: issue() waited for it)
        if (!B.aux_stream && (rc = band_stream_create(&B.aux_stream, 1))) return rc;
        if (!S.cleared) BG_HIP(hipEventCreateWithFlags(&S.cleared, hipEventDisableTiming));
        BG_HIP(hipMemsetAsync(S.d_aux, 0, auxw * 4, B.aux_stream));
        BG_HIP(hipEventRecord(S.cleared, B.aux_stream));
        BG_HIP(hipStreamWaitEvent(sp, S.cleared, 0));
        a.pairs = (const BandPair*)S.d_pairs;
        a.rowc = (const int2*)S.d_rowc;
        a.row_off = (const uint32_t*)S.d_roff;
        a.tb = (uint8_t*)S.d_tb;
        a.aux = (int32_t*)S.d_aux;
        a.pair0 = p0;
        a.n_pairs = (uint32_t)take;
        if ((rc = sp == st ? need_seq(st, waited_fill, p0 + take) : need_seq(sp, waited_pre, p0 + take))) return rc;
        if (on_device) BG_HIP(hipStreamWaitEvent(sp, S.built, 0));
        if (on_device && p0 + take < n_pairs) {  // the next sub-batch's k-mer join goes first (see issue_match)
            const Plan& N = plan[(n_chunk + 1) & 1];
            if (!(N.matched && N.p0 == p0 + take))
                if ((rc = issue_match(p0 + take, n_chunk + 1))) return rc;
            // Round 3: the fill waited for the join (124 KB of LDS per block: it could only run in a window between two
            // fills).  With K3i / K3p on 32-byte rings (two blocks = 66 KB per CU) the join (91 KB) and the chaining's
            // preparation (16 KB per wavefront) fit NEXT to a fill: no window, fills back to back.
            if (ctx->band_window) BG_HIP(hipStreamWaitEvent(st, B.set[(n_chunk + 1) & 1].matched, 0));
        }
        if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st));
        // Geometry by sub-batch size: K3v2 binds a pair to 8 lanes for ~30 ms whatever the batch (throughput comes from the
        // 16 384 pairs in flight); a small sub-batch — a whole small call, or the tail of a large one — finishes sooner
        // with one pair per wavefront (K3: 19 ms; measured cross-over between 2 048 and 4 096 pairs,
        // tools/exp/time_banded_small.py).  band_fill_v1: 1 always K3, -1 never (tests)
        // (the remainder sub-batch of a large call runs first, under the next one's band construction: K3v2 / K3p there)
        const bool small_batch = take <= 2048 && ctx->band_fill_v1 >= 0 && n_pairs <= 2048;
        if (sm == SCORE_PARAMS && ctx->band_fill_v1 <= 0 && !small_batch) {
            a.started = on_device ? B.d_started : nullptr;
            a.tb_flip = kTbFlip;
            // interior runs (band_split): scaled keys, x kept whole, a real y-prefix clip — semiglobal-like scorings
            a.ring32 = ctx->band_window ? 0 : 1;
            a.split = (narrow && !ctx->band_interior_off && cs.xclip_prefix <= BG_MIN_SCORE / 2 && cs.xclip_suffix <= BG_MIN_SCORE / 2 &&
                       cs.yclip_prefix > BG_MIN_SCORE / 2) ? 1 : 0;
            // ... and, where the scoring and the lengths fit its 16-bit strip-relative keys, K3p takes the runs first
            // (banded_fill2p.hip: target / threshold as computed there; at least 2^14 key units == 1024 score units of room)
            {
                const int64_t mk = ((int64_t)a.sc.match << 4) + 12, mis = ((int64_t)-a.sc.mismatch << 4) - 10;
                const int64_t target = (0xfff0 - (mk + mis) - ((int64_t)a.sc.match << 9) - 32) & ~15ll;
                const int64_t thresh = ((int64_t)a.sc.match << 9) + 16 + ((int64_t)-a.sc.go << 4) + 32;
                a.packed = (a.split && !ctx->band_packed_off && a.sc.match >= 0 && a.sc.match <= 64 && a.sc.mismatch <= -1 &&
                            a.sc.mismatch >= -1024 && a.sc.go <= -1 && a.sc.go >= -1024 && a.sc.ge <= 0 && a.sc.ge >= -1024 &&
                            max_y < 65536 && target - thresh >= (1 << 14)) ? 1 : 0;
                a.pk_thresh = (int32_t)ctx->band_packed_thresh;
                a.redo_count = B.d_started + 1;
                a.p_block512 = ctx->band_p_block512 ? 1 : 0;
            }
            if (on_device) B.started_target += band_fill2_blocks(a.n_pairs);
            // K3v2 / K3p (K3i): eight pairs per wavefront; the epilogue goes to the traceback stream, ahead of K4 — the fill
            // stream goes straight on with the next sub-batch (event timing keeps everything on one stream)
            launch_band_fill2(a, narrow, st, S.fill_gone, ctx->timing || ctx->band_window ? nullptr : st_tb, sp != st ? sp : nullptr, S.pre_done);
            S.fill_gone_valid = true;
        }
        else {
            a.tb_flip = 0;
            a.split = 0;  // (K4 reads it too)
            if (sp != st) {
                BG_HIP(hipEventRecord(S.pre_done, sp));
                BG_HIP(hipStreamWaitEvent(st, S.pre_done, 0));
            }
            fill<<<dim3((unsigned)((take + 3) / 4)), dim3(256), 0, st>>>(a);
            S.fill_gone_valid = false;
        }
        BG_HIP(hipGetLastError());
        if (ctx->timing) {
            BG_HIP(hipEventRecord(ctx->ev[1], st));
            BG_HIP(hipEventSynchronize(ctx->ev[1]));
            float ms = 0;
            BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            ctx->last.fill_ms += ms;
            ctx->last.fill_launches += 1;
        }
        BG_HIP(hipEventRecord(S.filled, st));
        // K4 on its own stream: it overlaps the next sub-batch's K3
        BG_HIP(hipStreamWaitEvent(st_tb, S.filled, 0));
        if (ctx->timing) BG_HIP(hipEventRecord(ctx->ev[0], st_tb));
        launch_band_traceback(a, st_tb);
        BG_HIP(hipGetLastError());
        if (ctx->timing) {
            BG_HIP(hipEventRecord(ctx->ev[1], st_tb));
            BG_HIP(hipEventSynchronize(ctx->ev[1]));
            float ms = 0;
            BG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            ctx->last.traceback_ms += ms;
            ctx->last.traceback_launches += 1;
        }
        // host-buffer flavour: this sub-batch's operations go, compacted, behind those of the sub-batches before it
        // (running byte count in B.d_cell) while the next fill runs; its records get their final ops_off
        uint64_t* dl = nullptr;
        if (compact_on_device) {
            if (n_chunk && n_chunk % bg_band_scratch::kDlSlots == 0) BG_HIP(hipStreamSynchronize(B.dl_stream));  // the ring comes round
            dl = B.d_dlslot + 2 * (n_chunk % bg_band_scratch::kDlSlots);
            if ((rc = bg_compact_ops_dev(d_out + p0, take, d_ops, (uint8_t*)B.d_cmp, true, B.d_cell, dl, B.d_cscan, true, st_tb))) return rc;
            BG_HIP(hipMemcpyAsync(dl + 1, B.d_cell, 8, hipMemcpyDeviceToDevice, st_tb));
        }
        BG_HIP(hipEventRecord(S.traced, st_tb));
        if (dl) {
            BG_HIP(hipStreamWaitEvent(B.dl_stream, S.traced, 0));
            if ((rc = bg_range_to_host((const uint8_t*)B.d_cmp, (uint8_t*)B.h_ops, dl, B.dl_stream))) return rc;
        }
        S.busy = true;
        lap("enqueue");
        *take_out = take;
        return BG_OK;
    };

    {
        uint64_t p0 = 0, n_chunk = 0;
        if ((rc = issue(0, 0))) return rc;
        // The caller's buffers are pageable: an upload keeps this thread inside the copy call for its whole duration
        // (~30 ms per slice), and the device idle if the next launches wait behind it.  So slice c + 2 goes up right
        // after the fill of sub-batch c has been launched (44 ms of kernels to hide behind), one slice per round — not
        // all of them after the first issue (measured: the first round took 90 ms instead of 58).
        if ((rc = upload_slices(2))) return rc;
        for (;;) {
            uint64_t take = 0;
            if ((rc = finish(n_chunk, &take))) return rc;
            p0 += take;
            if (p0 >= n_pairs) break;
            if ((rc = upload_slices(n_chunk + 3))) return rc;
            n_chunk++;
            if ((rc = issue(p0, n_chunk))) return rc;
        }
    }
    if (dio) {  // everything stays in HBM; records keep the strided ops_off like bg_align_batch_dev
        BG_HIP(hipStreamSynchronize(st_tb));
        BG_HIP(hipStreamSynchronize(st));
        for (auto& s : B.set) s.busy = false;
        lap("drain");
        return BG_OK;
    }
    // results: the records (their ops_off final) and the compact operations come back on the traceback stream
    BG_HIP(hipMemcpyAsync(out, d_out, n_pairs * sizeof(bg_alignment_t), hipMemcpyDeviceToHost, st_tb));
    uint64_t used = 0;
    if (compact_on_device) BG_HIP(hipMemcpyAsync(&used, B.d_cell, 8, hipMemcpyDeviceToHost, st_tb));
    BG_HIP(hipStreamSynchronize(st_tb));
    BG_HIP(hipStreamSynchronize(st));
    for (auto& s : B.set) s.busy = false;
    // (the operations are in B.h_ops once the download stream has drained: synchronised below)
    int status = BG_OK;
    bool cap_hit = false;
    uint64_t fit = used;  // bytes of whole pairs that fit the caller's buffer
    for (uint64_t p = 0; p < n_pairs; p++) {  // (while the operations travel)
        if (out[p].status && status == BG_OK) status = out[p].status;
        if (!compact_on_device) {  // no operations wanted: offsets of an (empty) compact buffer all the same
            out[p].ops_off = used;
            used += out[p].n_ops;
        } else if (!cap_hit && out[p].ops_off + out[p].n_ops > ops_cap) {
            cap_hit = true;
            fit = out[p].ops_off;
        }
    }
    if (cap_hit && status == BG_OK) status = BG_ERR_OPS_CAP;
    if (compact_on_device && used) {
        BG_HIP(hipStreamSynchronize(B.dl_stream));
        lap("drain + d2h");
        const uint8_t* h_ops = (const uint8_t*)B.h_ops;
        const uint64_t nb = std::min(fit, ops_cap);
        parallel_for(nb, 1 << 20, [&](unsigned, uint64_t lo, uint64_t hi) { memcpy(ops_buf + lo, h_ops + lo, hi - lo); });
    }
    if (ops_used) *ops_used = used;
    lap("compact ops");
    return status;
}

extern "C" int bg_align_banded_batch(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w,
                                     uint64_t n_pairs, const uint8_t* x, const uint64_t* x_off, const uint8_t* y,
                                     const uint64_t* y_off, bg_alignment_t* out, uint8_t* ops_buf, uint64_t ops_cap,
                                     uint64_t* ops_used, uint64_t* band_cells) {
    if (!sc || mode < BG_MODE_CUSTOM || mode > BG_MODE_LOCAL) return BG_ERR_INVALID_ARG;
    const bgband::ClipScores cs = clip_scores(sc, mode);
    const uint32_t kw[2] = {k, w};
    return banded_batch_impl(ctx, sc, mode, n_pairs, x, x_off, y, y_off, out, ops_buf, ops_cap, ops_used, band_cells,
                             [&](uint64_t p, bgband::Band& band, bgband::Workspace& ws) {
                                 band.create(x + x_off[p], (size_t)(x_off[p + 1] - x_off[p]), y + y_off[p],
                                             (size_t)(y_off[p + 1] - y_off[p]), k, w, cs, ws);
                                 return true;
                             },
                             kw);
}

// Device-resident flavour of bg_align_banded_batch: sequences, offsets, records and operation slots are
// device pointers (records keep ops_off = (p + 1) * ops_stride - n_ops, operations right-aligned in their
// slot, as bg_align_batch_dev leaves them).  Synchronous: it drives its own streams and returns when
// the results are in place; work queued on `stream` before the call is waited for first.
extern "C" int bg_align_banded_batch_dev(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w,
                                         uint64_t n_pairs, const uint8_t* d_x, const uint64_t* d_x_off, const uint8_t* d_y,
                                         const uint64_t* d_y_off, bg_alignment_t* d_out, uint8_t* d_ops, uint64_t ops_stride,
                                         uint64_t* band_cells, void* stream) {
    if (!ctx || !sc || mode < BG_MODE_CUSTOM || mode > BG_MODE_LOCAL) return BG_ERR_INVALID_ARG;
    if (n_pairs == 0) return BG_OK;
    if (!d_x_off || !d_y_off || !d_out) return BG_ERR_INVALID_ARG;
    BG_HIP(hipSetDevice(ctx->device));
    BG_HIP(hipStreamSynchronize((hipStream_t)stream));
    // the host side of the pipeline needs the lengths
    std::vector<uint64_t> x_off(n_pairs + 1), y_off(n_pairs + 1);
    BG_HIP(hipMemcpy(x_off.data(), d_x_off, (n_pairs + 1) * 8, hipMemcpyDeviceToHost));
    BG_HIP(hipMemcpy(y_off.data(), d_y_off, (n_pairs + 1) * 8, hipMemcpyDeviceToHost));
    const bgband::ClipScores cs = clip_scores(sc, mode);
    const uint32_t kw[2] = {k, w};
    const BandDevIO dio = {d_x, d_x_off, d_y, d_y_off, d_out, d_ops, ops_stride};
    return banded_batch_impl(ctx, sc, mode, n_pairs, nullptr, x_off.data(), nullptr, y_off.data(), nullptr, nullptr, 0, nullptr,
                             band_cells,
                             [&](uint64_t p, bgband::Band& band, bgband::Workspace& ws) {
                                 // a pair the device builder hands back: fetch its sequences for the host builder
                                 const size_t m = (size_t)(x_off[p + 1] - x_off[p]), n = (size_t)(y_off[p + 1] - y_off[p]);
                                 std::vector<uint8_t> hx(m + 1), hy(n + 1);
                                 if (m && hipMemcpy(hx.data(), d_x + x_off[p], m, hipMemcpyDeviceToHost) != hipSuccess) return false;
                                 if (n && hipMemcpy(hy.data(), d_y + y_off[p], n, hipMemcpyDeviceToHost) != hipSuccess) return false;
                                 band.create(hx.data(), m, hy.data(), n, k, w, cs, ws);
                                 return true;
                             },
                             kw, &dio);
}

// compute_alignment (banded.rs:406-869) over caller-supplied bands: n + 1 half-open row ranges per pair at
// band_off[p] — what custom_with_matches / custom_with_match_path / custom_with_expanded_matches /
// *_with_prehash (banded.rs:294-401, 938-970) reach after building their band on the host.
extern "C" int bg_align_banded_bands_batch(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs, const uint8_t* x,
                                           const uint64_t* x_off, const uint8_t* y, const uint64_t* y_off,
                                           const uint64_t* band_off, const uint32_t* band_start, const uint32_t* band_end,
                                           bg_alignment_t* out, uint8_t* ops_buf, uint64_t ops_cap, uint64_t* ops_used,
                                           uint64_t* band_cells) {
    if (n_pairs && (!band_off || !band_start || !band_end)) return BG_ERR_INVALID_ARG;
    return banded_batch_impl(ctx, sc, mode, n_pairs, x, x_off, y, y_off, out, ops_buf, ops_cap, ops_used, band_cells,
                             [&](uint64_t p, bgband::Band& band, bgband::Workspace&) {
                                 const size_t m = (size_t)(x_off[p + 1] - x_off[p]), n = (size_t)(y_off[p + 1] - y_off[p]);
                                 band.reset(m, n);
                                 for (size_t j = 0; j <= n; j++) {
                                     band.start[j] = band_start[band_off[p] + j];
                                     band.end[j] = band_end[band_off[p] + j];
                                     if (band.end[j] > m + 1 && band.end[j] > band.start[j]) return false;
                                 }
                                 return true;
                             });
}

namespace {
std::vector<bgband::Match> to_matches(const uint32_t* xy, uint64_t n) {
    std::vector<bgband::Match> v(n);
    for (uint64_t i = 0; i < n; i++) v[i] = {xy[2 * i], xy[2 * i + 1]};
    return v;
}
}  // namespace

// Band::create_with_matches (banded.rs:1301-1328; path == NULL) or Band::create_from_match_path
// (1330-1367) for a batch; matches of pair p are matches_xy[2*match_off[p] .. 2*match_off[p+1]).
extern "C" int bg_band_from_matches_batch(const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w, uint64_t n_pairs,
                                          const uint64_t* x_off, const uint64_t* y_off, const uint32_t* matches_xy,
                                          const uint64_t* match_off, const uint32_t* path, const uint64_t* path_off,
                                          const uint64_t* band_off, uint32_t* start, uint32_t* end, uint64_t* band_cells) {
    if (!sc || !x_off || !y_off || !match_off || (n_pairs && (!band_off || !start || !end))) return BG_ERR_INVALID_ARG;
    if (path && !path_off) return BG_ERR_INVALID_ARG;
    int rc = check_scoring(sc);
    if (rc) return rc;
    const bgband::ClipScores cs = clip_scores(sc, mode);
    std::atomic<bool> bad{false};
    parallel_for(n_pairs, 4, [&](unsigned, uint64_t lo, uint64_t hi) {
        bgband::Band band;
        bgband::Workspace ws;
        for (uint64_t p = lo; p < hi; p++) {
            const uint32_t m = (uint32_t)(x_off[p + 1] - x_off[p]), n = (uint32_t)(y_off[p + 1] - y_off[p]);
            const std::vector<bgband::Match> mm = to_matches(matches_xy + 2 * match_off[p], match_off[p + 1] - match_off[p]);
            bool ok = true;
            if (path) {
                std::vector<uint32_t> pp(path + path_off[p], path + path_off[p + 1]);
                for (uint32_t idx : pp) ok = ok && idx < mm.size();
                if (!mm.empty() && pp.empty()) ok = false;  // path[0] panics in the reference
                if (ok) band.create_from_match_path(m, n, k, w, cs, pp, mm);
            } else {
                ok = band.create_with_matches(m, n, k, w, cs, mm, ws);
            }
            if (!ok) {
                bad = true;
                continue;
            }
            memcpy(start + band_off[p], band.start.data(), (size_t)(n + 1) * 4);
            memcpy(end + band_off[p], band.end.data(), (size_t)(n + 1) * 4);
            if (band_cells) band_cells[p] = band.num_cells();
        }
    });
    return bad ? BG_ERR_INVALID_ARG : BG_OK;
}

// ---- sparse.rs helpers on the host (callers of custom_with_matches / _expanded_matches need them) ----
// All take matches as (x, y) uint32 pairs; return the number of entries the result has (which may exceed
// `cap`: call again with a larger buffer), or UINT64_MAX when the reference would assert (unsorted matches).
extern "C" uint64_t bg_sparse_find_kmer_matches(const uint8_t* x, uint64_t m, const uint8_t* y, uint64_t n, uint32_t k,
                                                uint32_t* out_xy, uint64_t cap) {
    std::vector<bgband::Match> mm;
    bgband::find_kmer_matches(x, m, y, n, k, mm);
    for (uint64_t i = 0; i < mm.size() && i < cap; i++) {
        out_xy[2 * i] = mm[i].x;
        out_xy[2 * i + 1] = mm[i].y;
    }
    return mm.size();
}

extern "C" uint64_t bg_sparse_sdpkpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k, uint32_t match_score,
                                     int32_t gap_open, int32_t gap_extend, uint32_t* path, uint64_t cap) {
    std::vector<uint32_t> pp;
    const auto mm = to_matches(matches_xy, n_matches);
    for (size_t i = 1; i < mm.size(); i++)
        if (!(mm[i - 1] < mm[i])) return UINT64_MAX;
    if (!bgband::sdpkpp_path(mm, k, match_score, gap_open, gap_extend, pp)) return UINT64_MAX;
    for (uint64_t i = 0; i < pp.size() && i < cap; i++) path[i] = pp[i];
    return pp.size();
}

extern "C" uint64_t bg_sparse_lcskpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k, uint32_t* path, uint64_t cap,
                                     uint32_t* score) {
    std::vector<uint32_t> pp;
    if (!bgband::lcskpp_path(to_matches(matches_xy, n_matches), k, pp, score)) return UINT64_MAX;
    for (uint64_t i = 0; i < pp.size() && i < cap; i++) path[i] = pp[i];
    return pp.size();
}

extern "C" uint64_t bg_sparse_sdpkpp_union_lcskpp_path(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k,
                                                       uint32_t match_score, int32_t gap_open, int32_t gap_extend,
                                                       uint32_t* path, uint64_t cap) {
    std::vector<uint32_t> pp;
    if (!bgband::sdpkpp_union_lcskpp_path(to_matches(matches_xy, n_matches), k, match_score, gap_open, gap_extend, pp))
        return UINT64_MAX;
    for (uint64_t i = 0; i < pp.size() && i < cap; i++) path[i] = pp[i];
    return pp.size();
}

extern "C" uint64_t bg_sparse_expand_kmer_matches(const uint8_t* x, uint64_t m, const uint8_t* y, uint64_t n, uint32_t k,
                                                  const uint32_t* matches_xy, uint64_t n_matches, uint32_t allowed_mismatches,
                                                  uint32_t* out_xy, uint64_t cap) {
    std::vector<bgband::Match> out;
    if (!bgband::expand_kmer_matches(x, m, y, n, k, to_matches(matches_xy, n_matches), allowed_mismatches, out)) return UINT64_MAX;
    for (uint64_t i = 0; i < out.size() && i < cap; i++) {
        out_xy[2 * i] = out[i].x;
        out_xy[2 * i + 1] = out[i].y;
    }
    return out.size();
}
