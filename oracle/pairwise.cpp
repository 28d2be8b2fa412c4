// ORACLE (test infrastructure, not product code) — see oracle.h.
//
// Line-faithful restatement of rust-bio 4.0.1 `bio::alignment::pairwise::Aligner`
// (/root/reference/src/alignment/pairwise/mod.rs).  Every block cites the lines it follows.
// Integer arithmetic is i32 and wraps like a Rust release build (compiled with -fwrapv).
#include "pairwise_impl.h"

#include <atomic>
#include <thread>

namespace orc {

// Test hook (tests/test_oracle_lf_property.py; off unless a test switches it on): the one thing the LF flavour of the
// engine's local kernel (rust-bio_amd/csrc/sw_fill_pk16.inc) leaves out of this algorithm — the x-suffix-clip fold of the
// columns before n — so that "no local alignment needs it" can be checked against the restatement itself, on the CPU:
// with the hook on, custom() keeps no fold for 0 < j < n and a traceback that asks for such an Lx[j] throws.
// The hook is a TEMPLATE parameter of the restatement: custom() reads the flag once and runs custom_impl<false> — whose
// inner loop is byte for byte the reference's, and what bench.py times as cpu_baseline — unless a test switched it on.
std::atomic<int> g_lf_hook{0};
std::atomic<uint64_t> g_lf_lx_reads{0};

Alignment Aligner::custom(const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
    return g_lf_hook.load(std::memory_order_relaxed) ? custom_impl<true>(x, m, y, n) : custom_impl<false>(x, m, y, n);
}

// pairwise/mod.rs:591-922  Aligner::custom
template <bool LF_HOOK>
Alignment Aligner::custom_impl(const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
    const Scoring& sc = scoring;
    traceback.init(m, n);  // mod.rs:593

    // mod.rs:597-672 — initial conditions, both rolling buffers
    for (int k = 0; k < 2; k++) {
        I[k].assign(m + 1, MIN_SCORE);
        D[k].assign(m + 1, MIN_SCORE);
        S[k].assign(m + 1, MIN_SCORE);
        S[k][0] = 0;

        if (k == 0) {
            TracebackCell tb;
            tb.set_all(TB_START);
            traceback.set(0, 0, tb);
            Lx.assign(n + 1, 0);
            Ly.assign(m + 1, 0);
            Sn.assign(m + 1, MIN_SCORE);
            Sn[0] = sc.yclip_suffix;
            Ly[0] = n;
        }

        for (size_t i = 1; i <= m; i++) {
            TracebackCell tb;
            tb.set_all(TB_START);
            if (i == 1) {
                I[k][i] = sc.gap_open;
                tb.set_i_bits(TB_START);
            } else {
                // Insert all i characters
                int32_t i_score = sc.gap_open + sc.gap_extend * ((int32_t)i - 1);
                int32_t c_score = sc.xclip_prefix + sc.gap_open;  // Clip then insert
                if (i_score > c_score) {
                    I[k][i] = i_score;
                    tb.set_i_bits(TB_INS);
                } else {
                    I[k][i] = c_score;
                    tb.set_i_bits(TB_XCLIP_PREFIX);
                }
            }

            if (i == m) {
                tb.set_s_bits(TB_XCLIP_SUFFIX);
            } else {
                S[k][i] = MIN_SCORE;
            }

            if (I[k][i] > S[k][i]) {
                S[k][i] = I[k][i];
                tb.set_s_bits(TB_INS);
            }

            if (sc.xclip_prefix > S[k][i]) {
                S[k][i] = sc.xclip_prefix;
                tb.set_s_bits(TB_XCLIP_PREFIX);
            }

            // Track the score if we do a suffix clip (x) after this character
            if (i != m && S[k][i] + sc.xclip_suffix > S[k][m]) {
                S[k][m] = S[k][i] + sc.xclip_suffix;
                Lx[0] = m - i;
            }

            if (k == 0) traceback.set(i, 0, tb);
            // Track the score if we do suffix clip (y) from here
            if (S[k][i] + sc.yclip_suffix > Sn[i]) {
                Sn[i] = S[k][i] + sc.yclip_suffix;
                Ly[i] = n;
            }
        }
    }

    // mod.rs:674-806 — fill
    for (size_t j = 1; j <= n; j++) {
        const size_t curr = j % 2;
        const size_t prev = 1 - curr;

        {
            // mod.rs:678-717 — i = 0
            TracebackCell tb;
            I[curr][0] = MIN_SCORE;

            if (j == 1) {
                D[curr][0] = sc.gap_open;
                tb.set_d_bits(TB_START);
            } else {
                int32_t d_score = sc.gap_open + sc.gap_extend * ((int32_t)j - 1);
                int32_t c_score = sc.yclip_prefix + sc.gap_open;
                if (d_score > c_score) {
                    D[curr][0] = d_score;
                    tb.set_d_bits(TB_DEL);
                } else {
                    D[curr][0] = c_score;
                    tb.set_d_bits(TB_YCLIP_PREFIX);
                }
            }
            if (D[curr][0] > sc.yclip_prefix) {
                S[curr][0] = D[curr][0];
                tb.set_s_bits(TB_DEL);
            } else {
                S[curr][0] = sc.yclip_prefix;
                tb.set_s_bits(TB_YCLIP_PREFIX);
            }

            if (j == n && Sn[0] > S[curr][0]) {
                S[curr][0] = Sn[0];
                tb.set_s_bits(TB_YCLIP_SUFFIX);
            } else if (S[curr][0] + sc.yclip_suffix > Sn[0]) {
                Sn[0] = S[curr][0] + sc.yclip_suffix;
                Ly[0] = n - j;
            }

            traceback.set(0, j, tb);
        }

        for (size_t i = 1; i <= m; i++) S[curr][i] = MIN_SCORE;  // mod.rs:719-721

        const uint8_t q = y[j - 1];
        const int32_t xclip_score =
            sc.xclip_prefix +
            std::max(sc.yclip_prefix, sc.gap_open + sc.gap_extend * ((int32_t)j - 1));
        for (size_t i = 1; i < m + 1; i++) {
            const uint8_t p = x[i - 1];
            TracebackCell tb;

            int32_t m_score = S[prev][i - 1] + sc.score(p, q);

            int32_t i_score = I[curr][i - 1] + sc.gap_extend;
            int32_t s_score = S[curr][i - 1] + sc.gap_open;
            int32_t best_i_score;
            if (i_score > s_score) {
                best_i_score = i_score;
                tb.set_i_bits(TB_INS);
            } else {
                best_i_score = s_score;
                tb.set_i_bits(traceback.get(i - 1, j).get_s_bits());
            }

            int32_t d_score = D[prev][i] + sc.gap_extend;
            s_score = S[prev][i] + sc.gap_open;
            int32_t best_d_score;
            if (d_score > s_score) {
                best_d_score = d_score;
                tb.set_d_bits(TB_DEL);
            } else {
                best_d_score = s_score;
                tb.set_d_bits(traceback.get(i, j - 1).get_s_bits());
            }

            tb.set_s_bits(TB_XCLIP_SUFFIX);
            int32_t best_s_score = S[curr][i];

            if (m_score > best_s_score) {
                best_s_score = m_score;
                tb.set_s_bits(p == q ? TB_MATCH : TB_SUBST);
            }
            if (best_i_score > best_s_score) {
                best_s_score = best_i_score;
                tb.set_s_bits(TB_INS);
            }
            if (best_d_score > best_s_score) {
                best_s_score = best_d_score;
                tb.set_s_bits(TB_DEL);
            }
            if (xclip_score > best_s_score) {
                best_s_score = xclip_score;
                tb.set_s_bits(TB_XCLIP_PREFIX);
            }
            int32_t yclip_score =
                sc.yclip_prefix + sc.gap_open + sc.gap_extend * ((int32_t)i - 1);
            if (yclip_score > best_s_score) {
                best_s_score = yclip_score;
                tb.set_s_bits(TB_YCLIP_PREFIX);
            }

            S[curr][i] = best_s_score;
            I[curr][i] = best_i_score;
            D[curr][i] = best_d_score;

            // Track the score if we do suffix clip (x) from here
            if (!(LF_HOOK && j != n) &&  // (test hook, compiled out of custom_impl<false>: see the top of this file)
                S[curr][i] + sc.xclip_suffix > S[curr][m]) {
                S[curr][m] = S[curr][i] + sc.xclip_suffix;
                Lx[j] = m - i;
            }
            // Track the score if we do suffix clip (y) from here
            if (S[curr][i] + sc.yclip_suffix > Sn[i]) {
                Sn[i] = S[curr][i] + sc.yclip_suffix;
                Ly[i] = n - j;
            }

            traceback.set(i, j, tb);
        }
    }

    // mod.rs:808-821 — suffix clipping in the j = n column
    for (size_t i = 0; i <= m; i++) {
        const size_t j = n;
        const size_t curr = j % 2;
        if (Sn[i] > S[curr][i]) {
            S[curr][i] = Sn[i];
            traceback.get_mut(i, j).set_s_bits(TB_YCLIP_SUFFIX);
        }
        if (S[curr][i] + sc.xclip_suffix > S[curr][m]) {
            S[curr][m] = S[curr][i] + sc.xclip_suffix;
            Lx[j] = m - i;
            traceback.get_mut(m, j).set_s_bits(TB_XCLIP_SUFFIX);
        }
    }

    // mod.rs:823-843 — recompute the last column of I
    for (size_t i = 1; i <= m; i++) {
        const size_t j = n;
        const size_t curr = j % 2;
        int32_t s_score = S[curr][i - 1] + sc.gap_open;
        if (s_score > I[curr][i]) {
            I[curr][i] = s_score;
            uint16_t s_bit = traceback.get(i - 1, j).get_s_bits();
            traceback.get_mut(i, j).set_i_bits(s_bit);
        }
        if (s_score > S[curr][i]) {
            S[curr][i] = s_score;
            traceback.get_mut(i, j).set_s_bits(TB_INS);
            if (S[curr][i] + sc.xclip_suffix > S[curr][m]) {
                S[curr][m] = S[curr][i] + sc.xclip_suffix;
                Lx[j] = m - i;
                traceback.get_mut(m, j).set_s_bits(TB_XCLIP_SUFFIX);
            }
        }
    }

    // mod.rs:845-921 — traceback
    size_t i = m, j = n;
    std::vector<Op> operations;
    operations.reserve(m);
    size_t xstart = 0, ystart = 0, xend = m, yend = n;

    uint16_t last_layer = traceback.get(i, j).get_s_bits();
    const size_t guard = 4 * (m + n) + 64;  // the reference would loop forever / OOM
    for (size_t steps = 0;; steps++) {
        if (steps > guard) throw OracleError("traceback does not terminate");
        uint16_t next_layer;
        if (last_layer == TB_START) break;
        switch (last_layer) {
            case TB_INS:
                operations.push_back({ORC_OP_INS, 0});
                next_layer = traceback.get(i, j).get_i_bits();
                i -= 1;
                break;
            case TB_DEL:
                operations.push_back({ORC_OP_DEL, 0});
                next_layer = traceback.get(i, j).get_d_bits();
                j -= 1;
                break;
            case TB_MATCH:
                operations.push_back({ORC_OP_MATCH, 0});
                next_layer = traceback.get(i - 1, j - 1).get_s_bits();
                i -= 1;
                j -= 1;
                break;
            case TB_SUBST:
                operations.push_back({ORC_OP_SUBST, 0});
                next_layer = traceback.get(i - 1, j - 1).get_s_bits();
                i -= 1;
                j -= 1;
                break;
            case TB_XCLIP_PREFIX:
                operations.push_back({ORC_OP_XCLIP, i});
                xstart = i;
                i = 0;
                next_layer = traceback.get(0, j).get_s_bits();
                break;
            case TB_XCLIP_SUFFIX:
                if (LF_HOOK && j != n && j != 0) {  // (test hook)
                    g_lf_lx_reads++;
                    throw OracleError("LF hook: the traceback asked for Lx[j] of a column before n");
                }
                operations.push_back({ORC_OP_XCLIP, Lx[j]});
                i -= Lx[j];
                xend = i;
                next_layer = traceback.get(i, j).get_s_bits();
                break;
            case TB_YCLIP_PREFIX:
                operations.push_back({ORC_OP_YCLIP, j});
                ystart = j;
                j = 0;
                next_layer = traceback.get(i, 0).get_s_bits();
                break;
            case TB_YCLIP_SUFFIX:
                operations.push_back({ORC_OP_YCLIP, Ly[i]});
                j -= Ly[i];
                yend = j;
                next_layer = traceback.get(i, j).get_s_bits();
                break;
            default:
                throw OracleError("Dint expect this!");
        }
        last_layer = next_layer;
    }

    std::reverse(operations.begin(), operations.end());
    Alignment a;
    a.score = S[n % 2][m];
    a.ystart = ystart;
    a.xstart = xstart;
    a.yend = yend;
    a.xend = xend;
    a.ylen = n;
    a.xlen = m;
    a.operations = std::move(operations);
    a.mode = ORC_MODE_CUSTOM;
    return a;
}

// mod.rs:925-1015 — the three wrappers save, overwrite and restore the clip penalties
Alignment Aligner::with_clips(int32_t xp, int32_t xs, int32_t yp, int32_t ys, int mode,
                              bool filter, const uint8_t* x, size_t m, const uint8_t* y,
                              size_t n) {
    int32_t saved[4] = {scoring.xclip_prefix, scoring.xclip_suffix, scoring.yclip_prefix,
                        scoring.yclip_suffix};
    scoring.xclip_prefix = xp;
    scoring.xclip_suffix = xs;
    scoring.yclip_prefix = yp;
    scoring.yclip_suffix = ys;
    Alignment a = custom(x, m, y, n);
    a.mode = mode;
    if (filter) a.filter_clip_operations();
    scoring.xclip_prefix = saved[0];
    scoring.xclip_suffix = saved[1];
    scoring.yclip_prefix = saved[2];
    scoring.yclip_suffix = saved[3];
    return a;
}

Alignment Aligner::global(const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
    return with_clips(MIN_SCORE, MIN_SCORE, MIN_SCORE, MIN_SCORE, ORC_MODE_GLOBAL, false, x, m, y,
                      n);  // mod.rs:925-951 (no filter)
}
Alignment Aligner::semiglobal(const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
    return with_clips(MIN_SCORE, MIN_SCORE, 0, 0, ORC_MODE_SEMIGLOBAL, true, x, m, y,
                      n);  // mod.rs:954-983
}
Alignment Aligner::local(const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
    return with_clips(0, 0, 0, 0, ORC_MODE_LOCAL, true, x, m, y, n);  // mod.rs:986-1015
}

Alignment Aligner::run(int mode, const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
    switch (mode) {
        case ORC_MODE_GLOBAL: return global(x, m, y, n);
        case ORC_MODE_SEMIGLOBAL: return semiglobal(x, m, y, n);
        case ORC_MODE_LOCAL: return local(x, m, y, n);
        default: return custom(x, m, y, n);
    }
}

int export_alignment(const Alignment& a, orc_alignment_t* out, uint64_t* ops, uint64_t ops_cap) {
    out->score = a.score;
    out->ystart = a.ystart;
    out->xstart = a.xstart;
    out->yend = a.yend;
    out->xend = a.xend;
    out->ylen = a.ylen;
    out->xlen = a.xlen;
    out->n_ops = a.operations.size();
    out->mode = a.mode;
    if (a.operations.size() > ops_cap) return -1;
    for (size_t t = 0; t < a.operations.size(); t++)
        ops[t] = (uint64_t)a.operations[t].kind | ((uint64_t)a.operations[t].len << 8);
    return 0;
}

Scoring scoring_from_c(const orc_scoring_t* sc) {
    Scoring s;
    s.gap_open = sc->gap_open;
    s.gap_extend = sc->gap_extend;
    s.xclip_prefix = sc->xclip_prefix;
    s.xclip_suffix = sc->xclip_suffix;
    s.yclip_prefix = sc->yclip_prefix;
    s.yclip_suffix = sc->yclip_suffix;
    s.match_score = sc->match_score;
    s.mismatch_score = sc->mismatch_score;
    s.match_scores_some = sc->match_scores_some != 0;
    s.matrix = sc->matrix;
    return s;
}

}  // namespace orc

extern "C" void orc_test_lf_hook(int on) { orc::g_lf_hook = on; }
extern "C" uint64_t orc_test_lf_lx_reads(void) { return orc::g_lf_lx_reads.load(); }

extern "C" int orc_align(const orc_scoring_t* sc, int mode, const uint8_t* x, uint64_t m,
                         const uint8_t* y, uint64_t n, orc_alignment_t* out, uint64_t* ops,
                         uint64_t ops_cap) {
    try {
        orc::Aligner al(orc::scoring_from_c(sc));
        orc::Alignment a = al.run(mode, x, m, y, n);
        return orc::export_alignment(a, out, ops, ops_cap);
    } catch (const orc::OracleError&) {
        return -2;
    }
}

extern "C" int orc_align_batch(const orc_scoring_t* sc, int mode, uint64_t n_pairs,
                               const uint8_t* x, const uint64_t* x_off, const uint8_t* y,
                               const uint64_t* y_off, orc_alignment_t* out, uint64_t* ops,
                               uint64_t ops_stride, int threads) {
    if (threads < 1) threads = 1;
    std::vector<int> rc(threads, 0);
    auto work = [&](int t) {
        orc::Aligner al(orc::scoring_from_c(sc));  // one Aligner per thread, reused per pair
        for (uint64_t p = t; p < n_pairs; p += threads) {
            try {
                orc::Alignment a = al.run(mode, x + x_off[p], x_off[p + 1] - x_off[p],
                                          y + y_off[p], y_off[p + 1] - y_off[p]);
                int r = orc::export_alignment(a, &out[p], ops ? ops + p * ops_stride : nullptr,
                                              ops ? ops_stride : 0);
                if (r && ops) rc[t] = r;
            } catch (const orc::OracleError&) {
                rc[t] = -2;
            }
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (int r : rc)
        if (r) return r;
    return 0;
}
