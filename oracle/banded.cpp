// ORACLE (test infrastructure, not product code) — see oracle.h.
//
// Line-faithful restatement of rust-bio 4.0.1 `bio::alignment::pairwise::banded`
// (/root/reference/src/alignment/pairwise/banded.rs): Band (1047-1380), compute_alignment
// (406-869) and the mode wrappers (872-1004).  Arithmetic wraps like a Rust release build.
#include <thread>

#include "pairwise_impl.h"
#include "sparse_impl.h"

namespace orc {

constexpr size_t MAX_CELLS = 5000000;        // banded.rs:104
constexpr int32_t DEFAULT_MATCH_SCORE = 2;   // banded.rs:105

static inline size_t sat_sub(size_t a, size_t b) { return a > b ? a - b : 0; }

// banded.rs:1047-1380
struct Band {
    size_t rows = 0, cols = 0;
    std::vector<size_t> start, end;  // ranges[j] = start[j]..end[j]

    Band() {}
    Band(size_t m, size_t n) : rows(m + 1), cols(n + 1), start(n + 1, m + 1), end(n + 1, 0) {}  // 1061-1067

    // banded.rs:1071-1107
    void add_kmer(Match s, size_t k, size_t w) {
        const size_t r = s.first, c = s.second;
        if (k == 0) return;
        size_t i = sat_sub(r, w);
        for (size_t j = sat_sub(c, w); j < std::min(c + w + 1, cols); j++) start[j] = std::min(start[j], i);
        i = sat_sub(r, w);
        for (size_t j = std::min(c + w, cols); j < std::min(c + k + w, cols); j++) {
            start[j] = std::min(start[j], i);
            i += 1;
        }
        i = r + w + k;
        size_t j = sat_sub(c + k - 1, w);
        for (;;) {
            if (j <= sat_sub(c, w)) break;
            j -= 1;
            i -= 1;
            end[j] = std::max(end[j], std::min(i, rows));
        }
        i = std::min(r + w + k, rows);
        for (size_t jj = sat_sub(c + k - 1, w); jj < std::min(c + k + w, cols); jj++) end[jj] = std::max(end[jj], i);
    }

    // banded.rs:1111-1120
    void add_entry(Match pos, size_t w) {
        const size_t r = pos.first, c = pos.second;
        const size_t istart = sat_sub(r, w);
        const size_t iend = std::min(r + w + 1, rows);
        for (size_t j = sat_sub(c, w); j < std::min(c + w + 1, cols); j++) {
            start[j] = std::min(start[j], istart);
            end[j] = std::max(end[j], iend);
        }
    }

    // banded.rs:1123-1137 — u32 arithmetic, wrapping like a release build
    void add_gap(Match s, Match e, size_t w) {
        const uint32_t nrows = e.first - s.first;
        const uint32_t ncols = e.second - s.second;
        if (nrows > ncols) {
            for (uint32_t r = s.first; r < e.first; r++) {
                const uint32_t c = s.second + (e.second - s.second) * (r - s.first) / (e.first - s.first);
                add_entry({r, c}, w);
            }
        } else {
            for (uint32_t c = s.second; c < e.second; c++) {
                const uint32_t r = s.first + (e.first - s.first) * (c - s.second) / (e.second - s.second);
                add_entry({r, c}, w);
            }
        }
    }

    // banded.rs:1150-1276
    void set_boundaries(Match first, Match last, size_t k, size_t w, const Scoring& scoring) {
        const size_t lazy_extend = 2 * k;
        // -------------- START --------------
        {
            const size_t r = first.first, c = first.second;
            if (!(r == 0 && c == 0)) {
                int32_t score_to_start = r > 0 ? scoring.xclip_prefix : 0;
                score_to_start += c > 0 ? scoring.yclip_prefix : 0;
                if (score_to_start == 0) {
                    const size_t d = std::min(lazy_extend, std::min(r, c));
                    add_kmer({(uint32_t)(r - d), (uint32_t)(c - d)}, d, w);
                    add_gap({(uint32_t)sat_sub(r, lazy_extend), (uint32_t)sat_sub(c, lazy_extend)},
                            {(uint32_t)(r - d), (uint32_t)(c - d)}, w);
                } else {
                    const int32_t diagonal_score = r > c ? scoring.xclip_prefix : (r < c ? scoring.yclip_prefix : 0);
                    if (diagonal_score == 0) {
                        const size_t d = std::min(r, c);
                        add_kmer({(uint32_t)(r - d), (uint32_t)(c - d)}, d, w);
                        const Match s{(uint32_t)sat_sub(r, lazy_extend), (uint32_t)sat_sub(c, lazy_extend)};
                        const Match e{(uint32_t)(r - d), (uint32_t)(c - d)};
                        if (s.first <= e.first && s.second <= e.second) add_gap(s, e, w);
                    } else {
                        add_gap({0u, 0u}, first, w);
                    }
                }
            }
        }
        // -------------- END --------------
        {
            const size_t r = last.first + k, c = last.second + k;
            if (!(r == rows && c == cols)) {
                int32_t score_from_end = r == rows ? 0 : scoring.xclip_suffix;
                score_from_end += c == cols ? 0 : scoring.yclip_suffix;
                if (score_from_end == 0) {
                    const size_t d = std::min(lazy_extend, std::min(rows - r, cols - c));
                    add_kmer({(uint32_t)r, (uint32_t)c}, d, w);
                    const size_t r1 = std::min(rows, r + d) - 1, c1 = std::min(cols, c + d) - 1;
                    const size_t r2 = std::min(rows, r + lazy_extend), c2 = std::min(cols, c + lazy_extend);
                    if (r1 <= r2 && c1 <= c2) add_gap({(uint32_t)r1, (uint32_t)c1}, {(uint32_t)r2, (uint32_t)c2}, w);
                } else {
                    const size_t dr = rows - r, dc = cols - c;
                    const int32_t diagonal_score = dr > dc ? scoring.xclip_suffix : (dr < dc ? scoring.yclip_suffix : 0);
                    if (diagonal_score == 0) {
                        const size_t d = std::min(dr, dc);
                        add_kmer({(uint32_t)r, (uint32_t)c}, d, w);
                        const size_t r1 = std::min(rows, r + d) - 1, c1 = std::min(cols, c + d) - 1;
                        const size_t r2 = std::min(rows, r + lazy_extend), c2 = std::min(cols, c + lazy_extend);
                        if (r1 <= r2 && c1 <= c2) add_gap({(uint32_t)r1, (uint32_t)c1}, {(uint32_t)r2, (uint32_t)c2}, w);
                    } else {
                        add_gap({(uint32_t)r, (uint32_t)c}, {(uint32_t)rows, (uint32_t)cols}, w);
                    }
                }
            }
        }
    }

    void full_matrix() {  // banded.rs:1369-1372
        start.assign(cols, 0);
        end.assign(cols, rows);
    }
    size_t num_cells() const {  // banded.rs:1374-1380
        size_t cells = 0;
        for (size_t j = 0; j < start.size(); j++) cells += sat_sub(end[j], start[j]);
        return cells;
    }

    // banded.rs:1330-1367
    static Band create_from_match_path(size_t m, size_t n, size_t k, size_t w, const Scoring& scoring,
                                       const std::vector<size_t>& path, const std::vector<Match>& matches) {
        Band band(m, n);
        if (matches.empty()) {
            band.full_matrix();
            return band;
        }
        const size_t ps = path[0], pe = path[path.size() - 1];
        band.set_boundaries(matches[ps], matches[pe], k, w, scoring);
        bool have_prev = false;
        Match prev{0, 0};
        for (size_t idx : path) {
            const Match curr = matches[idx];
            const bool continues = have_prev && curr.first == prev.first + 1 && curr.second == prev.second + 1;
            if (continues) {
                band.add_entry({prev.first + (uint32_t)k, prev.second + (uint32_t)k}, w);
            } else {
                if (have_prev)
                    band.add_gap({prev.first + (uint32_t)(k - 1), prev.second + (uint32_t)(k - 1)}, curr, w);
                band.add_kmer(curr, k, w);
            }
            prev = curr;
            have_prev = true;
        }
        return band;
    }

    // banded.rs:1301-1328
    static Band create_with_matches(size_t m, size_t n, size_t k, size_t w, const Scoring& scoring,
                                    const std::vector<Match>& matches) {
        if (matches.empty()) {
            Band band(m, n);
            band.full_matrix();
            return band;
        }
        const int32_t match_score = scoring.match_scores_some ? scoring.match_score : DEFAULT_MATCH_SCORE;
        const SparseResult res = sdpkpp(matches, k, (uint32_t)match_score, scoring.gap_open, scoring.gap_extend);
        return create_from_match_path(m, n, k, w, scoring, res.path, matches);
    }

    // banded.rs:1278-1287
    static Band create(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w,
                       const Scoring& scoring) {
        return create_with_matches(m, n, k, w, scoring, find_kmer_matches(x, m, y, n, k));
    }
};

// banded.rs:122-135
struct BandedAligner {
    std::vector<int32_t> S[2], I[2], D[2];
    std::vector<size_t> Lx, Ly;
    std::vector<int32_t> Sn;
    Traceback traceback;
    Scoring scoring;
    Band band;
    size_t k, w;
    BandedAligner(const Scoring& s, size_t k_, size_t w_) : scoring(s), k(k_), w(w_) {}
    // how the band of the next run() is made (the custom_with_* entry points)
    int variant = 0;
    std::vector<Match> v_matches;
    std::vector<size_t> v_path;
    int v_allowed_mismatches = -1;
    bool v_use_lcskpp_union = false;

    // banded.rs:406-869
    Alignment compute_alignment(const uint8_t* x, size_t m, const uint8_t* y, size_t n) {
        if (band.num_cells() > MAX_CELLS) {  // 407-420
            Alignment a;
            a.score = MIN_SCORE;
            a.mode = ORC_MODE_CUSTOM;
            return a;
        }
        const Scoring& sc = scoring;
        const std::vector<size_t>&bs = band.start, &be = band.end;
        traceback.init(m, n);  // 423
        for (int k2 = 0; k2 < 2; k2++) {
            I[k2].assign(m + 1, MIN_SCORE);
            D[k2].assign(m + 1, MIN_SCORE);
            S[k2].assign(m + 1, MIN_SCORE);
        }
        Lx.assign(n + 1, 0);
        Ly.assign(m + 1, 0);
        Sn.assign(m + 1, MIN_SCORE);

        {  // j = 0, banded.rs:440-509
            const size_t curr = 0;
            const size_t i_start = bs[0], i_end = be[0];
            if (i_start == 0) S[curr][0] = 0;
            for (size_t i = std::max<size_t>(1, i_start); i < i_end; i++) {
                TracebackCell tb;
                tb.set_all(TB_START);
                if (i == 1) {
                    I[curr][i] = sc.gap_open;
                    tb.set_i_bits(TB_START);
                } else {
                    int32_t i_score = sc.gap_open + sc.gap_extend * ((int32_t)i - 1);
                    int32_t c_score = sc.xclip_prefix + sc.gap_open;
                    if (i_score > c_score) {
                        I[curr][i] = i_score;
                        tb.set_i_bits(TB_INS);
                    } else {
                        I[curr][i] = c_score;
                        tb.set_i_bits(TB_XCLIP_PREFIX);
                    }
                }
                if (i == m) tb.set_s_bits(TB_XCLIP_SUFFIX);
                if (I[curr][i] > S[curr][i]) {
                    S[curr][i] = I[curr][i];
                    tb.set_s_bits(TB_INS);
                }
                if (sc.xclip_prefix > S[curr][i]) {
                    S[curr][i] = sc.xclip_prefix;
                    tb.set_s_bits(TB_XCLIP_PREFIX);
                }
                if (S[curr][i] + sc.xclip_suffix > S[curr][m]) {
                    S[curr][m] = S[curr][i] + sc.xclip_suffix;
                    Lx[0] = m - i;
                    traceback.get_mut(m, 0).set_s_bits(TB_XCLIP_SUFFIX);
                }
                traceback.set(i, 0, tb);
            }
            for (size_t i = i_end; i < std::min(m + 1, be[std::min<size_t>(n, 1)]); i++) {
                S[curr][i] = MIN_SCORE;
                I[curr][i] = MIN_SCORE;
            }
            if (i_end < m + 1) S[curr][m] = MIN_SCORE;
            if (sc.yclip_prefix > sc.yclip_suffix) {
                Sn[0] = sc.yclip_prefix;
                traceback.get_mut(0, n).set_s_bits(TB_YCLIP_PREFIX);
            } else {
                Sn[0] = sc.yclip_suffix;
                Ly[0] = n;
                traceback.get_mut(0, n).set_s_bits(TB_YCLIP_SUFFIX);
            }
        }

        for (size_t j = 1; j <= n; j++) {  // banded.rs:511-681
            const size_t curr = j % 2, prev = 1 - curr;
            const size_t i_start = bs[j], i_end = be[j];
            if (i_start == 0) {
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
                if (S[curr][0] + sc.yclip_suffix > Sn[0]) {
                    Sn[0] = S[curr][0] + sc.yclip_suffix;
                    Ly[0] = n - j;
                    traceback.get_mut(0, n).set_s_bits(TB_YCLIP_SUFFIX);
                }
                traceback.set(0, j, tb);
            }
            for (size_t i = sat_sub(i_start, 1); i < i_start; i++) {  // 556-560 (i_start may be m+1)
                if (i > m) throw OracleError("index out of bounds");
                S[curr][i] = MIN_SCORE;
                I[curr][i] = MIN_SCORE;
                D[curr][i] = MIN_SCORE;
            }
            S[curr][m] = MIN_SCORE;

            const uint8_t q = y[j - 1];
            const int32_t xclip_score =
                sc.xclip_prefix + std::max(j == n ? std::max(sc.yclip_prefix, Sn[0]) : sc.yclip_prefix,
                                           sc.gap_open + sc.gap_extend * ((int32_t)j - 1));

            for (size_t i = std::max<size_t>(1, i_start); i < i_end; i++) {
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
                if (j == n) {
                    int32_t clip_score = Sn[i - 1] + sc.gap_open;
                    if (clip_score > best_i_score) {
                        best_i_score = clip_score;
                        tb.set_i_bits(TB_YCLIP_SUFFIX);
                    }
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

                if (i == m)
                    tb.set_s_bits(TB_XCLIP_SUFFIX);
                else
                    S[curr][i] = MIN_SCORE;
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
                int32_t yclip_score = sc.yclip_prefix + sc.gap_open + sc.gap_extend * ((int32_t)i - 1);
                if (yclip_score > best_s_score) {
                    best_s_score = yclip_score;
                    tb.set_s_bits(TB_YCLIP_PREFIX);
                }

                S[curr][i] = best_s_score;
                I[curr][i] = best_i_score;
                D[curr][i] = best_d_score;

                if (S[curr][i] + sc.xclip_suffix > S[curr][m]) {
                    S[curr][m] = S[curr][i] + sc.xclip_suffix;
                    Lx[j] = m - i;
                    traceback.get_mut(m, j).set_s_bits(TB_XCLIP_SUFFIX);
                }
                if (S[curr][i] + sc.yclip_suffix > Sn[i]) {
                    Sn[i] = S[curr][i] + sc.yclip_suffix;
                    Ly[i] = n - j;
                    traceback.get_mut(i, n).set_s_bits(TB_YCLIP_SUFFIX);
                }
                traceback.set(i, j, tb);
            }

            if (S[curr][m] + sc.yclip_suffix > Sn[m]) {  // 665-670
                Sn[m] = S[curr][m] + sc.yclip_suffix;
                Ly[m] = n - j;
                traceback.get_mut(m, n).set_s_bits(TB_YCLIP_SUFFIX);
            }
            if (i_end < m + 1) {
                traceback.get_mut(m, j).set_s_bits(TB_XCLIP_SUFFIX);
                S[curr][m] = MIN_SCORE;
            }
            for (size_t i = i_end; i < std::min(m + 1, be[std::min(n, j + 1)]); i++) {
                S[curr][i] = MIN_SCORE;
                I[curr][i] = MIN_SCORE;
                D[curr][i] = MIN_SCORE;
            }
        }

        for (size_t i = 0; i <= m; i++) {  // banded.rs:684-701
            const size_t j = n, curr = j % 2;
            if (i != m && (i < bs[j] || i > be[j])) S[curr][i] = MIN_SCORE;
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

        for (size_t i = std::max<size_t>(1, bs[n]); i < be[n]; i++) {  // banded.rs:705-723
            const size_t j = n, curr = j % 2;
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

        for (size_t j = 1; j <= n; j++) {  // banded.rs:725-744
            int32_t d_score = sc.gap_open + sc.gap_extend * ((int32_t)j - 1);
            if (d_score > sc.yclip_prefix)
                traceback.get_mut(0, j).set_s_bits(TB_DEL);
            else
                traceback.get_mut(0, j).set_s_bits(TB_YCLIP_PREFIX);
            if (j == n) {
                int32_t best_score = std::max(d_score, sc.yclip_prefix);
                if (sc.yclip_suffix > best_score) {
                    best_score = sc.yclip_suffix;
                    traceback.get_mut(0, j).set_s_bits(TB_YCLIP_SUFFIX);
                }
                if (sc.xclip_suffix + best_score > S[n % 2][m]) {
                    S[n % 2][m] = sc.xclip_suffix + best_score;
                    Lx[n] = m;
                    traceback.get_mut(m, n).set_s_bits(TB_XCLIP_SUFFIX);
                }
            }
        }

        for (size_t i = 1; i <= m; i++) {  // banded.rs:746-765
            int32_t c_score = sc.gap_open + sc.gap_extend * ((int32_t)i - 1);
            if (c_score > sc.xclip_prefix)
                traceback.get_mut(i, 0).set_s_bits(TB_INS);
            else
                traceback.get_mut(i, 0).set_s_bits(TB_XCLIP_PREFIX);
            if (i == m) {
                int32_t best_score = std::max(c_score, sc.xclip_prefix);
                if (sc.xclip_suffix > best_score) {
                    best_score = sc.xclip_suffix;
                    traceback.get_mut(i, 0).set_s_bits(TB_XCLIP_SUFFIX);
                }
                if (sc.yclip_suffix + best_score > S[n % 2][m]) {
                    S[n % 2][m] = sc.yclip_suffix + best_score;
                    Ly[m] = n;
                    traceback.get_mut(m, n).set_s_bits(TB_YCLIP_SUFFIX);
                }
            }
        }

        // banded.rs:767-868 — traceback
        size_t i = m, j = n;
        std::vector<Op> operations;
        operations.reserve(m);
        size_t xstart = 0, ystart = 0, xend = m, yend = n;
        uint16_t last_layer = traceback.get(i, j).get_s_bits();
        const size_t guard = 4 * (m + n) + 64;
        for (size_t steps = 0;; steps++) {
            if (steps > guard) throw OracleError("traceback does not terminate");
            uint16_t next_layer;
            if (last_layer == TB_START) break;
            switch (last_layer) {
                case TB_INS:
                    operations.push_back({ORC_OP_INS, 0});
                    next_layer = traceback.get(i, j).get_i_bits();
                    if (i == 0) throw OracleError("attempt to subtract with overflow");
                    i -= 1;
                    break;
                case TB_DEL:
                    operations.push_back({ORC_OP_DEL, 0});
                    next_layer = traceback.get(i, j).get_d_bits();
                    if (j == 0) throw OracleError("attempt to subtract with overflow");
                    j -= 1;
                    break;
                case TB_MATCH:
                case TB_SUBST:
                    operations.push_back({(uint8_t)(last_layer == TB_MATCH ? ORC_OP_MATCH : ORC_OP_SUBST), 0});
                    if (i == 0 || j == 0) throw OracleError("attempt to subtract with overflow");
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
                    operations.push_back({ORC_OP_XCLIP, Lx[j]});
                    if (Lx[j] > i) throw OracleError("attempt to subtract with overflow");
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
                    if (Ly[i] > j) throw OracleError("attempt to subtract with overflow");
                    j -= Ly[i];
                    yend = j;
                    next_layer = traceback.get(i, j).get_s_bits();
                    break;
                default:
                    throw OracleError("Dint expect this!");
            }
            last_layer = next_layer;
        }
        // 833-855: the traceback ended outside the band other than at (0, 0)
        if (i != 0) {
            int32_t i_score = sc.gap_open + sc.gap_extend * ((int32_t)i - 1);
            if (i_score > sc.xclip_prefix) {
                operations.resize(operations.size() + i, Op{ORC_OP_INS, 0});
                xstart = 0;
            } else {
                operations.push_back({ORC_OP_XCLIP, i});
                xstart = i;
            }
        }
        if (j != 0) {
            int32_t d_score = sc.gap_open + sc.gap_extend * ((int32_t)j - 1);
            if (d_score > sc.yclip_prefix) {
                operations.resize(operations.size() + j, Op{ORC_OP_DEL, 0});
                ystart = 0;
            } else {
                operations.push_back({ORC_OP_YCLIP, j});
                ystart = j;
            }
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

    // banded.rs:282-285, 872-1004: the wrappers set the clips BEFORE the band is created
    Alignment run(int mode, const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t* cells) {
        int32_t saved[4] = {scoring.xclip_prefix, scoring.xclip_suffix, scoring.yclip_prefix, scoring.yclip_suffix};
        bool filter = false;
        if (mode == ORC_MODE_GLOBAL) {  // 872-899
            scoring.xclip_prefix = scoring.xclip_suffix = scoring.yclip_prefix = scoring.yclip_suffix = MIN_SCORE;
        } else if (mode == ORC_MODE_SEMIGLOBAL) {  // 901-931
            scoring.xclip_prefix = scoring.xclip_suffix = MIN_SCORE;
            scoring.yclip_prefix = scoring.yclip_suffix = 0;
            filter = true;
        } else if (mode == ORC_MODE_LOCAL) {  // 972-1003
            scoring.xclip_prefix = scoring.xclip_suffix = scoring.yclip_prefix = scoring.yclip_suffix = 0;
            filter = true;
        }
        // variant: 0 create (282-285), 1 custom_with_matches (313-321), 2 custom_with_match_path (391-401),
        //          3 custom_with_expanded_matches (338-389)
        if (variant == 0) {
            band = Band::create(x, m, y, n, k, w, scoring);
        } else if (variant == 1) {
            band = Band::create_with_matches(m, n, k, w, scoring, v_matches);
        } else if (variant == 2) {
            band = Band::create_from_match_path(m, n, k, w, scoring, v_path, v_matches);
        } else {
            const std::vector<Match> expanded =
                v_allowed_mismatches >= 0 ? expand_kmer_matches(x, m, y, n, k, v_matches, (size_t)v_allowed_mismatches) : v_matches;
            if (v_use_lcskpp_union) {
                const int32_t match_score = scoring.match_scores_some ? scoring.match_score : DEFAULT_MATCH_SCORE;
                const std::vector<size_t> path =
                    sdpkpp_union_lcskpp_path(expanded, k, (uint32_t)match_score, scoring.gap_open, scoring.gap_extend);
                band = Band::create_from_match_path(m, n, k, w, scoring, path, expanded);
            } else {
                band = Band::create_with_matches(m, n, k, w, scoring, expanded);
            }
        }
        if (cells) *cells = band.num_cells();
        Alignment a = compute_alignment(x, m, y, n);
        if (mode != ORC_MODE_CUSTOM) a.mode = mode;
        if (filter) a.filter_clip_operations();
        scoring.xclip_prefix = saved[0];
        scoring.xclip_suffix = saved[1];
        scoring.yclip_prefix = saved[2];
        scoring.yclip_suffix = saved[3];
        return a;
    }
};

}  // namespace orc

extern "C" int orc_banded_align(const orc_scoring_t* sc, int mode, uint32_t k, uint32_t w, const uint8_t* x,
                                uint64_t m, const uint8_t* y, uint64_t n, orc_alignment_t* out, uint64_t* ops,
                                uint64_t ops_cap, uint64_t* band_cells) {
    try {
        orc::BandedAligner al(orc::scoring_from_c(sc), k, w);
        size_t cells = 0;
        orc::Alignment a = al.run(mode, x, m, y, n, &cells);
        if (band_cells) *band_cells = cells;
        return orc::export_alignment(a, out, ops, ops_cap);
    } catch (const std::exception&) {
        return -2;
    }
}

extern "C" int orc_banded_align_batch(const orc_scoring_t* sc, int mode, uint32_t k, uint32_t w, uint64_t n_pairs,
                                      const uint8_t* x, const uint64_t* x_off, const uint8_t* y,
                                      const uint64_t* y_off, orc_alignment_t* out, uint64_t* ops,
                                      uint64_t ops_stride, uint64_t* band_cells, int threads) {
    if (threads < 1) threads = 1;
    std::vector<int> rc(threads, 0);
    auto work = [&](int t) {
        orc::BandedAligner al(orc::scoring_from_c(sc), k, w);
        for (uint64_t p = t; p < n_pairs; p += threads) {
            try {
                size_t cells = 0;
                orc::Alignment a = al.run(mode, x + x_off[p], x_off[p + 1] - x_off[p], y + y_off[p],
                                          y_off[p + 1] - y_off[p], &cells);
                if (band_cells) band_cells[p] = cells;
                int r = orc::export_alignment(a, &out[p], ops ? ops + p * ops_stride : nullptr, ops ? ops_stride : 0);
                if (r && ops) rc[t] = r;
            } catch (const std::exception&) {
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

extern "C" uint64_t orc_band_create(const orc_scoring_t* sc, uint32_t k, uint32_t w, const uint8_t* x, uint64_t m,
                                    const uint8_t* y, uint64_t n, uint32_t* start, uint32_t* end) {
    orc::Band b = orc::Band::create(x, m, y, n, k, w, orc::scoring_from_c(sc));
    for (uint64_t j = 0; j <= n; j++) {
        start[j] = (uint32_t)b.start[j];
        end[j] = (uint32_t)b.end[j];
    }
    return b.num_cells();
}

// Band primitives for the geometry KATs (banded.rs:1469-1618): ops = {kind, r, c, k, w} with
// kind 0 add_entry, 1 add_kmer, 2 add_gap (r,c -> k,w as end point, w in op[5])
extern "C" void orc_band_apply(uint64_t m, uint64_t n, const uint32_t* ops, uint64_t n_ops, uint32_t* start,
                               uint32_t* end) {
    orc::Band b(m, n);
    for (uint64_t t = 0; t < n_ops; t++) {
        const uint32_t* o = ops + 6 * t;
        if (o[0] == 0) b.add_entry({o[1], o[2]}, o[4]);
        if (o[0] == 1) b.add_kmer({o[1], o[2]}, o[3], o[4]);
        if (o[0] == 2) b.add_gap({o[1], o[2]}, {o[3], o[4]}, o[5]);
    }
    for (uint64_t j = 0; j <= n; j++) {
        start[j] = (uint32_t)b.start[j];
        end[j] = (uint32_t)b.end[j];
    }
}

// custom_with_matches / custom_with_match_path / custom_with_expanded_matches (banded.rs:313-401), any mode.
// Optionally exports the band (n + 1 half-open row ranges).
extern "C" int orc_banded_align_with(const orc_scoring_t* sc, int mode, uint32_t k, uint32_t w, const uint8_t* x,
                                     uint64_t m, const uint8_t* y, uint64_t n, int variant, const uint32_t* matches_xy,
                                     uint64_t n_matches, const uint32_t* path, uint64_t n_path, int allowed_mismatches,
                                     int use_lcskpp_union, orc_alignment_t* out, uint64_t* ops, uint64_t ops_cap,
                                     uint64_t* band_cells, uint32_t* band_start, uint32_t* band_end) {
    try {
        orc::BandedAligner al(orc::scoring_from_c(sc), k, w);
        al.variant = variant;
        for (uint64_t i = 0; i < n_matches; i++) al.v_matches.emplace_back(matches_xy[2 * i], matches_xy[2 * i + 1]);
        for (uint64_t i = 0; i < n_path; i++) al.v_path.push_back(path[i]);
        al.v_allowed_mismatches = allowed_mismatches;
        al.v_use_lcskpp_union = use_lcskpp_union != 0;
        size_t cells = 0;
        orc::Alignment a = al.run(mode, x, m, y, n, &cells);
        if (band_cells) *band_cells = cells;
        if (band_start && band_end)
            for (uint64_t j = 0; j <= n; j++) {
                band_start[j] = (uint32_t)al.band.start[j];
                band_end[j] = (uint32_t)al.band.end[j];
            }
        return orc::export_alignment(a, out, ops, ops_cap);
    } catch (const std::exception&) {
        return -2;
    }
}
