// Host-side band construction of the banded aligner (product code, runs on host threads inside
// bg_align_banded_batch / bg_band_create_batch).  It stands where rust-bio's own host code runs
//   sparse::find_kmer_matches   /root/reference/src/alignment/sparse.rs:337-402
//   sparse::sdpkpp              /root/reference/src/alignment/sparse.rs:188-295 (+ PrevPtr 145-167,
//                               MaxBitTree /root/reference/src/data_structures/bit_tree.rs:45-101)
//   Band::create*               /root/reference/src/alignment/pairwise/banded.rs:1047-1380
// and must produce identical per-column row ranges (the chain's tie-breaking included), because
// the banded DP result depends on the exact band.  k-mer matching is a sort-merge join over
// 64-bit rolling hashes (verified with memcmp) instead of a hash map; the sparse DP keeps the
// reference's event order and candidate ordering; band rasterisation is the same integer maths.
#include "band_host.h"

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cstring>

namespace bgband {

namespace {

// Rolling polynomial hash (mod 2^64) of every k-mer of s: h_i = sum_t s[i+t] * B^(k-1-t)
constexpr uint64_t kHashBase = 0x9E3779B97F4A7C15ull | 1ull;

void kmer_hashes(const uint8_t* s, size_t n, size_t k, std::vector<uint64_t>& out) {
    out.clear();
    if (n < k) return;
    out.resize(n - k + 1);
    uint64_t top = 1;  // B^(k-1)
    for (size_t t = 1; t < k; t++) top *= kHashBase;
    uint64_t h = 0;
    for (size_t t = 0; t < k; t++) h = h * kHashBase + (uint64_t)s[t] + 1;
    out[0] = h;
    for (size_t i = 1; i + k <= n; i++) {
        h = (h - ((uint64_t)s[i - 1] + 1) * top) * kHashBase + (uint64_t)s[i + k - 1] + 1;
        out[i] = h;
    }
}

// the reference's derived Ord on PrevPtr: (plane, score, d, id, x, y).  x and y are functions of id
// (the end point of k-mer id), so the order is decided by the first four fields: two 64-bit keys.
struct Frag {
    uint64_t hi = 0;  // plane << 32 | score
    uint64_t lo = 0;  // d << 32 | id
    uint32_t score() const { return (uint32_t)hi; }
    uint32_t id() const { return (uint32_t)lo; }
};
inline bool frag_less(const Frag& a, const Frag& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

struct Event {
    uint32_t x, y, tag;
    bool operator<(const Event& o) const {
        if (x != o.x) return x < o.x;
        if (y != o.y) return y < o.y;
        return tag < o.tag;
    }
};

inline size_t ssub(size_t a, size_t b) { return a > b ? a - b : 0; }

}  // namespace

// All exact k-mer matches, sorted by (x, y) (sparse.rs:337-402: the reference hashes the shorter
// sequence and probes with the other one; the hash-map iteration order is erased by its final sort).
// Here: rolling hashes, a chained hash table over the indexed sequence, memcmp to confirm.
void find_kmer_matches(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, std::vector<Match>& out) {
    out.clear();
    const bool hash_x = m < n;  // sparse.rs:338-344: the shorter sequence is indexed, y on ties
    const uint8_t* hs = hash_x ? x : y;
    const size_t hn = hash_x ? m : n;
    const uint8_t* os = hash_x ? y : x;
    const size_t on = hash_x ? n : m;
    if (hn < k || on < k) return;
    thread_local std::vector<uint64_t> hi, hp;
    thread_local std::vector<uint32_t> head, next;
    kmer_hashes(hs, hn, k, hi);
    kmer_hashes(os, on, k, hp);
    const size_t ni = hi.size();
    unsigned bits = 4;
    while (((size_t)1 << bits) < 2 * ni) bits++;
    head.assign((size_t)1 << bits, 0xFFFFFFFFu);
    next.resize(ni);
    auto bucket = [bits](uint64_t h) { return (size_t)((h * 0xD6E8FEB86659FD93ull) >> (64 - bits)); };
    for (size_t i = ni; i-- > 0;) {  // descending, so that chains list positions in ascending order
        const size_t bkt = bucket(hi[i]);
        next[i] = head[bkt];
        head[bkt] = (uint32_t)i;
    }
    for (size_t p = 0; p < hp.size(); p++) {
        const uint64_t h = hp[p];
        for (uint32_t i = head[bucket(h)]; i != 0xFFFFFFFFu; i = next[i]) {
            if (hi[i] != h || (k && memcmp(hs + i, os + p, k) != 0)) continue;
            if (hash_x)
                out.push_back({i, (uint32_t)p});
            else
                out.push_back({(uint32_t)p, i});
        }
    }
    std::sort(out.begin(), out.end());
}

bool sdpkpp_path(const std::vector<Match>& matches, size_t k_, uint32_t match_score, int32_t gap_open,
                 int32_t gap_extend, std::vector<uint32_t>& path) {
    path.clear();
    const uint32_t nm = (uint32_t)matches.size();
    if (nm == 0) return true;
    if (gap_open > 0 || gap_extend > 0) return false;
    const uint32_t k = (uint32_t)k_;
    const uint32_t go = (uint32_t)(-(int64_t)gap_open), ge = (uint32_t)(-(int64_t)gap_extend);
    thread_local std::vector<Event> ev;
    thread_local std::vector<Frag> tree;
    thread_local std::vector<uint32_t> score;
    thread_local std::vector<int32_t> back;
    ev.clear();
    ev.reserve(2 * (size_t)nm);
    uint32_t span = 0;
    for (uint32_t i = 0; i < nm; i++) {
        ev.push_back({matches[i].x, matches[i].y, i + nm});   // start of k-mer i
        ev.push_back({matches[i].x + k, matches[i].y + k, i});  // end of k-mer i
        span = std::max(span, std::max(matches[i].x + k, matches[i].y + k));
    }
    std::sort(ev.begin(), ev.end());
    // prefix-max Fenwick tree over y of the best fragment ending at or before y
    tree.assign((size_t)span + 1, Frag());
    score.assign(nm, 0);
    back.assign(nm, 0);
    uint32_t best_score = k;  // (k, 0): sparse.rs:234
    int32_t best_idx = 0;
    auto better = [](uint32_t s1, int32_t i1, uint32_t s2, int32_t i2) { return s1 > s2 || (s1 == s2 && i1 > i2); };
    for (const Event& e : ev) {
        const uint32_t p = e.tag % nm;
        if (e.tag >= nm) {  // start event
            score[p] = k * match_score;
            back[p] = -1;
            Frag bp;
            for (size_t i = (size_t)e.y + 1; i > 0; i -= i & (~i + 1))
                if (frag_less(bp, tree[i])) bp = tree[i];
            if (bp.score() > 0) {
                const uint32_t bx = matches[bp.id()].x + k, by = matches[bp.id()].y + k;
                const uint32_t gap = std::max(e.x - bx, e.y - by);
                const uint32_t pen = gap > 0 ? go + gap * ge : 0;
                const uint32_t sum = bp.score() + k * match_score;
                const uint32_t ns = sum > pen ? sum - pen : 0;
                if (better(ns, (int32_t)bp.id(), score[p], back[p])) {
                    score[p] = ns;
                    back[p] = (int32_t)bp.id();
                }
                if (better(score[p], (int32_t)p, best_score, best_idx)) {
                    best_score = score[p];
                    best_idx = (int32_t)p;
                }
            }
        } else {  // end event
            if (e.x > k && e.y > k) {
                const Match key{e.x - k - 1, e.y - k - 1};
                auto it = std::lower_bound(matches.begin(), matches.end(), key);
                if (it != matches.end() && it->x == key.x && it->y == key.y) {
                    const int32_t c = (int32_t)(it - matches.begin());
                    const uint32_t cs = score[c] + match_score;
                    if (better(cs, c, score[p], back[p])) {
                        score[p] = cs;
                        back[p] = c;
                    }
                    if (better(score[p], (int32_t)p, best_score, best_idx)) {
                        best_score = score[p];
                        best_idx = (int32_t)p;
                    }
                }
            }
            Frag f;
            const uint32_t d = e.x + e.y;
            f.hi = (uint64_t)(uint32_t)(score[p] + d * ge) << 32 | score[p];
            f.lo = (uint64_t)d << 32 | p;
            for (size_t i = (size_t)e.y + 1; i < tree.size(); i += i & (~i + 1))
                if (frag_less(tree[i], f)) tree[i] = f;
        }
    }
    for (int32_t q = best_idx; q >= 0; q = back[q]) path.push_back((uint32_t)q);
    std::reverse(path.begin(), path.end());
    return true;
}

// lcskpp (sparse.rs:67-143): best chain of k-mer matches by covered bases.  dp values are the tuples
// (score, predecessor index); the Fenwick tree holds (score, match index) with the tuple order.
bool lcskpp_path(const std::vector<Match>& matches, size_t k_, std::vector<uint32_t>& path, uint32_t* score_out) {
    path.clear();
    if (score_out) *score_out = 0;
    const uint32_t nm = (uint32_t)matches.size();
    if (nm == 0) return true;
    for (uint32_t i = 1; i < nm; i++)
        if (!(matches[i - 1] < matches[i])) return false;
    const uint32_t k = (uint32_t)k_;
    std::vector<Event> ev;
    ev.reserve(2 * (size_t)nm);
    uint32_t span = 0;
    for (uint32_t i = 0; i < nm; i++) {
        ev.push_back({matches[i].x, matches[i].y, i + nm});
        ev.push_back({matches[i].x + k, matches[i].y + k, i});
        span = std::max(span, std::max(matches[i].x + k, matches[i].y + k));
    }
    std::sort(ev.begin(), ev.end());
    std::vector<uint64_t> tree((size_t)span + 1, 0);  // score << 32 | index
    std::vector<uint32_t> score(nm, 0);
    std::vector<int32_t> back(nm, 0);
    auto better = [](uint32_t s1, int32_t i1, uint32_t s2, int32_t i2) { return s1 > s2 || (s1 == s2 && i1 > i2); };
    uint32_t best_score = k;
    int32_t best_idx = 0;
    for (const Event& e : ev) {
        const uint32_t p = e.tag % nm;
        if (e.tag >= nm) {
            score[p] = k;
            back[p] = -1;
            uint64_t bp = 0;
            for (size_t i = (size_t)e.y + 1; i > 0; i -= i & (~i + 1)) bp = std::max(bp, tree[i]);
            if ((uint32_t)(bp >> 32) > 0) {
                score[p] = k + (uint32_t)(bp >> 32);
                back[p] = (int32_t)(uint32_t)bp;
                if (better(score[p], (int32_t)p, best_score, best_idx)) {
                    best_score = score[p];
                    best_idx = (int32_t)p;
                }
            }
        } else {
            if (e.x > k && e.y > k) {
                const Match key{e.x - k - 1, e.y - k - 1};
                auto it = std::lower_bound(matches.begin(), matches.end(), key);
                if (it != matches.end() && it->x == key.x && it->y == key.y) {
                    const int32_t c = (int32_t)(it - matches.begin());
                    if (better(score[c] + 1, c, score[p], back[p])) {
                        score[p] = score[c] + 1;
                        back[p] = c;
                    }
                    if (better(score[p], (int32_t)p, best_score, best_idx)) {
                        best_score = score[p];
                        best_idx = (int32_t)p;
                    }
                }
            }
            const uint64_t f = (uint64_t)score[p] << 32 | p;
            for (size_t i = (size_t)e.y + 1; i < tree.size(); i += i & (~i + 1)) tree[i] = std::max(tree[i], f);
        }
    }
    for (int32_t q = best_idx; q >= 0; q = back[q]) path.push_back((uint32_t)q);
    std::reverse(path.begin(), path.end());
    if (score_out) *score_out = best_score;
    return true;
}

// sdpkpp_union_lcskpp_path (sparse.rs:297-329): lcskpp prefix + the sdpkpp path + lcskpp suffix
bool sdpkpp_union_lcskpp_path(const std::vector<Match>& matches, size_t k, uint32_t match_score, int32_t gap_open,
                              int32_t gap_extend, std::vector<uint32_t>& out) {
    out.clear();
    if (matches.empty()) return true;
    std::vector<uint32_t> lcs, sdp;
    if (!lcskpp_path(matches, k, lcs, nullptr) || !sdpkpp_path(matches, k, match_score, gap_open, gap_extend, sdp)) return false;
    size_t pre = 0, post = lcs.size();
    auto it = std::lower_bound(lcs.begin(), lcs.end(), sdp.front());
    if (it != lcs.end() && *it == sdp.front()) pre = (size_t)(it - lcs.begin());
    it = std::lower_bound(lcs.begin(), lcs.end(), sdp.back());
    if (it != lcs.end() && *it == sdp.back()) post = (size_t)(it - lcs.begin()) + 1;
    out.insert(out.end(), lcs.begin(), lcs.begin() + pre);
    out.insert(out.end(), sdp.begin(), sdp.end());
    out.insert(out.end(), lcs.begin() + post, lcs.end());
    return true;
}

// expand_kmer_matches (sparse.rs:404-500): grow every match along its diagonal, to the left until the
// previous match of the diagonal and then to the right until the next one, tolerating
// `allowed_mismatches` mismatching bases per run.  The reference keeps "last/next match of a diagonal"
// in hash maps; here the matches are bucketed by diagonal, which visits them in the same order.
bool expand_kmer_matches(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, const std::vector<Match>& sorted,
                         size_t allowed_mismatches, std::vector<Match>& out) {
    out.clear();
    for (size_t i = 1; i < sorted.size(); i++)
        if (!(sorted[i - 1] < sorted[i])) return false;
    std::vector<Match> left(sorted);
    {
        // matches of one diagonal appear in increasing x in the sorted order
        std::vector<std::pair<int64_t, uint32_t>> by_diag;  // (diagonal, index)
        for (uint32_t i = 0; i < sorted.size(); i++) by_diag.push_back({(int64_t)sorted[i].x - (int64_t)sorted[i].y, i});
        std::stable_sort(by_diag.begin(), by_diag.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (size_t t = 0; t < by_diag.size(); t++) {
            const Match cur = sorted[by_diag[t].second];
            const bool has_prev = t > 0 && by_diag[t - 1].first == by_diag[t].first;
            const int64_t mn = std::min(cur.x, cur.y);
            // exclusive lower end of the walk: the previous match of the diagonal, or one before the diagonal's start
            const int64_t stop_x = has_prev ? (int64_t)sorted[by_diag[t - 1].second].x : (int64_t)cur.x - mn - 1;
            size_t miss = 0;
            for (int64_t cx = (int64_t)cur.x - 1, cy = (int64_t)cur.y - 1; cx > stop_x; cx--, cy--) {
                miss += x[cx] == y[cy] ? 0 : 1;
                if (miss > allowed_mismatches) break;
                left.push_back({(uint32_t)cx, (uint32_t)cy});
            }
        }
    }
    std::sort(left.begin(), left.end());
    out = left;
    {
        std::vector<std::pair<int64_t, uint32_t>> by_diag;
        for (uint32_t i = 0; i < left.size(); i++) by_diag.push_back({(int64_t)left[i].x - (int64_t)left[i].y, i});
        std::stable_sort(by_diag.begin(), by_diag.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (size_t t = 0; t < by_diag.size(); t++) {
            const Match cur = left[by_diag[t].second];
            const bool has_next = t + 1 < by_diag.size() && by_diag[t + 1].first == by_diag[t].first;
            const uint32_t room = std::min((uint32_t)m - cur.x, (uint32_t)n - cur.y);
            const uint32_t max_inc = room > (uint32_t)k - 1 ? room - ((uint32_t)k - 1) : 0;
            const uint32_t stop_x = has_next ? left[by_diag[t + 1].second].x : cur.x + max_inc;  // exclusive
            size_t miss = 0;
            for (uint32_t cx = cur.x + 1, cy = cur.y + 1; cx < stop_x; cx++, cy++) {
                miss += x[cx + k - 1] == y[cy + k - 1] ? 0 : 1;
                if (miss > allowed_mismatches) break;
                out.push_back({cx, cy});
            }
        }
    }
    std::sort(out.begin(), out.end());
    return true;
}

// ---------------------------------------------------------------------------------- Band
void Band::reset(size_t m, size_t n) {
    rows = m + 1;
    cols = n + 1;
    start.assign(cols, (uint32_t)(m + 1));  // empty range m+1..0 (banded.rs:1061-1067)
    end.assign(cols, 0);
}

void Band::add_entry(uint32_t r_, uint32_t c_, size_t w) {  // banded.rs:1111-1120
    const size_t r = r_, c = c_;
    const uint32_t lo = (uint32_t)ssub(r, w), hi = (uint32_t)std::min(r + w + 1, rows);
    for (size_t j = ssub(c, w), je = std::min(c + w + 1, cols); j < je; j++) {
        start[j] = std::min(start[j], lo);
        end[j] = std::max(end[j], hi);
    }
}

void Band::add_kmer(uint32_t r_, uint32_t c_, size_t k, size_t w) {  // banded.rs:1071-1107
    if (k == 0) return;
    const size_t r = r_, c = c_;
    size_t i = ssub(r, w);
    for (size_t j = ssub(c, w), je = std::min(c + w + 1, cols); j < je; j++) start[j] = std::min(start[j], (uint32_t)i);
    for (size_t j = std::min(c + w, cols), je = std::min(c + k + w, cols); j < je; j++, i++)
        start[j] = std::min(start[j], (uint32_t)i);
    i = r + w + k;
    for (size_t j = ssub(c + k - 1, w); j > ssub(c, w);) {
        j--;
        i--;
        end[j] = std::max(end[j], (uint32_t)std::min(i, rows));
    }
    const uint32_t e = (uint32_t)std::min(r + w + k, rows);
    for (size_t j = ssub(c + k - 1, w), je = std::min(c + k + w, cols); j < je; j++) end[j] = std::max(end[j], e);
}

void Band::add_gap(uint32_t r0, uint32_t c0, uint32_t r1, uint32_t c1, size_t w) {  // banded.rs:1123-1137
    const uint32_t nr = r1 - r0, nc = c1 - c0;  // u32 arithmetic wraps like the reference's release build
    if (nr > nc) {
        for (uint32_t r = r0; r < r1; r++) add_entry(r, c0 + (c1 - c0) * (r - r0) / (r1 - r0), w);
    } else {
        for (uint32_t c = c0; c < c1; c++) add_entry(r0 + (r1 - r0) * (c - c0) / (c1 - c0), c, w);
    }
}

void Band::set_boundaries(Match first, Match last, size_t k, size_t w, const ClipScores& cs) {  // banded.rs:1150-1276
    const size_t lazy = 2 * k;
    {
        const size_t r = first.x, c = first.y;
        if (r != 0 || c != 0) {
            const int32_t to_start = (r > 0 ? cs.xclip_prefix : 0) + (c > 0 ? cs.yclip_prefix : 0);
            if (to_start == 0) {
                const size_t d = std::min(lazy, std::min(r, c));
                add_kmer((uint32_t)(r - d), (uint32_t)(c - d), d, w);
                add_gap((uint32_t)ssub(r, lazy), (uint32_t)ssub(c, lazy), (uint32_t)(r - d), (uint32_t)(c - d), w);
            } else {
                const int32_t diag = r > c ? cs.xclip_prefix : (r < c ? cs.yclip_prefix : 0);
                if (diag == 0) {
                    const size_t d = std::min(r, c);
                    add_kmer((uint32_t)(r - d), (uint32_t)(c - d), d, w);
                    const uint32_t sr = (uint32_t)ssub(r, lazy), sc = (uint32_t)ssub(c, lazy);
                    const uint32_t er = (uint32_t)(r - d), ec = (uint32_t)(c - d);
                    if (sr <= er && sc <= ec) add_gap(sr, sc, er, ec, w);
                } else {
                    add_gap(0, 0, first.x, first.y, w);
                }
            }
        }
    }
    {
        const size_t r = (size_t)last.x + k, c = (size_t)last.y + k;
        if (!(r == rows && c == cols)) {
            const int32_t from_end = (r == rows ? 0 : cs.xclip_suffix) + (c == cols ? 0 : cs.yclip_suffix);
            const size_t dr = rows - r, dc = cols - c;
            bool diagonal = false;
            size_t d = 0;
            if (from_end == 0) {
                d = std::min(lazy, std::min(dr, dc));
                diagonal = true;
            } else {
                const int32_t diag = dr > dc ? cs.xclip_suffix : (dr < dc ? cs.yclip_suffix : 0);
                if (diag == 0) {
                    d = std::min(dr, dc);
                    diagonal = true;
                }
            }
            if (diagonal) {
                add_kmer((uint32_t)r, (uint32_t)c, d, w);
                const size_t r1 = std::min(rows, r + d) - 1, c1 = std::min(cols, c + d) - 1;
                const size_t r2 = std::min(rows, r + lazy), c2 = std::min(cols, c + lazy);
                if (r1 <= r2 && c1 <= c2) add_gap((uint32_t)r1, (uint32_t)c1, (uint32_t)r2, (uint32_t)c2, w);
            } else {
                add_gap((uint32_t)r, (uint32_t)c, (uint32_t)rows, (uint32_t)cols, w);
            }
        }
    }
}

uint64_t Band::num_cells() const {  // banded.rs:1374-1380
    uint64_t cells = 0;
    for (size_t j = 0; j < cols; j++) cells += end[j] > start[j] ? end[j] - start[j] : 0;
    return cells;
}

bool Band::monotone() const {
    uint32_t ps = 0, pe = 0;
    bool any = false;
    for (size_t j = 0; j < cols; j++) {
        if (end[j] <= start[j]) continue;
        if (any && (start[j] < ps || end[j] < pe)) return false;
        ps = start[j];
        pe = end[j];
        any = true;
    }
    return true;
}

// BG_TRACE diagnostics: thread-CPU nanoseconds spent in k-mer matching / sparse DP / band rasterisation
std::atomic<uint64_t> g_prof[4];
static inline uint64_t cpu_ns() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec; }
#define PROF_T0 uint64_t pt0 = cpu_ns()
#define PROF_LAP(i) do { uint64_t pt1 = cpu_ns(); g_prof[i] += pt1 - pt0; pt0 = pt1; } while (0)
// Band::create_from_match_path (banded.rs:1330-1367)
void Band::create_from_match_path(size_t m, size_t n, size_t k, size_t w, const ClipScores& cs,
                                  const std::vector<uint32_t>& path, const std::vector<Match>& mm) {
    reset(m, n);
    if (mm.empty() || path.empty()) {  // banded.rs:1341-1344 (an empty path with matches panics in the reference)
        std::fill(start.begin(), start.end(), 0u);
        std::fill(end.begin(), end.end(), (uint32_t)rows);
        return;
    }
    set_boundaries(mm[path.front()], mm[path.back()], k, w, cs);
    bool have = false;
    Match prev{0, 0};
    for (uint32_t idx : path) {
        const Match cur = mm[idx];
        if (have && cur.x == prev.x + 1 && cur.y == prev.y + 1) {
            add_entry(prev.x + (uint32_t)k, prev.y + (uint32_t)k, w);
        } else {
            if (have) add_gap(prev.x + (uint32_t)(k - 1), prev.y + (uint32_t)(k - 1), cur.x, cur.y, w);
            add_kmer(cur.x, cur.y, k, w);
        }
        prev = cur;
        have = true;
    }
}

// Band::create_with_matches (banded.rs:1301-1328); false if the matches are not sorted (sparse.rs:212-217)
bool Band::create_with_matches(size_t m, size_t n, size_t k, size_t w, const ClipScores& cs, const std::vector<Match>& mm,
                               Workspace& ws) {
    if (mm.empty()) {  // banded.rs:1309-1313
        reset(m, n);
        std::fill(start.begin(), start.end(), 0u);
        std::fill(end.begin(), end.end(), (uint32_t)rows);
        return true;
    }
    for (size_t i = 1; i < mm.size(); i++)
        if (!(mm[i - 1] < mm[i])) return false;
    const uint32_t reward = (uint32_t)(cs.match_scores_some ? cs.match_score : 2);  // banded.rs:105,1315-1318
    PROF_T0;
    sdpkpp_path(mm, k, reward, cs.gap_open, cs.gap_extend, ws.path);
    PROF_LAP(1);
    create_from_match_path(m, n, k, w, cs, ws.path, mm);
    PROF_LAP(2);
    return true;
}

// Band::create (banded.rs:1278-1287)
void Band::create(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, const ClipScores& cs,
                  Workspace& ws) {
    PROF_T0;
    find_kmer_matches(x, m, y, n, k, ws.matches);
    PROF_LAP(0);
    create_with_matches(m, n, k, w, cs, ws.matches, ws);
}

}  // namespace bgband
