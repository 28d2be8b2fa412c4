// ORACLE (test infrastructure, not product code) — see oracle.h.
//
// Restatement of rust-bio 4.0.1 `bio::alignment::sparse`
// (/root/reference/src/alignment/sparse.rs) and the prefix-max Fenwick tree it uses
// (/root/reference/src/data_structures/bit_tree.rs:45-101).
#include "sparse_impl.h"

#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>

namespace orc {

// bit_tree.rs:45-87 — FenwickTree<T, MaxOp>; T::default() is the identity
template <typename T>
struct MaxBitTree {
    std::vector<T> tree;
    explicit MaxBitTree(size_t len) : tree(len + 1, T{}) {}
    T get(size_t idx) const {
        idx += 1;
        T sum{};
        while (idx > 0) {
            sum = std::max(sum, tree[idx]);
            idx -= idx & (~idx + 1);
        }
        return sum;
    }
    void set(size_t idx, const T& val) {
        idx += 1;
        while (idx < tree.size()) {
            tree[idx] = std::max(tree[idx], val);
            idx += idx & (~idx + 1);
        }
    }
};

// sparse.rs:145-167 — derived Ord = lexicographic over the fields in declaration order
struct PrevPtr {
    uint32_t plane = 0, score = 0, d = 0;
    size_t id = 0;
    uint32_t x = 0, y = 0;
    static PrevPtr make(uint32_t score, uint32_t x, uint32_t y, size_t id, uint32_t gap_extend) {
        PrevPtr p;
        p.d = x + y;
        p.plane = score + p.d * gap_extend;
        p.score = score;
        p.id = id;
        p.x = x;
        p.y = y;
        return p;
    }
    bool operator<(const PrevPtr& o) const {
        return std::tie(plane, score, d, id, x, y) < std::tie(o.plane, o.score, o.d, o.id, o.x, o.y);
    }
};

static void check_sorted(const std::vector<Match>& matches) {
    for (size_t i = 1; i < matches.size(); i++)
        if (!(matches[i - 1] < matches[i])) throw std::runtime_error("incoming matches must be sorted");
}

static bool find_match(const std::vector<Match>& matches, Match key, size_t* idx) {
    auto it = std::lower_bound(matches.begin(), matches.end(), key);
    if (it != matches.end() && *it == key) {
        *idx = (size_t)(it - matches.begin());
        return true;
    }
    return false;
}

// sparse.rs:67-143
SparseResult lcskpp(const std::vector<Match>& matches, size_t k_) {
    SparseResult res;
    if (matches.empty()) return res;
    const uint32_t k = (uint32_t)k_;
    check_sorted(matches);
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> events;
    uint32_t n = 0;
    const uint32_t nm = (uint32_t)matches.size();
    for (uint32_t idx = 0; idx < nm; idx++) {
        const uint32_t x = matches[idx].first, y = matches[idx].second;
        events.emplace_back(x, y, idx + nm);
        events.emplace_back(x + k, y + k, idx);
        n = std::max(n, x + k);
        n = std::max(n, y + k);
    }
    std::sort(events.begin(), events.end());
    MaxBitTree<std::pair<uint32_t, uint32_t>> max_col_dp(n);
    std::vector<std::pair<uint32_t, int32_t>> dp(events.size(), {0, 0});
    std::pair<uint32_t, int32_t> best_dp(k, 0);
    for (const auto& ev : events) {
        const size_t p = std::get<2>(ev) % nm;
        const uint32_t j = std::get<1>(ev);
        const bool is_start = std::get<2>(ev) >= nm;
        if (is_start) {
            dp[p] = {k, -1};
            const auto best = max_col_dp.get(j);
            if (best.first > 0) {
                dp[p] = {k + best.first, (int32_t)best.second};
                best_dp = std::max(best_dp, std::pair<uint32_t, int32_t>(dp[p].first, (int32_t)p));
            }
        } else {
            if (std::get<0>(ev) > k && std::get<1>(ev) > k) {
                size_t cont;
                if (find_match(matches, {std::get<0>(ev) - k - 1, std::get<1>(ev) - k - 1}, &cont)) {
                    const std::pair<uint32_t, int32_t> cand(dp[cont].first + 1, (int32_t)cont);
                    dp[p] = std::max(dp[p], cand);
                    best_dp = std::max(best_dp, std::pair<uint32_t, int32_t>(dp[p].first, (int32_t)p));
                }
            }
            max_col_dp.set(std::get<1>(ev), {dp[p].first, (uint32_t)p});
        }
    }
    int32_t prev = best_dp.second;
    while (prev >= 0) {
        res.path.push_back((size_t)prev);
        prev = dp[prev].second;
    }
    std::reverse(res.path.begin(), res.path.end());
    res.score = best_dp.first;
    return res;
}

// sparse.rs:188-295
SparseResult sdpkpp(const std::vector<Match>& matches, size_t k_, uint32_t match_score,
                    int32_t gap_open, int32_t gap_extend) {
    SparseResult res;
    if (matches.empty()) return res;
    const uint32_t k = (uint32_t)k_;
    if (!(gap_open <= 0 && gap_extend <= 0)) throw std::runtime_error("gap parameters cannot be positive");
    const uint32_t go = (uint32_t)(-gap_open), ge = (uint32_t)(-gap_extend);
    check_sorted(matches);
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> events;
    uint32_t n = 0;
    const uint32_t nm = (uint32_t)matches.size();
    for (uint32_t idx = 0; idx < nm; idx++) {
        const uint32_t x = matches[idx].first, y = matches[idx].second;
        events.emplace_back(x, y, idx + nm);
        events.emplace_back(x + k, y + k, idx);
        n = std::max(n, x + k);
        n = std::max(n, y + k);
    }
    std::sort(events.begin(), events.end());
    MaxBitTree<PrevPtr> max_col_dp(n);
    std::vector<std::pair<uint32_t, int32_t>> dp(events.size(), {0, 0});
    std::pair<uint32_t, int32_t> best_dp(k, 0);
    for (const auto& ev : events) {
        const size_t p = std::get<2>(ev) % nm;
        const uint32_t j = std::get<1>(ev);
        const bool is_start = std::get<2>(ev) >= nm;
        if (is_start) {
            dp[p] = {k * match_score, -1};
            const PrevPtr best_prev = max_col_dp.get(j);
            if (best_prev.score > 0) {
                const uint32_t cur_x = std::get<0>(ev), cur_y = std::get<1>(ev);
                const uint32_t gap = std::max(cur_x - best_prev.x, cur_y - best_prev.y);
                const uint32_t gap_penalty = gap > 0 ? go + gap * ge : 0;
                const uint32_t reward = k * match_score;
                const uint32_t sum = best_prev.score + reward;
                const uint32_t new_score = sum > gap_penalty ? sum - gap_penalty : 0;  // saturating_sub
                dp[p] = std::max(dp[p], std::pair<uint32_t, int32_t>(new_score, (int32_t)best_prev.id));
                best_dp = std::max(best_dp, std::pair<uint32_t, int32_t>(dp[p].first, (int32_t)p));
            }
        } else {
            if (std::get<0>(ev) > k && std::get<1>(ev) > k) {
                size_t cont;
                if (find_match(matches, {std::get<0>(ev) - k - 1, std::get<1>(ev) - k - 1}, &cont)) {
                    const std::pair<uint32_t, int32_t> cand(dp[cont].first + match_score, (int32_t)cont);
                    dp[p] = std::max(dp[p], cand);
                    best_dp = std::max(best_dp, std::pair<uint32_t, int32_t>(dp[p].first, (int32_t)p));
                }
            }
            max_col_dp.set(std::get<1>(ev), PrevPtr::make(dp[p].first, std::get<0>(ev), std::get<1>(ev), p, ge));
        }
    }
    int32_t prev = best_dp.second;
    while (prev >= 0) {
        res.path.push_back((size_t)prev);
        prev = dp[prev].second;
    }
    std::reverse(res.path.begin(), res.path.end());
    res.score = best_dp.first;
    return res;
}

// sparse.rs:337-402 — the hash map only groups equal k-mers; the result is sorted, so any
// associative container gives the same matches
std::vector<Match> find_kmer_matches(const uint8_t* seq1, size_t n1, const uint8_t* seq2, size_t n2, size_t k) {
    std::vector<Match> matches;
    const bool hash1 = n1 < n2;  // sparse.rs:338: the shorter one is hashed, seq2 on ties
    const uint8_t* hs = hash1 ? seq1 : seq2;
    const size_t hn = hash1 ? n1 : n2;
    const uint8_t* os = hash1 ? seq2 : seq1;
    const size_t on = hash1 ? n2 : n1;
    std::map<std::string, std::vector<uint32_t>> set;
    for (size_t i = 0; i + k <= hn; i++) set[std::string((const char*)hs + i, k)].push_back((uint32_t)i);
    for (size_t i = 0; i + k <= on; i++) {
        auto it = set.find(std::string((const char*)os + i, k));
        if (it == set.end()) continue;
        for (uint32_t pos : it->second) {
            if (hash1)
                matches.emplace_back(pos, (uint32_t)i);
            else
                matches.emplace_back((uint32_t)i, pos);
        }
    }
    std::sort(matches.begin(), matches.end());
    return matches;
}

// sparse.rs:297-329
std::vector<size_t> sdpkpp_union_lcskpp_path(const std::vector<Match>& matches, size_t k, uint32_t match_score,
                                             int32_t gap_open, int32_t gap_extend) {
    if (matches.empty()) return {};
    const SparseResult lcs = lcskpp(matches, k);
    const SparseResult sdp = sdpkpp(matches, k, match_score, gap_open, gap_extend);
    // binary_search(..).unwrap_or(0) / Ok(ind) => ind + 1, Err(_) => len
    auto bsearch = [&](size_t key, size_t* idx) {
        auto it = std::lower_bound(lcs.path.begin(), lcs.path.end(), key);
        if (it != lcs.path.end() && *it == key) {
            *idx = (size_t)(it - lcs.path.begin());
            return true;
        }
        return false;
    };
    size_t pre = 0, post = lcs.path.size(), t;
    if (bsearch(sdp.path[0], &t)) pre = t;
    if (bsearch(sdp.path.back(), &t)) post = t + 1;
    std::vector<size_t> u;
    for (size_t i = 0; i < pre; i++) u.push_back(lcs.path[i]);
    for (size_t i = 0; i < sdp.path.size(); i++) u.push_back(sdp.path[i]);
    for (size_t i = post; i < lcs.path.size(); i++) u.push_back(lcs.path[i]);
    return u;
}

// sparse.rs:404-500
std::vector<Match> expand_kmer_matches(const uint8_t* seq1, size_t n1, const uint8_t* seq2, size_t n2, size_t k,
                                       const std::vector<Match>& sorted_matches, size_t allowed_mismatches) {
    check_sorted(sorted_matches);
    typedef std::pair<int32_t, int32_t> P;
    std::map<int32_t, P> last_match_along_diagonal;  // HashMapFx in the reference (lookups only)
    std::vector<Match> left(sorted_matches);
    for (const Match& tm : sorted_matches) {
        const int32_t diag = (int32_t)tm.first - (int32_t)tm.second;
        const int32_t min_xy = (int32_t)std::min(tm.first, tm.second);
        const P dflt((int32_t)tm.first - min_xy - 1, (int32_t)tm.second - min_xy - 1);
        auto it = last_match_along_diagonal.find(diag);
        const P last_match = it != last_match_along_diagonal.end() ? it->second : dflt;
        size_t n_mismatches = 0;
        P curr((int32_t)tm.first - 1, (int32_t)tm.second - 1);
        for (;;) {
            if (last_match >= curr) break;
            n_mismatches += seq1[curr.first] == seq2[curr.second] ? 0 : 1;
            if (n_mismatches > allowed_mismatches) break;
            left.emplace_back((uint32_t)curr.first, (uint32_t)curr.second);
            curr = P(curr.first - 1, curr.second - 1);
        }
        last_match_along_diagonal[diag] = P((int32_t)tm.first, (int32_t)tm.second);
    }
    std::sort(left.begin(), left.end());
    std::vector<Match> expanded(left);
    std::reverse(left.begin(), left.end());
    std::map<int32_t, Match> next_match_along_diagonal;
    for (const Match& tm : left) {
        const int32_t diag = (int32_t)tm.first - (int32_t)tm.second;
        const uint32_t room = std::min((uint32_t)n1 - tm.first, (uint32_t)n2 - tm.second);
        const uint32_t max_inc = room > (uint32_t)k - 1 ? room - ((uint32_t)k - 1) : 0;  // saturating_sub
        auto it = next_match_along_diagonal.find(diag);
        const Match next_match = it != next_match_along_diagonal.end() ? it->second : Match(tm.first + max_inc, tm.second + max_inc);
        size_t n_mismatches = 0;
        Match curr(tm.first + 1, tm.second + 1);
        for (;;) {
            if (curr >= next_match) break;
            n_mismatches += seq1[curr.first + k - 1] == seq2[curr.second + k - 1] ? 0 : 1;
            if (n_mismatches > allowed_mismatches) break;
            expanded.push_back(curr);
            curr = Match(curr.first + 1, curr.second + 1);
        }
        next_match_along_diagonal[diag] = tm;
    }
    std::sort(expanded.begin(), expanded.end());
    return expanded;
}

}  // namespace orc

extern "C" uint64_t orc_find_kmer_matches(const uint8_t* x, uint64_t m, const uint8_t* y, uint64_t n,
                                          uint32_t k, uint32_t* out_xy, uint64_t cap) {
    auto mm = orc::find_kmer_matches(x, m, y, n, k);
    for (uint64_t i = 0; i < mm.size() && i < cap; i++) {
        out_xy[2 * i] = mm[i].first;
        out_xy[2 * i + 1] = mm[i].second;
    }
    return mm.size();
}

static std::vector<orc::Match> to_matches(const uint32_t* xy, uint64_t n) {
    std::vector<orc::Match> v(n);
    for (uint64_t i = 0; i < n; i++) v[i] = {xy[2 * i], xy[2 * i + 1]};
    return v;
}

extern "C" uint64_t orc_sdpkpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k,
                               uint32_t match_score, int32_t gap_open, int32_t gap_extend,
                               uint32_t* path, uint64_t cap, uint32_t* score) {
    auto r = orc::sdpkpp(to_matches(matches_xy, n_matches), k, match_score, gap_open, gap_extend);
    for (uint64_t i = 0; i < r.path.size() && i < cap; i++) path[i] = (uint32_t)r.path[i];
    *score = r.score;
    return r.path.size();
}

extern "C" uint64_t orc_lcskpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k,
                               uint32_t* path, uint64_t cap, uint32_t* score) {
    auto r = orc::lcskpp(to_matches(matches_xy, n_matches), k);
    for (uint64_t i = 0; i < r.path.size() && i < cap; i++) path[i] = (uint32_t)r.path[i];
    *score = r.score;
    return r.path.size();
}

extern "C" uint64_t orc_sdpkpp_union_lcskpp_path(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k,
                                                 uint32_t match_score, int32_t gap_open, int32_t gap_extend,
                                                 uint32_t* path, uint64_t cap) {
    auto r = orc::sdpkpp_union_lcskpp_path(to_matches(matches_xy, n_matches), k, match_score, gap_open, gap_extend);
    for (uint64_t i = 0; i < r.size() && i < cap; i++) path[i] = (uint32_t)r[i];
    return r.size();
}

extern "C" uint64_t orc_expand_kmer_matches(const uint8_t* x, uint64_t m, const uint8_t* y, uint64_t n, uint32_t k,
                                            const uint32_t* matches_xy, uint64_t n_matches, uint32_t allowed_mismatches,
                                            uint32_t* out_xy, uint64_t cap) {
    auto r = orc::expand_kmer_matches(x, m, y, n, k, to_matches(matches_xy, n_matches), allowed_mismatches);
    for (uint64_t i = 0; i < r.size() && i < cap; i++) {
        out_xy[2 * i] = r[i].first;
        out_xy[2 * i + 1] = r[i].second;
    }
    return r.size();
}
