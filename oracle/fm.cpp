// ORACLE (test infrastructure, not product code) — see oracle.h.
//
// Restatement of rust-bio 4.0.1's FM-index path:
//   suffix_array      /root/reference/src/data_structures/suffix_array.rs:264-284, 426-466
//   bwt / less / Occ  /root/reference/src/data_structures/bwt.rs:39-49, 94-125, 129-182, 186-199
//   backward_search   /root/reference/src/data_structures/fmindex.rs:144-208
//   Alphabet          /root/reference/src/alphabets/mod.rs:49-60, 91-116
// The suffix array uses prefix doubling instead of SA-IS: the reference's sentinel transform
// (transform_text) makes every suffix distinct, so the suffix array is unique and any correct
// construction reproduces it.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "oracle.h"

static uint64_t g_fmd_ext_calls = 0;  // backward_ext calls (forward_ext goes through it) since the last reset
extern "C" uint64_t orc_fmd_ext_calls(int reset) {
    const uint64_t v = g_fmd_ext_calls;
    if (reset) g_fmd_ext_calls = 0;
    return v;
}

namespace {

// alphabets/mod.rs:37-116 — a set of bytes
struct Alphabet {
    bool has[256] = {};
    Alphabet(const uint8_t* syms, uint64_t n) {
        for (uint64_t i = 0; i < n; i++) has[syms[i]] = true;
    }
    int max_symbol() const {
        for (int c = 255; c >= 0; c--)
            if (has[c]) return c;
        return -1;
    }
    size_t len() const {
        size_t k = 0;
        for (bool b : has) k += b;
        return k;
    }
};

}  // namespace

struct orc_occ {
    std::vector<std::vector<uint64_t>> occ;
    uint32_t k;
};

// suffix_array.rs:426-441 sentinel / sentinel_count, 444-466 transform_text, 264-284 suffix_array
extern "C" int orc_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa_out) {
    if (n == 0) return -1;  // text[text.len() - 1] panics
    const uint8_t sentinel = text[n - 1];
    uint64_t sentinel_count = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (text[i] < sentinel) return -1;  // assert!(text.iter().all(|&a| a >= sentinel))
        sentinel_count += text[i] == sentinel;
    }
    // RankTransform over Alphabet::new(text): rank = index among present symbols
    Alphabet alpha(text, n);
    uint32_t ranks[256];
    uint32_t r = 0;
    for (int c = 0; c < 256; c++)
        if (alpha.has[c]) ranks[c] = r++;
    const uint64_t offset = sentinel_count - 1;
    std::vector<uint64_t> t(n);
    uint64_t s = sentinel_count;
    for (uint64_t i = 0; i < n; i++) {
        if (text[i] == sentinel) {
            s -= 1;
            t[i] = s;  // first sentinel gets the largest sentinel rank, the last one 0
        } else {
            t[i] = ranks[text[i]] + offset;
        }
    }
    // prefix doubling on the transformed text
    std::vector<uint64_t> sa(n), rank(t), tmp(n);
    std::iota(sa.begin(), sa.end(), 0);
    for (uint64_t k = 1;; k <<= 1) {
        auto key2 = [&](uint64_t i) -> int64_t { return i + k < n ? (int64_t)rank[i + k] : -1; };
        auto cmp = [&](uint64_t a, uint64_t b) {
            if (rank[a] != rank[b]) return rank[a] < rank[b];
            return key2(a) < key2(b);
        };
        if (k == 1) {
            std::sort(sa.begin(), sa.end(), [&](uint64_t a, uint64_t b) {
                if (rank[a] != rank[b]) return rank[a] < rank[b];
                return key2(a) < key2(b);
            });
        } else {
            std::sort(sa.begin(), sa.end(), cmp);
        }
        tmp[sa[0]] = 0;
        for (uint64_t i = 1; i < n; i++) tmp[sa[i]] = tmp[sa[i - 1]] + (cmp(sa[i - 1], sa[i]) ? 1 : 0);
        rank = tmp;
        if (rank[sa[n - 1]] == n - 1) break;
        if (k > n) break;
    }
    std::copy(sa.begin(), sa.end(), sa_out);
    return 0;
}

// bwt.rs:39-49
extern "C" void orc_bwt(const uint8_t* text, const uint64_t* pos, uint64_t n, uint8_t* bwt) {
    for (uint64_t r = 0; r < n; r++) {
        uint64_t p = pos[r];
        bwt[r] = p > 0 ? text[p - 1] : text[n - 1];
    }
}

// bwt.rs:186-199 (+ utils::prescan, utils/mod.rs:25-34: exclusive prefix sum)
extern "C" uint64_t orc_less(const uint8_t* bwt, uint64_t n, const uint8_t* alphabet,
                             uint64_t n_sym, uint64_t* less_out) {
    Alphabet alpha(alphabet, n_sym);
    const uint64_t m = (uint64_t)alpha.max_symbol() + 2;
    if (!less_out) return m;
    std::vector<uint64_t> less(m, 0);
    for (uint64_t i = 0; i < n; i++) {
        if (bwt[i] >= m) return 0;  // index out of bounds panic
        less[bwt[i]] += 1;
    }
    uint64_t acc = 0;
    for (uint64_t i = 0; i < m; i++) {
        uint64_t v = less[i];
        less[i] = acc;
        acc += v;
    }
    std::copy(less.begin(), less.end(), less_out);
    return m;
}

// bwt.rs:94-125
extern "C" orc_occ* orc_occ_new(const uint8_t* bwt, uint64_t n, uint32_t k,
                                const uint8_t* alphabet, uint64_t n_sym) {
    Alphabet alphab(alphabet, n_sym);
    const uint64_t m = (uint64_t)alphab.max_symbol() + 1;
    std::vector<uint64_t> alpha;
    for (int c = 0; c < 256; c++)
        if (alphab.has[c]) alpha.push_back(c);
    // include sentinel '$'
    if ((uint64_t)'$' < m && !alphab.has['$']) alpha.push_back('$');
    auto* o = new orc_occ;
    o->k = k;
    o->occ.assign(m, {});
    std::vector<uint64_t> curr_occ(m, 0);
    for (uint64_t a : alpha) o->occ[a].reserve(n / k);
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = bwt[i];
        if (c >= m) {  // curr_occ[c as usize] out of bounds → panic
            delete o;
            return nullptr;
        }
        curr_occ[c] += 1;
        if (i % k == 0)
            for (uint64_t a : alpha) o->occ[a].push_back(curr_occ[a]);
    }
    return o;
}

extern "C" void orc_occ_free(orc_occ* o) { delete o; }

extern "C" const uint64_t* orc_occ_row(const orc_occ* o, uint32_t a, uint64_t* len) {
    if (a >= o->occ.size()) {
        *len = 0;
        return nullptr;
    }
    *len = o->occ[a].size();
    return o->occ[a].data();
}

static inline uint64_t bytecount(const uint8_t* b, uint64_t lo, uint64_t hi_incl, uint8_t a) {
    // bytecount::count(&bwt[lo..=hi_incl], a)
    uint64_t c = 0;
    for (uint64_t i = lo; i <= hi_incl && i + 1 != 0; i++) c += b[i] == a;
    return c;
}

// bwt.rs:129-182
extern "C" int orc_occ_get(const orc_occ* o, const uint8_t* bwt, uint64_t n, uint64_t r,
                           uint8_t a, uint64_t* out) {
    const uint64_t k = o->k;
    if (a >= o->occ.size()) return -1;
    const std::vector<uint64_t>& row = o->occ[a];
    const uint64_t lo_checkpoint = r / k;
    if (lo_checkpoint >= row.size()) return -1;
    const uint64_t lo_occ = row[lo_checkpoint];
    if (k > 64) {
        const uint64_t hi_checkpoint = lo_checkpoint + 1;
        if (hi_checkpoint < row.size()) {
            const uint64_t hi_occ = row[hi_checkpoint];
            if (lo_occ == hi_occ) {
                *out = lo_occ;
                return 0;
            }
            const uint64_t hi_idx = hi_checkpoint * k;
            if ((hi_idx - r) < (k / 2)) {
                if (hi_idx >= n) return -1;
                *out = hi_occ - (r + 1 <= hi_idx ? bytecount(bwt, r + 1, hi_idx, a) : 0);
                return 0;
            }
        }
    }
    const uint64_t lo_idx = lo_checkpoint * k;
    if (r >= n) return -1;
    *out = (lo_idx + 1 <= r ? bytecount(bwt, lo_idx + 1, r, a) : 0) + lo_occ;
    return 0;
}

// fmindex.rs:144-208
extern "C" int orc_backward_search(const uint8_t* bwt, uint64_t n, const uint64_t* less,
                                   uint64_t less_len, const orc_occ* occ, const uint8_t* pattern,
                                   uint64_t plen, uint64_t* lower, uint64_t* upper,
                                   uint64_t* matched_len_out) {
    uint64_t l = 0, r = n - 1;
    uint64_t pl = l, pr = r;
    uint64_t matched_len = 0;
    bool complete_match = true;
    *lower = *upper = *matched_len_out = 0;

    for (uint64_t t = plen; t-- > 0;) {
        const uint8_t a = pattern[t];
        if (a >= less_len) return ORC_BS_PANIC;  // self.less.borrow()[a as usize]
        const uint64_t less_a = less[a];
        pl = l;
        pr = r;
        uint64_t occ_r;
        if (orc_occ_get(occ, bwt, n, r, a, &occ_r)) return ORC_BS_PANIC;
        if (occ_r == 0) {
            complete_match = false;
            break;
        }
        uint64_t occ_l = 0;
        if (l > 0 && orc_occ_get(occ, bwt, n, l - 1, a, &occ_l)) return ORC_BS_PANIC;
        l = less_a + (l > 0 ? occ_l : 0);
        r = less_a + occ_r - 1;
        if (l > r) {
            complete_match = false;
            break;
        }
        matched_len += 1;
    }

    if (matched_len > 0) {
        if (complete_match) {
            *lower = l;
            *upper = r + 1;
            *matched_len_out = matched_len;
            return ORC_BS_COMPLETE;
        }
        *lower = pl;
        *upper = pr + 1;
        *matched_len_out = matched_len;
        return ORC_BS_PARTIAL;
    }
    return ORC_BS_ABSENT;
}

extern "C" void orc_backward_search_batch(const uint8_t* bwt, uint64_t n, const uint64_t* less,
                                          uint64_t less_len, const orc_occ* occ, uint64_t n_q,
                                          const uint8_t* pat, const uint64_t* pat_off,
                                          uint8_t* tag, uint64_t* lower, uint64_t* upper,
                                          uint64_t* matched_len, int threads) {
    if (threads < 1) threads = 1;
    auto work = [&](int t) {
        // contiguous shards: one shared read-only index, as in the reference's Arc example
        // (src/lib.rs:173-210)
        uint64_t lo = n_q * t / threads, hi = n_q * (t + 1) / threads;
        for (uint64_t q = lo; q < hi; q++)
            tag[q] = (uint8_t)orc_backward_search(bwt, n, less, less_len, occ, pat + pat_off[q],
                                                  pat_off[q + 1] - pat_off[q], &lower[q],
                                                  &upper[q], &matched_len[q]);
    };
    if (threads == 1) {
        work(0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(work, t);
    for (auto& t : th) t.join();
}

// ---- SampledSuffixArray (suffix_array.rs:86-184) -------------------------------------------
// `sample` (86-120): every s-th SA entry is kept; rows whose BWT character is the sentinel (the last
// byte of the text, suffix_array.rs `sentinel()`) are kept as extra rows unless already sampled.
struct orc_sampled_sa {
    std::vector<uint64_t> sample;
    std::vector<std::pair<uint64_t, uint64_t>> extra;  // (row, position), sorted by row (HashMap in the reference)
    uint64_t s = 1;
    uint8_t sentinel = 0;
};

extern "C" orc_sampled_sa* orc_sa_sample(const uint64_t* sa, uint64_t n, const uint8_t* text, const uint8_t* bwt,
                                         uint64_t sampling_rate) {
    orc_sampled_sa* h = new orc_sampled_sa;
    h->s = sampling_rate;
    h->sentinel = n ? text[n - 1] : 0;
    for (uint64_t i = 0; i < n; i++) {  // suffix_array.rs:100-111
        if (i % sampling_rate == 0)
            h->sample.push_back(sa[i]);
        else if (bwt[i] == h->sentinel)
            h->extra.push_back({i, sa[i]});
    }
    return h;
}
extern "C" void orc_sa_sample_free(orc_sampled_sa* h) { delete h; }
extern "C" uint64_t orc_sa_sample_counts(const orc_sampled_sa* h, uint64_t* n_extra) {
    if (n_extra) *n_extra = h->extra.size();
    return h->sample.size();
}
extern "C" void orc_sa_sample_export(const orc_sampled_sa* h, uint64_t* sample, uint64_t* extra_row, uint64_t* extra_pos) {
    std::copy(h->sample.begin(), h->sample.end(), sample);
    for (size_t i = 0; i < h->extra.size(); i++) {
        extra_row[i] = h->extra[i].first;
        extra_pos[i] = h->extra[i].second;
    }
}

// SampledSuffixArray::get (suffix_array.rs:157-184).  Returns -1 for None (index >= len) and -2 where the
// reference would panic (Occ::get on a byte outside the alphabet).
extern "C" int orc_sampled_sa_get(const orc_sampled_sa* h, const uint8_t* bwt, uint64_t n, const uint64_t* less,
                                  uint64_t less_len, const orc_occ* occ, uint64_t index, uint64_t* out) {
    if (index >= n) return -1;
    uint64_t pos = index, offset = 0;
    for (;;) {
        if (pos % h->s == 0) {  // 162-164
            *out = h->sample[pos / h->s] + offset;
            return 0;
        }
        const uint8_t c = bwt[pos];
        if (c == h->sentinel) {  // 168-175
            auto it = std::lower_bound(h->extra.begin(), h->extra.end(), std::make_pair(pos, (uint64_t)0));
            if (it == h->extra.end() || it->first != pos) return -2;
            *out = it->second + offset;
            return 0;
        }
        uint64_t o = 0;
        if ((uint64_t)c >= less_len || orc_occ_get(occ, bwt, n, pos - 1, c, &o) != 0) return -2;
        pos = less[c] + o;  // 177-178
        offset += 1;
    }
}

// ---- FMDIndex (fmindex.rs:250-576): bi-directional search and supermaximal exact matches ----------
namespace {

struct BiInterval {  // fmindex.rs:254-259
    uint64_t lower = 0, lower_rev = 0, size = 0, match_size = 0;
    BiInterval swapped() const { return BiInterval{lower_rev, lower, size, match_size}; }  // 275-282
};

struct Panic {};

// dna::complement (alphabets/dna.rs:37-69): IUPAC table, identity elsewhere, case preserved
uint8_t dna_complement(uint8_t a) {
    static uint8_t comp[256];
    static bool init = false;
    if (!init) {
        for (int v = 0; v < 256; v++) comp[v] = (uint8_t)v;
        const char* from = "AGCTYRWSKMDVHBN";
        const char* to = "TCGARYWSMKHBDVN";
        for (int i = 0; from[i]; i++) {
            comp[(uint8_t)from[i]] = (uint8_t)to[i];
            comp[(uint8_t)from[i] + 32] = (uint8_t)(to[i] + 32);
        }
        init = true;
    }
    return comp[a];
}

struct Fmd {
    const uint8_t* bwt;
    uint64_t n;
    const uint64_t* less;
    uint64_t less_len;
    const orc_occ* occ;
    uint64_t less_of(uint64_t a) const {  // fmindex.rs:228-230: index out of bounds panics
        if (a >= less_len) throw Panic();
        return less[a];
    }
    uint64_t occ_of(uint64_t r, uint8_t a) const {
        uint64_t o = 0;
        if (r >= n || orc_occ_get(occ, bwt, n, r, a, &o) != 0) throw Panic();
        return o;
    }
    // fmindex.rs:504-514
    BiInterval init_interval_with(uint8_t a) const {
        const uint8_t comp_a = dna_complement(a);
        const uint64_t lower = less_of(a);
        return BiInterval{lower, less_of(comp_a), less_of((uint64_t)a + 1) - lower, 1};
    }
    // fmindex.rs:517-524
    BiInterval init_interval() const { return BiInterval{0, 0, n, 0}; }
    // fmindex.rs:527-558
    BiInterval backward_ext(const BiInterval& interval, uint8_t a) const {
        g_fmd_ext_calls++;  // (measurement aid of bench.py's SMEM leg: extensions the algorithm makes per read)
        uint64_t s = 0, o = 0, l = interval.lower_rev;
        for (const char* p = "$TGCNAtgcna"; *p; p++) {
            const uint8_t b = (uint8_t)*p;
            l += s;
            o = interval.lower == 0 ? 0 : occ_of(interval.lower - 1, b);
            if (interval.lower + interval.size == 0) throw Panic();  // usize underflow
            s = occ_of(interval.lower + interval.size - 1, b) - o;
            if (b == a) break;
        }
        const uint64_t k = less_of(a) + o;
        return BiInterval{k, l, s, interval.match_size + 1};
    }
    // fmindex.rs:560-564
    BiInterval forward_ext(const BiInterval& interval, uint8_t a) const {
        return backward_ext(interval.swapped(), dna_complement(a)).swapped();
    }
    struct Smem {
        BiInterval iv;
        uint64_t pos, len;
    };
    // fmindex.rs:363-434
    std::vector<Smem> smems(const uint8_t* pattern, uint64_t plen, uint64_t i, uint64_t l) const {
        std::vector<std::pair<BiInterval, uint64_t>> curr, prev;
        std::vector<Smem> matches;
        if (i >= plen) throw Panic();  // pattern[i]
        uint64_t match_len = 0;
        BiInterval interval = init_interval_with(pattern[i]);
        if (interval.size != 0) match_len += 1;
        for (uint64_t t = i + 1; t < plen; t++) {
            const BiInterval fwd = forward_ext(interval, pattern[t]);
            if (interval.size != fwd.size) curr.push_back({interval, match_len});
            if (fwd.size == 0) break;
            interval = fwd;
            match_len += 1;
        }
        curr.push_back({interval, match_len});
        std::reverse(curr.begin(), curr.end());
        std::swap(curr, prev);
        int64_t j = (int64_t)plen;
        for (int64_t k = (int64_t)i - 1; k >= -1; k--) {
            const uint8_t a = k == -1 ? (uint8_t)'$' : pattern[k];
            curr.clear();
            int64_t last_size = -1;
            for (const auto& e : prev) {
                const BiInterval fwd = backward_ext(e.first, a);
                if ((fwd.size == 0 || k == -1) && curr.empty() && k < j && e.second >= l) {
                    j = k;
                    matches.push_back({e.first, (uint64_t)(k + 1), e.second});
                }
                if (fwd.size != 0 && (int64_t)fwd.size != last_size) {
                    last_size = (int64_t)fwd.size;
                    curr.push_back({fwd, e.second + 1});
                }
            }
            if (curr.empty()) break;
            std::swap(curr, prev);
        }
        return matches;
    }
    // fmindex.rs:479-501
    std::vector<Smem> all_smems(const uint8_t* pattern, uint64_t plen, uint64_t l) const {
        std::vector<Smem> out;
        uint64_t i0 = 0;
        while (i0 < plen) {
            std::vector<Smem> cur = smems(pattern, plen, i0, l);
            uint64_t next_i0 = i0 + 1;
            for (const Smem& s : cur)
                if (s.pos + s.len > next_i0) next_i0 = s.pos + s.len;
            i0 = next_i0;
            out.insert(out.end(), cur.begin(), cur.end());
        }
        return out;
    }
};

}  // namespace

// FMDIndex::from (fmindex.rs:311-329): 1 if the BWT is a word over n_alphabet + '$'
extern "C" int orc_fmd_check(const uint8_t* bwt, uint64_t n) {
    for (uint64_t i = 0; i < n; i++)
        if (!strchr("ACGTNacgtn$", bwt[i]) || bwt[i] == 0) return 0;
    return 1;
}

// smems (all == 0: overlapping position i) / all_smems (all != 0).  Records of 6 uint64:
// lower, lower_rev, size, match_size, pattern position, length.  Returns the number of records
// (may exceed cap), or -1 where the reference would panic.
extern "C" int64_t orc_fmd_smems(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                                 const orc_occ* occ, const uint8_t* pattern, uint64_t plen, uint64_t i, uint64_t l,
                                 int all, uint64_t* out, uint64_t cap) {
    const Fmd f{bwt, n, less, less_len, occ};
    try {
        const std::vector<Fmd::Smem> r = all ? f.all_smems(pattern, plen, l) : f.smems(pattern, plen, i, l);
        for (size_t t = 0; t < r.size() && t < cap; t++) {
            uint64_t* o = out + 6 * t;
            o[0] = r[t].iv.lower;
            o[1] = r[t].iv.lower_rev;
            o[2] = r[t].iv.size;
            o[3] = r[t].iv.match_size;
            o[4] = r[t].pos;
            o[5] = r[t].len;
        }
        return (int64_t)r.size();
    } catch (const Panic&) {
        return -1;
    }
}

// init_interval (op 0), init_interval_with(a) (op 1), backward_ext (op 2), forward_ext (op 3); iv = 4 uint64
extern "C" int orc_fmd_interval(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                                const orc_occ* occ, int op, const uint64_t* iv, uint8_t a, uint64_t* out) {
    const Fmd f{bwt, n, less, less_len, occ};
    try {
        BiInterval r;
        const BiInterval in = iv ? BiInterval{iv[0], iv[1], iv[2], iv[3]} : BiInterval{};
        if (op == 0)
            r = f.init_interval();
        else if (op == 1)
            r = f.init_interval_with(a);
        else if (op == 2)
            r = f.backward_ext(in, a);
        else
            r = f.forward_ext(in, a);
        out[0] = r.lower;
        out[1] = r.lower_rev;
        out[2] = r.size;
        out[3] = r.match_size;
        return 0;
    } catch (const Panic&) {
        return -1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Interval of a pattern BY DEFINITION, without a suffix array (test infrastructure for texts the restatement above cannot
// sort in test time: the 4.4 G-symbol run of tools/exp/fm_wide_big.py).  The interval backward_search returns for a pattern
// that occurs — Interval { lower, upper } over the suffix array, fmindex.rs:63-79, 100-102, 144-208 — is the range of
// suffixes that start with it; in a suffix array sorted like suffix_array.rs:264-284 sorts (plain byte order; the text
// ends in a unique smallest sentinel) that is
//      lower = #{ i : text[i..] < P },   upper = lower + #{ i : P is a prefix of text[i..] }
// and the positions Interval::occ yields are those i.  One pass over the text for a batch of patterns, `threads` slices;
// pos_out[p * pos_cap ..] receives up to pos_cap occurrence positions of pattern p, ascending.
// Pinned: tests/test_oracle_fm.py compares it with orc_backward_search + orc_suffix_array on small texts.
extern "C" void orc_intervals_by_scan(const uint8_t* text, uint64_t n, uint64_t n_pat, const uint8_t* pat, const uint64_t* pat_off,
                                      uint64_t* lower, uint64_t* upper, uint64_t* pos_out, uint64_t pos_cap, uint64_t* n_pos,
                                      int threads) {
    if (threads < 1) threads = 1;
    // Thousands of patterns per pass (round 6: the stratified sample of the 64-bit index needs >= 2 000 intervals beyond 2^32):
    // comparing every suffix with every pattern of its first byte is out of reach, so the comparison is cut in two by the
    // first kK bytes.  key(i) = text[i .. i + kK) as a big-endian integer, bytes past the end as 0 (a suffix that ends is
    // smaller than anything longer; patterns hold no 0 byte — one that does, or is shorter than kK, takes the slow path
    // below).  For a pattern P with key kp:  text[i..] < P  iff  key(i) < kp, or key(i) == kp and the rest compares smaller —
    // the first part is a histogram over the sorted distinct pattern keys (one binary search per suffix, entered through a
    // table over the first two bytes), the second a byte comparison for the few suffixes that share P's first kK bytes.
    constexpr uint64_t kK = 8;
    std::vector<uint32_t> fast, slow;
    for (uint64_t p = 0; p < n_pat; p++) {
        const uint64_t m = pat_off[p + 1] - pat_off[p];
        bool zero = false;
        for (uint64_t k = 0; k < m; k++) zero = zero || pat[pat_off[p] + k] == 0;
        if (m >= kK && !zero)
            fast.push_back((uint32_t)p);
        else if (m)
            slow.push_back((uint32_t)p);
    }
    auto key_of = [&](uint32_t p) {
        uint64_t k = 0;
        for (uint64_t u = 0; u < kK; u++) k = (k << 8) | pat[pat_off[p] + u];
        return k;
    };
    std::vector<uint64_t> keys;  // sorted, distinct
    for (uint32_t p : fast) keys.push_back(key_of(p));
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    const size_t nk = keys.size();
    std::vector<std::vector<uint32_t>> with_key(nk);  // the patterns of each key
    for (uint32_t p : fast) with_key[(size_t)(std::lower_bound(keys.begin(), keys.end(), key_of(p)) - keys.begin())].push_back(p);
    std::vector<uint32_t> entry(65537, 0);  // entry[h] = first key whose top two bytes are >= h
    {
        size_t j = 0;
        for (uint32_t h = 0; h <= 65536; h++) {
            while (j < nk && (keys[j] >> 48) < h) j++;
            entry[h] = (uint32_t)j;
        }
    }
    std::vector<std::vector<uint32_t>> bucket(256);  // slow patterns by first byte
    for (uint32_t p : slow) bucket[pat[pat_off[p]]].push_back(p);
    struct Part {
        std::vector<uint64_t> less, pref, gap, eq;  // gap[j]: suffixes with keys[j-1] < key < keys[j]; eq[j]: key == keys[j]
        std::vector<std::vector<uint64_t>> pos;
        uint64_t hist[256];
    };
    std::vector<Part> parts(threads);
    auto work = [&](int t) {
        Part& P = parts[t];
        P.less.assign(n_pat, 0);
        P.pref.assign(n_pat, 0);
        P.pos.assign(n_pat, {});
        P.gap.assign(nk + 1, 0);
        P.eq.assign(nk + 1, 0);
        std::fill(P.hist, P.hist + 256, 0);
        const uint64_t lo = n * (uint64_t)t / (uint64_t)threads, hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
        uint64_t key = 0;
        for (uint64_t u = 0; u + 1 < kK; u++) key = (key << 8) | (lo + u < n ? text[lo + u] : 0);
        for (uint64_t i = lo; i < hi; i++) {
            key = (key << 8) | (i + kK - 1 < n ? text[i + kK - 1] : 0);
            const uint8_t c = text[i];
            P.hist[c]++;
            if (nk) {
                const uint32_t h = (uint32_t)(key >> 48);
                size_t a = entry[h], b = entry[h + 1];  // first key >= `key` lies in [a, b]
                while (a < b) {
                    const size_t mid = (a + b) >> 1;
                    if (keys[mid] < key)
                        a = mid + 1;
                    else
                        b = mid;
                }
                if (a < nk && keys[a] == key) {
                    P.eq[a]++;
                    for (uint32_t p : with_key[a]) {  // the first kK bytes agree: compare on
                        const uint8_t* q = pat + pat_off[p];
                        const uint64_t m = pat_off[p + 1] - pat_off[p];
                        uint64_t k = kK;
                        while (k < m && i + k < n && text[i + k] == q[k]) k++;
                        if (k == m) {
                            P.pref[p]++;
                            if (P.pos[p].size() < pos_cap) P.pos[p].push_back(i);
                        } else if (i + k >= n || text[i + k] < q[k]) {
                            P.less[p]++;
                        }
                    }
                } else {
                    P.gap[a]++;
                }
            }
            for (uint32_t p : bucket[c]) {  // slow path (the definition, byte by byte): first bytes agree
                const uint8_t* q = pat + pat_off[p];
                const uint64_t m = pat_off[p + 1] - pat_off[p];
                uint64_t k = 1;
                while (k < m && i + k < n && text[i + k] == q[k]) k++;
                if (k == m) {
                    P.pref[p]++;
                    if (P.pos[p].size() < pos_cap) P.pos[p].push_back(i);
                } else if (i + k >= n || text[i + k] < q[k]) {  // the suffix ended (a proper prefix of P) or its byte is smaller
                    P.less[p]++;
                }
            }
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    // suffixes whose key is below keys[j]
    std::vector<uint64_t> below(nk + 1, 0);
    {
        uint64_t acc = 0;
        for (size_t j = 0; j < nk; j++) {
            for (int t = 0; t < threads; t++) acc += parts[t].gap[j];
            below[j] = acc;
            for (int t = 0; t < threads; t++) acc += parts[t].eq[j];
        }
    }
    std::vector<uint8_t> is_fast(n_pat, 0);
    for (uint32_t p : fast) is_fast[p] = 1;
    for (uint64_t p = 0; p < n_pat; p++) {
        const uint64_t m = pat_off[p + 1] - pat_off[p];
        uint64_t less = 0, pref = 0, np = 0;
        for (int t = 0; t < threads; t++) {
            less += parts[t].less[p];
            pref += parts[t].pref[p];
            if (m && !is_fast[p])
                for (int c = 0; c < pat[pat_off[p]]; c++) less += parts[t].hist[c];  // suffixes with a smaller first byte
            for (uint64_t v : parts[t].pos[p])
                if (np < pos_cap) pos_out[p * pos_cap + np++] = v;
        }
        if (is_fast[p]) less += below[(size_t)(std::lower_bound(keys.begin(), keys.end(), key_of((uint32_t)p)) - keys.begin())];
        if (m == 0) pref = n;  // every suffix starts with the empty pattern
        lower[p] = less;
        upper[p] = less + pref;
        n_pos[p] = np;
    }
}
