// Host-side table builders behind the C ABI: suffix array (SA-IS), BWT, Less.
//
// These stand where rust-bio's own host code runs
//   suffix_array  /root/reference/src/data_structures/suffix_array.rs:264-284 (+ transform_text 444-466)
//   bwt           /root/reference/src/data_structures/bwt.rs:39-49
//   less          /root/reference/src/data_structures/bwt.rs:186-199
// and produce identical arrays.  The suffix array is unique once the reference's sentinel
// transform is applied (every sentinel becomes a distinct symbol; the first occurrence gets
// the largest sentinel rank, the last one rank 0), so this induced-sorting implementation —
// written from the published algorithm (Nong, Zhang, Chan 2011), recursive on the reduced
// LMS string, bucket-pointer arrays, type bits in a byte vector — only has to be a correct
// suffix sorter.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "biogpu.h"

namespace {

// s: text over [0, K) whose last symbol is the unique minimum; sa: output, length n.
template <typename Sym, typename Idx>
class InducedSorter {
   public:
    static void run(const Sym* s, Idx* sa, Idx n, Idx K) {
        if (n == 1) {
            sa[0] = 0;
            return;
        }
        InducedSorter w(s, sa, n, K);
        w.sort();
    }

   private:
    const Sym* s;
    Idx* sa;
    Idx n, K;
    std::vector<uint8_t> stype;  // 1 = S-type, 0 = L-type
    std::vector<Idx> bkt;
    static constexpr Idx EMPTY = (Idx)-1;

    InducedSorter(const Sym* s_, Idx* sa_, Idx n_, Idx K_) : s(s_), sa(sa_), n(n_), K(K_) {}

    bool is_lms(Idx i) const { return i > 0 && stype[i] && !stype[i - 1]; }

    void bucket_bounds(bool ends) {
        std::fill(bkt.begin(), bkt.end(), 0);
        for (Idx i = 0; i < n; i++) bkt[s[i]]++;
        Idx sum = 0;
        for (Idx c = 0; c < K; c++) {
            sum += bkt[c];
            bkt[c] = ends ? sum : sum - bkt[c];
        }
    }

    void induce() {
        // L-type suffixes, left to right from bucket heads
        bucket_bounds(false);
        for (Idx i = 0; i < n; i++) {
            Idx j = sa[i];
            if (j != EMPTY && j > 0 && !stype[j - 1]) sa[bkt[s[j - 1]]++] = j - 1;
        }
        // S-type suffixes, right to left from bucket tails
        bucket_bounds(true);
        for (Idx i = n; i-- > 0;) {
            Idx j = sa[i];
            if (j != EMPTY && j > 0 && stype[j - 1]) sa[--bkt[s[j - 1]]] = j - 1;
        }
    }

    void sort() {
        stype.assign(n, 0);
        stype[n - 1] = 1;
        for (Idx i = n - 1; i-- > 0;)
            stype[i] = (s[i] < s[i + 1] || (s[i] == s[i + 1] && stype[i + 1])) ? 1 : 0;
        bkt.assign(K, 0);

        // pass 1: sort LMS substrings by inducing from unsorted LMS positions
        bucket_bounds(true);
        std::fill(sa, sa + n, EMPTY);
        for (Idx i = 1; i < n; i++)
            if (is_lms(i)) sa[--bkt[s[i]]] = i;
        induce();

        // compact the sorted LMS positions to the front
        Idx n1 = 0;
        for (Idx i = 0; i < n; i++)
            if (sa[i] != EMPTY && is_lms(sa[i])) sa[n1++] = sa[i];
        // name LMS substrings; names are parked at sa[n1 + pos/2]
        std::fill(sa + n1, sa + n, EMPTY);
        Idx name = 0, prev = EMPTY;
        for (Idx i = 0; i < n1; i++) {
            Idx pos = sa[i];
            bool diff = (prev == EMPTY);
            if (!diff) {
                for (Idx d = 0;; d++) {
                    if (pos + d >= n || prev + d >= n || s[pos + d] != s[prev + d] ||
                        stype[pos + d] != stype[prev + d]) {
                        diff = true;
                        break;
                    }
                    if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
                }
            }
            if (diff) {
                name++;
                prev = pos;
            }
            sa[n1 + pos / 2] = name - 1;
        }
        // gather the reduced string at the tail of sa
        Idx j = n;
        for (Idx i = n; i-- > n1;)
            if (sa[i] != EMPTY) sa[--j] = sa[i];
        Idx* s1 = sa + n - n1;
        Idx* sa1 = sa;
        if (name < n1) {
            InducedSorter<Idx, Idx>::run(s1, sa1, n1, name);
        } else {
            for (Idx i = 0; i < n1; i++) sa1[s1[i]] = i;
        }

        // pass 2: map reduced suffixes back to text positions and induce the full order
        j = 0;
        for (Idx i = 1; i < n; i++)
            if (is_lms(i)) s1[j++] = i;
        for (Idx i = 0; i < n1; i++) sa1[i] = s1[sa1[i]];
        std::fill(sa + n1, sa + n, EMPTY);
        bucket_bounds(true);
        for (Idx i = n1; i-- > 0;) {
            Idx p = sa[i];
            sa[i] = EMPTY;
            sa[--bkt[s[p]]] = p;
        }
        induce();
    }
};

template <typename Idx>
int suffix_array_impl(const uint8_t* text, uint64_t n, uint64_t* sa_out) {
    // sentinel / sentinel_count (suffix_array.rs:426-441)
    const uint8_t sentinel = text[n - 1];
    uint64_t sentinel_count = 0;
    bool present[256] = {};
    for (uint64_t i = 0; i < n; i++) {
        if (text[i] < sentinel) return BG_ERR_SENTINEL;
        sentinel_count += text[i] == sentinel;
        present[text[i]] = true;
    }
    // transform_text (suffix_array.rs:444-466): RankTransform over the symbols present
    uint32_t rank[256] = {};
    uint32_t alpha_len = 0;
    for (int c = 0; c < 256; c++)
        if (present[c]) rank[c] = alpha_len++;
    const uint64_t offset = sentinel_count - 1;
    const uint64_t K = alpha_len + sentinel_count - 1;  // symbols 0 .. alpha_len+offset-1
    std::vector<Idx> sa(n);
    if (K <= 256) {
        std::vector<uint8_t> t(n);
        uint64_t s = sentinel_count;
        for (uint64_t i = 0; i < n; i++)
            t[i] = text[i] == sentinel ? (uint8_t)(--s) : (uint8_t)(rank[text[i]] + offset);
        InducedSorter<uint8_t, Idx>::run(t.data(), sa.data(), (Idx)n, (Idx)K);
    } else {
        std::vector<Idx> t(n);
        uint64_t s = sentinel_count;
        for (uint64_t i = 0; i < n; i++)
            t[i] = text[i] == sentinel ? (Idx)(--s) : (Idx)(rank[text[i]] + offset);
        InducedSorter<Idx, Idx>::run(t.data(), sa.data(), (Idx)n, (Idx)K);
    }
    for (uint64_t i = 0; i < n; i++) sa_out[i] = (uint64_t)sa[i];
    return BG_OK;
}

}  // namespace

extern "C" int bg_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa_out) {
    if (!text || !sa_out || n == 0) return BG_ERR_INVALID_ARG;
    if (n < 0x7FFFFFF0ull) return suffix_array_impl<uint32_t>(text, n, sa_out);
    return suffix_array_impl<uint64_t>(text, n, sa_out);
}

extern "C" int bg_bwt(const uint8_t* text, const uint64_t* sa, uint64_t n, uint8_t* bwt_out) {
    if (!text || !sa || !bwt_out) return BG_ERR_INVALID_ARG;
    for (uint64_t r = 0; r < n; r++) {
        const uint64_t p = sa[r];
        if (p >= n) return BG_ERR_INVALID_ARG;
        bwt_out[r] = p > 0 ? text[p - 1] : text[n - 1];
    }
    return BG_OK;
}

extern "C" int bg_less(const uint8_t* bwt, uint64_t n, const uint8_t* alphabet, uint32_t n_sym,
                       uint64_t* less_out, uint32_t* less_len) {
    if (!alphabet || n_sym == 0 || !less_len) return BG_ERR_INVALID_ARG;
    uint32_t max_symbol = 0;
    for (uint32_t i = 0; i < n_sym; i++) max_symbol = std::max<uint32_t>(max_symbol, alphabet[i]);
    const uint32_t m = max_symbol + 2;
    *less_len = m;
    if (!less_out) return BG_OK;
    if (!bwt && n) return BG_ERR_INVALID_ARG;
    std::vector<uint64_t> cnt(m, 0);
    for (uint64_t i = 0; i < n; i++) {
        if (bwt[i] >= m) return BG_ERR_OUT_OF_ALPHABET;  // less[c as usize] out of bounds
        cnt[bwt[i]]++;
    }
    uint64_t acc = 0;
    for (uint32_t c = 0; c < m; c++) {
        less_out[c] = acc;
        acc += cnt[c];
    }
    return BG_OK;
}
