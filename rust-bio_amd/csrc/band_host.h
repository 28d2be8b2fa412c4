// Host-side band construction of the banded aligner (see band_host.cpp).
#ifndef BG_BAND_HOST_H
#define BG_BAND_HOST_H
#include <cstddef>
#include <cstdint>
#include <vector>

namespace bgband {

struct Match {  // (x position, y position) of a common k-mer; ordered like the reference's tuple
    uint32_t x, y;
    bool operator<(const Match& o) const { return x < o.x || (x == o.x && y < o.y); }
};

struct ClipScores {  // the fields of Scoring<F> the band construction reads
    int32_t gap_open, gap_extend;
    int32_t xclip_prefix, xclip_suffix, yclip_prefix, yclip_suffix;
    int32_t match_score;
    bool match_scores_some;
};

struct Workspace {
    std::vector<Match> matches;
    std::vector<uint32_t> path;
};

void find_kmer_matches(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, std::vector<Match>& out);
// sparse.rs:188-295 / 67-143 / 297-329 / 404-500; `false` = the matches are not sorted (the reference asserts)
bool sdpkpp_path(const std::vector<Match>& matches, size_t k, uint32_t match_score, int32_t gap_open,
                 int32_t gap_extend, std::vector<uint32_t>& path);
bool lcskpp_path(const std::vector<Match>& matches, size_t k, std::vector<uint32_t>& path, uint32_t* score);
bool sdpkpp_union_lcskpp_path(const std::vector<Match>& matches, size_t k, uint32_t match_score, int32_t gap_open,
                              int32_t gap_extend, std::vector<uint32_t>& path);
bool expand_kmer_matches(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, const std::vector<Match>& sorted,
                         size_t allowed_mismatches, std::vector<Match>& out);

struct Band {
    size_t rows = 0, cols = 0;
    std::vector<uint32_t> start, end;  // per column: half-open row range [start, end)
    void reset(size_t m, size_t n);
    void add_entry(uint32_t r, uint32_t c, size_t w);
    void add_kmer(uint32_t r, uint32_t c, size_t k, size_t w);
    void add_gap(uint32_t r0, uint32_t c0, uint32_t r1, uint32_t c1, size_t w);
    void set_boundaries(Match first, Match last, size_t k, size_t w, const ClipScores& cs);
    void create(const uint8_t* x, size_t m, const uint8_t* y, size_t n, size_t k, size_t w, const ClipScores& cs,
                Workspace& ws);
    bool create_with_matches(size_t m, size_t n, size_t k, size_t w, const ClipScores& cs, const std::vector<Match>& matches,
                             Workspace& ws);
    void create_from_match_path(size_t m, size_t n, size_t k, size_t w, const ClipScores& cs,
                                const std::vector<uint32_t>& path, const std::vector<Match>& matches);
    uint64_t num_cells() const;
    bool monotone() const;  // starts and ends of non-empty columns never decrease
};

}  // namespace bgband
#endif
