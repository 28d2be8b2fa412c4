// ORACLE (test infrastructure, not product code) — see oracle.h.
#ifndef BIOGPU_ORACLE_SPARSE_IMPL_H
#define BIOGPU_ORACLE_SPARSE_IMPL_H
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "oracle.h"

namespace orc {
typedef std::pair<uint32_t, uint32_t> Match;  // (x position, y position), ordered as a tuple
struct SparseResult {                          // sparse.rs:40-47
    std::vector<size_t> path;
    uint32_t score = 0;
};
SparseResult lcskpp(const std::vector<Match>& matches, size_t k);
SparseResult sdpkpp(const std::vector<Match>& matches, size_t k, uint32_t match_score, int32_t gap_open,
                    int32_t gap_extend);
std::vector<size_t> sdpkpp_union_lcskpp_path(const std::vector<Match>& matches, size_t k, uint32_t match_score,
                                             int32_t gap_open, int32_t gap_extend);
std::vector<Match> expand_kmer_matches(const uint8_t* seq1, size_t n1, const uint8_t* seq2, size_t n2, size_t k,
                                       const std::vector<Match>& sorted_matches, size_t allowed_mismatches);
std::vector<Match> find_kmer_matches(const uint8_t* seq1, size_t n1, const uint8_t* seq2, size_t n2, size_t k);
}  // namespace orc
#endif
