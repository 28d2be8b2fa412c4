// ORACLE (test infrastructure, not product code) — see oracle.h.
// Shared types of the pairwise oracle: Scoring, TracebackCell, Traceback, Alignment, Aligner.
// Follows /root/reference/src/alignment/pairwise/mod.rs (lines cited per item).
#ifndef BIOGPU_ORACLE_PAIRWISE_IMPL_H
#define BIOGPU_ORACLE_PAIRWISE_IMPL_H
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "oracle.h"

namespace orc {

constexpr int32_t MIN_SCORE = ORC_MIN_SCORE;  // mod.rs:174

struct OracleError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// mod.rs:238-247 Scoring<F>; match_fn = MatchParams (mod.rs:208-217) or a tabulated closure
struct Scoring {
    int32_t gap_open = 0, gap_extend = 0;
    int32_t xclip_prefix = MIN_SCORE, xclip_suffix = MIN_SCORE;
    int32_t yclip_prefix = MIN_SCORE, yclip_suffix = MIN_SCORE;
    int32_t match_score = 0, mismatch_score = 0;
    bool match_scores_some = false;
    const int32_t* matrix = nullptr;
    inline int32_t score(uint8_t a, uint8_t b) const {
        if (matrix) return matrix[(size_t)a * 256 + b];
        return a == b ? match_score : mismatch_score;
    }
};

// mod.rs:1030-1047
constexpr uint8_t I_POS = 0, D_POS = 4, S_POS = 8;
constexpr uint16_t TB_START = 0b0000, TB_INS = 0b0001, TB_DEL = 0b0010, TB_SUBST = 0b0011,
                   TB_MATCH = 0b0100, TB_XCLIP_PREFIX = 0b0101, TB_XCLIP_SUFFIX = 0b0110,
                   TB_YCLIP_PREFIX = 0b0111, TB_YCLIP_SUFFIX = 0b1000, TB_MAX = 0b1000;

// mod.rs:1026-1114
struct TracebackCell {
    uint16_t v = 0;
    inline void set_bits(uint8_t pos, uint16_t value) {
        const uint16_t bits = (uint16_t)(0b1111 << pos);
        if (value > TB_MAX) throw OracleError("Expected a value <= TB_MAX");
        v = (uint16_t)((v & ~bits) | (value << pos));
    }
    inline void set_i_bits(uint16_t value) { set_bits(I_POS, value); }
    inline void set_d_bits(uint16_t value) { set_bits(D_POS, value); }
    inline void set_s_bits(uint16_t value) { set_bits(S_POS, value); }
    inline uint16_t get_bits(uint8_t pos) const { return (uint16_t)((v >> pos) & 0b1111); }
    inline uint16_t get_i_bits() const { return get_bits(I_POS); }
    inline uint16_t get_d_bits() const { return get_bits(D_POS); }
    inline uint16_t get_s_bits() const { return get_bits(S_POS); }
    inline void set_all(uint16_t value) {
        set_i_bits(value);
        set_d_bits(value);
        set_s_bits(value);
    }
};

// mod.rs:1118-1168 — row-major (m+1) x (n+1), every cell TB_START after init
struct Traceback {
    size_t rows = 0, cols = 0;
    std::vector<TracebackCell> matrix;
    void init(size_t m, size_t n) {
        matrix.clear();
        TracebackCell start;
        start.set_all(TB_START);
        rows = m + 1;
        cols = n + 1;
        matrix.resize(rows * cols, start);
    }
    inline void set(size_t i, size_t j, TracebackCell v) { matrix[i * cols + j] = v; }
    inline const TracebackCell& get(size_t i, size_t j) const { return matrix[i * cols + j]; }
    inline TracebackCell& get_mut(size_t i, size_t j) { return matrix[i * cols + j]; }
};

struct Op {
    uint8_t kind;
    size_t len;
    bool operator==(const Op& o) const { return kind == o.kind && len == o.len; }
};

// bio_types::alignment::Alignment as constructed at mod.rs:911-921
struct Alignment {
    int32_t score = 0;
    size_t ystart = 0, xstart = 0, yend = 0, xend = 0, ylen = 0, xlen = 0;
    std::vector<Op> operations;
    int mode = ORC_MODE_CUSTOM;
    // bio-types filter_clip_operations: retain everything except Xclip/Yclip
    void filter_clip_operations() {
        operations.erase(std::remove_if(operations.begin(), operations.end(),
                                        [](const Op& o) {
                                            return o.kind == ORC_OP_XCLIP || o.kind == ORC_OP_YCLIP;
                                        }),
                         operations.end());
    }
    bool operator==(const Alignment& o) const {
        return score == o.score && ystart == o.ystart && xstart == o.xstart && yend == o.yend &&
               xend == o.xend && ylen == o.ylen && xlen == o.xlen && operations == o.operations &&
               mode == o.mode;
    }
};

// mod.rs:472-481
struct Aligner {
    std::vector<int32_t> I[2], D[2], S[2];
    std::vector<size_t> Lx, Ly;
    std::vector<int32_t> Sn;
    Traceback traceback;
    Scoring scoring;
    explicit Aligner(const Scoring& s) : scoring(s) {}
    Alignment custom(const uint8_t* x, size_t m, const uint8_t* y, size_t n);
    template <bool LF_HOOK>
    Alignment custom_impl(const uint8_t* x, size_t m, const uint8_t* y, size_t n);  // LF_HOOK: tests only (pairwise.cpp)
    Alignment global(const uint8_t* x, size_t m, const uint8_t* y, size_t n);
    Alignment semiglobal(const uint8_t* x, size_t m, const uint8_t* y, size_t n);
    Alignment local(const uint8_t* x, size_t m, const uint8_t* y, size_t n);
    Alignment run(int mode, const uint8_t* x, size_t m, const uint8_t* y, size_t n);

   private:
    Alignment with_clips(int32_t xp, int32_t xs, int32_t yp, int32_t ys, int mode, bool filter,
                         const uint8_t* x, size_t m, const uint8_t* y, size_t n);
};

int export_alignment(const Alignment& a, orc_alignment_t* out, uint64_t* ops, uint64_t ops_cap);
Scoring scoring_from_c(const orc_scoring_t* sc);

}  // namespace orc
#endif
