// biogpu.hpp — C++17 host-side mirror of rust-bio's API for the accelerated path, on top of the C ABI
// (biogpu.h).  rust-bio is compiled Rust; no Rust toolchain exists in the build image, so this header
// plays the role of the Rust shim of INTEGRATION.md: same module paths (as namespaces), same names,
// same argument meaning, and the reference's panics/asserts as exceptions
// (bio::Panic ~ `panic!`, messages copied where the reference has one).
//
//   bio::alignment::{Alignment, AlignmentOperation, AlignmentMode}       (bio-types; mod.rs:911-921)
//   bio::alignment::pairwise::{Scoring, MatchParams, Aligner, MIN_SCORE} (pairwise/mod.rs:174-1015)
//   bio::alignment::pairwise::banded::Aligner                            (pairwise/banded.rs:122-1004)
//   bio::alphabets::{Alphabet, dna::{alphabet, n_alphabet, iupac_alphabet}}
//   bio::data_structures::suffix_array::{suffix_array, RawSuffixArray, SampledSuffixArray}
//   bio::data_structures::bwt::{bwt, less, Occ}
//   bio::data_structures::fmindex::{FMIndex, Interval, BackwardSearchResult}
//
// Additions the reference does not have: `*_batch` methods (the GPU wants many pairs/patterns per
// call; the single-item methods are batches of one) and `Context` (one per device).
#ifndef BIOGPU_HPP
#define BIOGPU_HPP
#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "biogpu.h"

namespace bio {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

using Text = std::vector<uint8_t>;
inline Text text(const char* s) { return Text(s, s + std::char_traits<char>::length(s)); }
inline Text text(const std::string& s) { return Text(s.begin(), s.end()); }

// One bg_ctx per device; not thread-safe, like the reference's `&mut Aligner` workspace.
class Context {
public:
    explicit Context(int device = 0) {
        const int rc = bg_init(device, &h_);
        if (rc) throw Panic(std::string("bg_init: ") + bg_strerror(rc) + " (" + bg_last_error() + ")");
    }
    ~Context() { bg_free(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    bg_ctx* raw() const { return h_; }
    static std::shared_ptr<Context> shared_default() {
        static std::shared_ptr<Context> c = std::make_shared<Context>(0);
        return c;
    }

private:
    bg_ctx* h_ = nullptr;
};

inline void check(int rc, const char* what) {
    if (rc == BG_OK) return;
    if (rc == BG_ERR_SENTINEL) throw Panic(bg_strerror(rc));  // the reference's assert message
    throw Panic(std::string(what) + ": " + bg_strerror(rc));
}

namespace alignment {

// bio_types::alignment::AlignmentOperation
struct AlignmentOperation {
    enum Kind : uint8_t { Match, Subst, Del, Ins, Xclip, Yclip } kind;
    size_t len = 0;  // Xclip(len) / Yclip(len)
    bool operator==(const AlignmentOperation& o) const { return kind == o.kind && len == o.len; }
    bool operator!=(const AlignmentOperation& o) const { return !(*this == o); }
};
constexpr AlignmentOperation Match{AlignmentOperation::Match, 0}, Subst{AlignmentOperation::Subst, 0},
    Del{AlignmentOperation::Del, 0}, Ins{AlignmentOperation::Ins, 0};
constexpr AlignmentOperation Xclip(size_t n) { return {AlignmentOperation::Xclip, n}; }
constexpr AlignmentOperation Yclip(size_t n) { return {AlignmentOperation::Yclip, n}; }

enum class AlignmentMode : uint8_t { Custom = 0, Global = 1, Semiglobal = 2, Local = 3 };

// bio_types::alignment::Alignment (constructed at pairwise/mod.rs:911-921)
struct Alignment {
    int32_t score = 0;
    size_t ystart = 0, xstart = 0, yend = 0, xend = 0, ylen = 0, xlen = 0;
    std::vector<AlignmentOperation> operations;
    AlignmentMode mode = AlignmentMode::Custom;
    bool operator==(const Alignment& o) const {
        return score == o.score && ystart == o.ystart && xstart == o.xstart && yend == o.yend && xend == o.xend &&
               ylen == o.ylen && xlen == o.xlen && operations == o.operations && mode == o.mode;
    }
    // Alignment::cigar(hard_clip) of bio-types 1.0 (restated from its documentation: parity unpinned);
    // panics for AlignmentMode::Custom like the crate
    std::string cigar(bool hard_clip) const {
        bg_alignment_t rec = {};
        rec.xstart = (uint32_t)xstart;
        rec.xend = (uint32_t)xend;
        rec.xlen = (uint32_t)xlen;
        rec.n_ops = (uint32_t)operations.size();
        rec.mode = (uint8_t)mode;
        std::vector<uint8_t> ops(operations.size() + 1);
        for (size_t i = 0; i < operations.size(); i++) ops[i] = (uint8_t)operations[i].kind;
        std::string out(2 * operations.size() + 64, '\0');
        uint64_t off[2] = {0, 0};
        const int rc = bg_cigar_batch(Context::shared_default()->raw(), 1, &rec, ops.data(), operations.size(), hard_clip ? 1 : 0, &out[0], out.size(), off);
        if (rc == BG_ERR_UNSUPPORTED) throw Panic(" Cigar fn not supported for custom alignment mode");
        check(rc, "bg_cigar_batch");
        out.resize(off[1]);
        return out;
    }
    // Alignment::pretty(x, y, ncol) of bio-types (restated from the crate: parity unpinned): rows x / marks / y in
    // blocks of ncol columns; panics where the crate's row-length assert fires (a non-ASCII byte)
    std::string pretty(const Text& x, const Text& y, size_t ncol) const {
        bg_alignment_t rec = {};
        rec.score = score;
        rec.xstart = (uint32_t)xstart;
        rec.xend = (uint32_t)xend;
        rec.ystart = (uint32_t)ystart;
        rec.yend = (uint32_t)yend;
        rec.xlen = (uint32_t)xlen;
        rec.ylen = (uint32_t)ylen;
        rec.n_ops = (uint32_t)operations.size();
        rec.mode = (uint8_t)mode;
        std::vector<uint8_t> ops(operations.size() + 1);
        for (size_t i = 0; i < operations.size(); i++) {
            ops[i] = (uint8_t)operations[i].kind;
            if (operations[i].kind >= AlignmentOperation::Xclip && rec.n_clips < 4) rec.clip_len[rec.n_clips++] = (uint32_t)operations[i].len;
        }
        const uint64_t xo[2] = {0, x.size()}, yo[2] = {0, y.size()};
        std::string out(3 * (x.size() + y.size()) + 5 * ((x.size() + y.size()) / std::max<size_t>(ncol, 1) + 2) + 16, '\0');
        uint64_t off[2] = {0, 0};
        const int rc = bg_pretty_batch(Context::shared_default()->raw(), 1, &rec, ops.data(), operations.size(), x.data(), xo, y.data(), yo,
                                       (uint32_t)ncol, &out[0], out.size(), off);
        if (rc == BG_ERR_UNSUPPORTED) throw Panic("assertion failed: x_pretty.len() == inb_pretty.len()");
        check(rc, "bg_pretty_batch");
        out.resize(off[1]);
        return out;
    }
};

// alignment::sparse (sparse.rs): the pieces the banded aligner's entry points take or produce
namespace sparse {
using Match = std::pair<uint32_t, uint32_t>;  // (x position, y position)
namespace detail {
inline std::vector<uint32_t> flat(const std::vector<Match>& mm) {
    std::vector<uint32_t> v;
    for (auto& m : mm) {
        v.push_back(m.first);
        v.push_back(m.second);
    }
    return v;
}
inline std::vector<Match> pairs(const std::vector<uint32_t>& v, uint64_t n) {
    std::vector<Match> mm(n);
    for (uint64_t i = 0; i < n; i++) mm[i] = {v[2 * i], v[2 * i + 1]};
    return mm;
}
}  // namespace detail
// hash_kmers (sparse.rs:350-359): only the identity of the indexed sequence matters to the engine
struct KmerHash {
    Text seq;
    size_t k;
};
inline KmerHash hash_kmers(const Text& seq, size_t k) { return {seq, k}; }
inline std::vector<Match> find_kmer_matches(const Text& a, const Text& b, size_t k) {  // sparse.rs:337-348
    const uint64_t n = bg_sparse_find_kmer_matches(a.data(), a.size(), b.data(), b.size(), (uint32_t)k, nullptr, 0);
    std::vector<uint32_t> v(2 * n + 2);
    bg_sparse_find_kmer_matches(a.data(), a.size(), b.data(), b.size(), (uint32_t)k, v.data(), n);
    return detail::pairs(v, n);
}
inline std::vector<Match> find_kmer_matches_seq2_hashed(const Text& a, const KmerHash& h, size_t k) {  // sparse.rs:383-402
    if (h.k != k) throw Panic("k-mer hash built with a different k");
    return find_kmer_matches(a, h.seq, k);
}
inline std::vector<size_t> sdpkpp_union_lcskpp_path(const std::vector<Match>& mm, size_t k, uint32_t match_score,
                                                    int32_t gap_open, int32_t gap_extend) {  // sparse.rs:297-329
    const auto f = detail::flat(mm);
    std::vector<uint32_t> p(2 * mm.size() + 2);
    const uint64_t n = bg_sparse_sdpkpp_union_lcskpp_path(f.data(), mm.size(), (uint32_t)k, match_score, gap_open, gap_extend,
                                                          p.data(), p.size());
    if (n == UINT64_MAX) throw Panic("incoming matches must be sorted");
    return std::vector<size_t>(p.begin(), p.begin() + n);
}
inline std::vector<Match> expand_kmer_matches(const Text& a, const Text& b, size_t k, const std::vector<Match>& sorted_matches,
                                              size_t allowed_mismatches) {  // sparse.rs:404-500
    const auto f = detail::flat(sorted_matches);
    uint64_t cap = sorted_matches.size() + a.size() + b.size() + 8;
    for (;;) {
        std::vector<uint32_t> v(2 * cap);
        const uint64_t n = bg_sparse_expand_kmer_matches(a.data(), a.size(), b.data(), b.size(), (uint32_t)k, f.data(),
                                                         sorted_matches.size(), (uint32_t)allowed_mismatches, v.data(), cap);
        if (n == UINT64_MAX) throw Panic("incoming matches must be sorted");
        if (n <= cap) return detail::pairs(v, n);
        cap = n;
    }
}
}  // namespace sparse

namespace pairwise {

constexpr int32_t MIN_SCORE = BG_MIN_SCORE;  // mod.rs:174

using MatchFn = std::function<int32_t(uint8_t, uint8_t)>;

// MatchParams (mod.rs:186-217)
struct MatchParams {
    int32_t match_score, mismatch_score;
    MatchParams(int32_t m, int32_t mm) : match_score(m), mismatch_score(mm) {
        if (m < 0) throw Panic("match_score can't be negative");
        if (mm > 0) throw Panic("mismatch_score can't be positive");
    }
    int32_t score(uint8_t a, uint8_t b) const { return a == b ? match_score : mismatch_score; }
};

// Scoring<F> (mod.rs:238-429)
class Scoring {
public:
    int32_t gap_open, gap_extend;
    MatchFn match_fn;                                       // empty for MatchParams scoring
    std::optional<std::pair<int32_t, int32_t>> match_scores;  // Some only via from_scores (mod.rs:272)
    int32_t xclip_prefix = MIN_SCORE, xclip_suffix = MIN_SCORE, yclip_prefix = MIN_SCORE, yclip_suffix = MIN_SCORE;

    static Scoring from_scores(int32_t gap_open, int32_t gap_extend, int32_t match_score, int32_t mismatch_score) {
        check_gaps(gap_open, gap_extend);  // mod.rs:265-266
        MatchParams mp(match_score, mismatch_score);
        Scoring s(gap_open, gap_extend);
        s.match_scores = std::make_pair(match_score, mismatch_score);
        return s;
    }
    static Scoring new_(int32_t gap_open, int32_t gap_extend, MatchFn f) {  // mod.rs:291-305
        check_gaps(gap_open, gap_extend);
        Scoring s(gap_open, gap_extend);
        s.match_fn = std::move(f);
        return s;
    }
    // consuming builders (mod.rs:322-428); the trailing underscore avoids the field names
    Scoring xclip(int32_t p) const { return Scoring(*this).set(p, &Scoring::xclip_prefix, &Scoring::xclip_suffix); }
    Scoring yclip(int32_t p) const { return Scoring(*this).set(p, &Scoring::yclip_prefix, &Scoring::yclip_suffix); }
    Scoring xclip_prefix_(int32_t p) const { return Scoring(*this).set(p, &Scoring::xclip_prefix, nullptr); }
    Scoring xclip_suffix_(int32_t p) const { return Scoring(*this).set(p, &Scoring::xclip_suffix, nullptr); }
    Scoring yclip_prefix_(int32_t p) const { return Scoring(*this).set(p, &Scoring::yclip_prefix, nullptr); }
    Scoring yclip_suffix_(int32_t p) const { return Scoring(*this).set(p, &Scoring::yclip_suffix, nullptr); }

    // effective values for the FFI; the closure is tabulated over all byte pairs (it cannot cross the FFI)
    bg_scoring_t to_c(std::vector<int32_t>& table) const {
        bg_scoring_t c = {};
        c.gap_open = gap_open;
        c.gap_extend = gap_extend;
        c.xclip_prefix = xclip_prefix;
        c.xclip_suffix = xclip_suffix;
        c.yclip_prefix = yclip_prefix;
        c.yclip_suffix = yclip_suffix;
        if (match_scores) {
            c.match_score = match_scores->first;
            c.mismatch_score = match_scores->second;
            c.match_scores_some = 1;
            c.matrix = nullptr;
        } else {
            table.resize(65536);
            for (int a = 0; a < 256; a++)
                for (int b = 0; b < 256; b++) table[a * 256 + b] = match_fn((uint8_t)a, (uint8_t)b);
            c.matrix = table.data();
        }
        return c;
    }

private:
    Scoring(int32_t go, int32_t ge) : gap_open(go), gap_extend(ge) {}
    static void check_gaps(int32_t go, int32_t ge) {
        if (go > 0) throw Panic("gap_open can't be positive");
        if (ge > 0) throw Panic("gap_extend can't be positive");
    }
    Scoring& set(int32_t p, int32_t Scoring::*a, int32_t Scoring::*b) {
        if (p > 0) throw Panic("Clipping penalty can't be positive");  // mod.rs:322 ff.
        this->*a = p;
        if (b) this->*b = p;
        return *this;
    }
};

namespace detail {

struct Batch {
    Text x, y;
    std::vector<uint64_t> x_off{0}, y_off{0};
    void push(const Text& a, const Text& b) {
        x.insert(x.end(), a.begin(), a.end());
        y.insert(y.end(), b.begin(), b.end());
        x_off.push_back(x.size());
        y_off.push_back(y.size());
    }
    size_t size() const { return x_off.size() - 1; }
};

inline Alignment to_alignment(const bg_alignment_t& r, const uint8_t* ops) {
    Alignment a;
    a.score = r.score;
    a.xstart = r.xstart;
    a.xend = r.xend;
    a.ystart = r.ystart;
    a.yend = r.yend;
    a.xlen = r.xlen;
    a.ylen = r.ylen;
    a.mode = (AlignmentMode)r.mode;
    size_t clip = 0;
    for (uint32_t t = 0; t < r.n_ops; t++) {
        const uint8_t o = ops[r.ops_off + t];
        if (o == BG_OP_XCLIP)
            a.operations.push_back(Xclip(r.clip_len[clip++]));
        else if (o == BG_OP_YCLIP)
            a.operations.push_back(Yclip(r.clip_len[clip++]));
        else
            a.operations.push_back({(AlignmentOperation::Kind)o, 0});
    }
    return a;
}

}  // namespace detail

// Aligner<F> (mod.rs:472-1015)
class Aligner {
public:
    static Aligner new_(int32_t gap_open, int32_t gap_extend, MatchFn f, std::shared_ptr<Context> ctx = nullptr) {
        return Aligner(Scoring::new_(gap_open, gap_extend, std::move(f)), std::move(ctx));
    }
    static Aligner with_capacity(size_t, size_t, int32_t gap_open, int32_t gap_extend, MatchFn f,
                                 std::shared_ptr<Context> ctx = nullptr) {
        return new_(gap_open, gap_extend, std::move(f), std::move(ctx));
    }
    static Aligner with_scoring(Scoring s, std::shared_ptr<Context> ctx = nullptr) { return Aligner(std::move(s), std::move(ctx)); }
    static Aligner with_capacity_and_scoring(size_t, size_t, Scoring s, std::shared_ptr<Context> ctx = nullptr) {
        return Aligner(std::move(s), std::move(ctx));
    }

    Alignment custom(const Text& x, const Text& y) { return one(AlignmentMode::Custom, x, y); }          // mod.rs:591
    Alignment global(const Text& x, const Text& y) { return one(AlignmentMode::Global, x, y); }          // mod.rs:925
    Alignment semiglobal(const Text& x, const Text& y) { return one(AlignmentMode::Semiglobal, x, y); }  // mod.rs:954
    Alignment local(const Text& x, const Text& y) { return one(AlignmentMode::Local, x, y); }            // mod.rs:986

    std::vector<Alignment> align_batch(AlignmentMode mode, const std::vector<std::pair<Text, Text>>& pairs) {
        detail::Batch b;
        for (auto& p : pairs) b.push(p.first, p.second);
        std::vector<int32_t> table;
        const bg_scoring_t sc = scoring.to_c(table);
        std::vector<bg_alignment_t> out(b.size());
        std::vector<uint8_t> ops(b.x.size() + b.y.size() + 4 * b.size() + 8);
        uint64_t used = 0;
        check(bg_align_batch(ctx_->raw(), &sc, (int)mode, b.size(), b.x.data(), b.x_off.data(), b.y.data(),
                             b.y_off.data(), out.data(), ops.data(), ops.size(), &used),
              "Aligner");
        std::vector<Alignment> res;
        for (auto& r : out) res.push_back(detail::to_alignment(r, ops.data()));
        return res;
    }

    Scoring scoring;

protected:
    Aligner(Scoring s, std::shared_ptr<Context> ctx)
        : scoring(std::move(s)), ctx_(ctx ? std::move(ctx) : Context::shared_default()) {}
    Alignment one(AlignmentMode m, const Text& x, const Text& y) { return align_batch(m, {{x, y}})[0]; }
    std::shared_ptr<Context> ctx_;
};

namespace banded {

// banded::Aligner<F> (banded.rs:122-1004): k = k-mer length, w = window (banded.rs:148-150)
class Aligner {
public:
    static Aligner new_(int32_t gap_open, int32_t gap_extend, MatchFn f, size_t k, size_t w,
                        std::shared_ptr<Context> ctx = nullptr) {
        return Aligner(Scoring::new_(gap_open, gap_extend, std::move(f)), k, w, std::move(ctx));
    }
    static Aligner with_scoring(Scoring s, size_t k, size_t w, std::shared_ptr<Context> ctx = nullptr) {
        return Aligner(std::move(s), k, w, std::move(ctx));
    }
    Scoring& get_mut_scoring() { return scoring; }  // banded.rs:272

    Alignment custom(const Text& x, const Text& y) { return one(AlignmentMode::Custom, x, y); }          // banded.rs:282
    Alignment global(const Text& x, const Text& y) { return one(AlignmentMode::Global, x, y); }          // banded.rs:872
    Alignment semiglobal(const Text& x, const Text& y) { return one(AlignmentMode::Semiglobal, x, y); }  // banded.rs:901
    Alignment local(const Text& x, const Text& y) { return one(AlignmentMode::Local, x, y); }            // banded.rs:972

    std::vector<Alignment> align_batch(AlignmentMode mode, const std::vector<std::pair<Text, Text>>& pairs) {
        detail::Batch b;
        for (auto& p : pairs) b.push(p.first, p.second);
        std::vector<int32_t> table;
        const bg_scoring_t sc = scoring.to_c(table);
        std::vector<bg_alignment_t> out(b.size());
        std::vector<uint8_t> ops(b.x.size() + b.y.size() + 4 * b.size() + 8);
        uint64_t used = 0;
        band_cells.assign(b.size(), 0);
        check(bg_align_banded_batch(ctx_->raw(), &sc, (int)mode, (uint32_t)k_, (uint32_t)w_, b.size(), b.x.data(),
                                    b.x_off.data(), b.y.data(), b.y_off.data(), out.data(), ops.data(), ops.size(), &used,
                                    band_cells.data()),
              "banded::Aligner");
        std::vector<Alignment> res;
        for (auto& r : out) res.push_back(detail::to_alignment(r, ops.data()));
        return res;
    }

    // ---- entry points that take matches / chains / a prehash from the caller (banded.rs:294-401, 938-970)
    Alignment custom_with_matches(const Text& x, const Text& y, const std::vector<sparse::Match>& matches) {
        return with_matches(AlignmentMode::Custom, x, y, matches, nullptr);
    }
    Alignment custom_with_match_path(const Text& x, const Text& y, const std::vector<sparse::Match>& matches,
                                     const std::vector<size_t>& path) {
        return with_matches(AlignmentMode::Custom, x, y, matches, &path);
    }
    Alignment custom_with_expanded_matches(const Text& x, const Text& y, std::vector<sparse::Match> matches,
                                           std::optional<size_t> allowed_mismatches, bool use_lcskpp_union) {
        if (allowed_mismatches) matches = sparse::expand_kmer_matches(x, y, k_, matches, *allowed_mismatches);
        if (!use_lcskpp_union) return with_matches(AlignmentMode::Custom, x, y, matches, nullptr);
        const int32_t ms = scoring.match_scores ? scoring.match_scores->first : 2;  // DEFAULT_MATCH_SCORE, banded.rs:105
        const auto path = sparse::sdpkpp_union_lcskpp_path(matches, k_, (uint32_t)ms, scoring.gap_open, scoring.gap_extend);
        return with_matches(AlignmentMode::Custom, x, y, matches, &path);
    }
    Alignment custom_with_prehash(const Text& x, const Text& y, const sparse::KmerHash& y_kmer_hash) {
        return with_matches(AlignmentMode::Custom, x, y, sparse::find_kmer_matches_seq2_hashed(x, y_kmer_hash, k_), nullptr);
    }
    Alignment semiglobal_with_prehash(const Text& x, const Text& y, const sparse::KmerHash& y_kmer_hash) {
        return with_matches(AlignmentMode::Semiglobal, x, y, sparse::find_kmer_matches_seq2_hashed(x, y_kmer_hash, k_), nullptr);
    }

    Scoring scoring;
    std::vector<uint64_t> band_cells;  // Band::num_cells of the last batch

private:
    Aligner(Scoring s, size_t k, size_t w, std::shared_ptr<Context> ctx)
        : scoring(std::move(s)), k_(k), w_(w), ctx_(ctx ? std::move(ctx) : Context::shared_default()) {}
    Alignment one(AlignmentMode m, const Text& x, const Text& y) { return align_batch(m, {{x, y}})[0]; }
    // the band from the caller's matches (host), then compute_alignment on the device
    Alignment with_matches(AlignmentMode mode, const Text& x, const Text& y, const std::vector<sparse::Match>& matches,
                           const std::vector<size_t>* path) {
        std::vector<int32_t> table;
        const bg_scoring_t sc = scoring.to_c(table);
        const uint64_t x_off[2] = {0, x.size()}, y_off[2] = {0, y.size()}, m_off[2] = {0, matches.size()}, b_off[2] = {0, y.size() + 1};
        const auto flat = sparse::detail::flat(matches);
        std::vector<uint32_t> p32;
        uint64_t p_off[2] = {0, 0};
        if (path) {
            p32.assign(path->begin(), path->end());
            p_off[1] = p32.size();
        }
        std::vector<uint32_t> start(y.size() + 1), end(y.size() + 1);
        band_cells.assign(1, 0);
        const int rc = bg_band_from_matches_batch(&sc, (int)mode, (uint32_t)k_, (uint32_t)w_, 1, x_off, y_off, flat.data(), m_off,
                                                  path ? p32.data() : nullptr, path ? p_off : nullptr, b_off, start.data(),
                                                  end.data(), band_cells.data());
        if (rc == BG_ERR_INVALID_ARG) throw Panic("incoming matches must be sorted / path index out of bounds");
        check(rc, "Band::create_with_matches");
        bg_alignment_t out;
        std::vector<uint8_t> ops(x.size() + y.size() + 16);
        uint64_t used = 0;
        check(bg_align_banded_bands_batch(ctx_->raw(), &sc, (int)mode, 1, x.data(), x_off, y.data(), y_off, b_off, start.data(),
                                          end.data(), &out, ops.data(), ops.size(), &used, band_cells.data()),
              "banded::Aligner");
        return detail::to_alignment(out, ops.data());
    }
    size_t k_, w_;
    std::shared_ptr<Context> ctx_;
};

}  // namespace banded
}  // namespace pairwise
}  // namespace alignment

namespace alphabets {

// Alphabet (alphabets/mod.rs:49-60): a set of bytes
struct Alphabet {
    Text symbols;
    explicit Alphabet(const char* s) : symbols(text(s)) {}
    explicit Alphabet(Text s) : symbols(std::move(s)) {}
    bool is_word(const Text& t) const {
        return std::all_of(t.begin(), t.end(), [&](uint8_t c) { return std::find(symbols.begin(), symbols.end(), c) != symbols.end(); });
    }
    uint8_t max_symbol() const { return *std::max_element(symbols.begin(), symbols.end()); }
};
namespace dna {
inline Alphabet alphabet() { return Alphabet("ACGTacgt"); }                            // dna.rs:13-15
inline Alphabet n_alphabet() { return Alphabet("ACGTNacgtn"); }                        // dna.rs:23-25
inline Alphabet iupac_alphabet() { return Alphabet("ACGTRYSWKMBDHVNZacgtryswkmbdhvnz"); }  // dna.rs:33-35
}  // namespace dna
}  // namespace alphabets

namespace data_structures {

namespace suffix_array {
using RawSuffixArray = std::vector<uint64_t>;
// suffix_array(text) (suffix_array.rs:264-284); panics unless the text ends with a unique smallest sentinel
inline RawSuffixArray suffix_array(const Text& t) {
    RawSuffixArray sa(t.size());
    check(bg_suffix_array(t.data(), t.size(), sa.data()), "suffix_array");
    return sa;
}
}  // namespace suffix_array

namespace bwt {
using BWT = Text;
using Less = std::vector<uint64_t>;
inline BWT bwt(const Text& t, const suffix_array::RawSuffixArray& sa) {  // bwt.rs:39-49
    BWT b(t.size());
    check(bg_bwt(t.data(), sa.data(), t.size(), b.data()), "bwt");
    return b;
}
inline Less less(const BWT& b, const alphabets::Alphabet& a) {  // bwt.rs:186-199
    Less l((size_t)a.max_symbol() + 2);
    uint32_t len = (uint32_t)l.size();
    check(bg_less(b.data(), b.size(), a.symbols.data(), (uint32_t)a.symbols.size(), l.data(), &len), "less");
    l.resize(len);
    return l;
}
// Occ (bwt.rs:76-183).  The device index keeps its own (denser) counters; this host object carries the
// sampling rate and alphabet and answers `get` by the definition (count of a in bwt[0..=r]).
struct Occ {
    uint32_t k;
    alphabets::Alphabet alphabet;
    Occ(const BWT&, uint32_t k_, const alphabets::Alphabet& a) : k(k_), alphabet(a) {}
    size_t get(const BWT& b, size_t r, uint8_t a) const { return (size_t)std::count(b.begin(), b.begin() + r + 1, a); }
};
}  // namespace bwt

namespace fmindex {

class FMIndex;

struct Interval {  // fmindex.rs:69-80
    size_t lower = 0, upper = 0;
    bool operator==(const Interval& o) const { return lower == o.lower && upper == o.upper; }
    // Interval::occ over a host suffix array
    std::vector<size_t> occ(const suffix_array::RawSuffixArray& sa) const {
        if (upper > sa.size() && upper > lower) throw Panic("Interval out of range of suffix array");
        std::vector<size_t> v;
        for (size_t p = lower; p < upper; p++) v.push_back((size_t)sa[p]);
        return v;
    }
    // ... or on the device, over the suffix array attached to `fm`
    inline std::vector<size_t> occ(const FMIndex& fm) const;
};

struct BackwardSearchResult {  // fmindex.rs:92-96
    enum Kind { Complete, Partial, Absent } kind = Absent;
    Interval interval;
    size_t matched_len = 0;
    bool operator==(const BackwardSearchResult& o) const {
        return kind == o.kind && (kind == Absent || (interval == o.interval && (kind == Complete || matched_len == o.matched_len)));
    }
};

// FMIndex::new(bwt, less, occ) (fmindex.rs:245-247): uploads the index
class FMIndex {
public:
    FMIndex(const bwt::BWT& b, const bwt::Less& l, const bwt::Occ& occ, std::shared_ptr<Context> ctx = nullptr)
        : ctx_(ctx ? std::move(ctx) : Context::shared_default()), n_(b.size()) {
        check(bg_fm_build(ctx_->raw(), b.data(), b.size(), l.data(), (uint32_t)l.size(), occ.k, occ.alphabet.symbols.data(),
                          (uint32_t)occ.alphabet.symbols.size(), &h_),
              "FMIndex::new");
    }
    ~FMIndex() { bg_fm_free(h_); }
    FMIndex(const FMIndex&) = delete;
    FMIndex& operator=(const FMIndex&) = delete;

    // FMIndexable::backward_search (fmindex.rs:144-208); a byte outside the alphabet panics like the
    // reference's index-out-of-bounds (fmindex.rs:229, bwt.rs:158)
    BackwardSearchResult backward_search(const Text& pattern) const { return backward_search_batch({pattern})[0]; }
    std::vector<BackwardSearchResult> backward_search_batch(const std::vector<Text>& patterns) const {
        Text pat;
        std::vector<uint64_t> off{0};
        for (auto& p : patterns) {
            pat.insert(pat.end(), p.begin(), p.end());
            off.push_back(pat.size());
        }
        const size_t n = patterns.size();
        std::vector<uint8_t> tag(n);
        std::vector<uint64_t> lo(n), hi(n);
        std::vector<uint32_t> ml(n);
        const int rc = bg_fm_backward_search_batch(h_, n, pat.data(), off.data(), tag.data(), lo.data(), hi.data(), ml.data());
        if (rc == BG_ERR_OUT_OF_ALPHABET) throw Panic("index out of bounds: the pattern holds a symbol outside the alphabet");
        check(rc, "backward_search");
        std::vector<BackwardSearchResult> res(n);
        for (size_t q = 0; q < n; q++) {
            if (tag[q] == BG_FM_COMPLETE)
                res[q] = {BackwardSearchResult::Complete, {(size_t)lo[q], (size_t)hi[q]}, ml[q]};
            else if (tag[q] == BG_FM_PARTIAL)
                res[q] = {BackwardSearchResult::Partial, {(size_t)lo[q], (size_t)hi[q]}, ml[q]};
        }
        return res;
    }

    // suffix arrays for Interval::occ on the device
    void attach(const suffix_array::RawSuffixArray& sa) { check(bg_fm_set_suffix_array(h_, sa.data(), sa.size()), "RawSuffixArray"); }
    std::vector<std::vector<size_t>> occ_batch(const std::vector<Interval>& ivs) const {
        std::vector<uint64_t> lo, hi, off(ivs.size() + 1);
        uint64_t total = 0;
        for (auto& v : ivs) {
            lo.push_back(v.lower);
            hi.push_back(v.upper);
            total += v.upper > v.lower ? v.upper - v.lower : 0;
        }
        std::vector<uint64_t> pos(std::max<uint64_t>(total, 1));
        const int rc = bg_interval_occ_batch(h_, ivs.size(), lo.data(), hi.data(), off.data(), pos.data(), total);
        if (rc == BG_ERR_INVALID_ARG) throw Panic("Interval out of range of suffix array");  // fmindex.rs:77
        check(rc, "Interval::occ");
        std::vector<std::vector<size_t>> res(ivs.size());
        for (size_t v = 0; v < ivs.size(); v++) res[v].assign(pos.begin() + off[v], pos.begin() + off[v + 1]);
        return res;
    }
    std::vector<std::optional<size_t>> sa_get_batch(const std::vector<uint64_t>& rows) const {
        std::vector<uint64_t> pos(rows.size());
        check(bg_sa_get_batch(h_, rows.size(), rows.data(), pos.data()), "SuffixArray::get");
        std::vector<std::optional<size_t>> res(rows.size());
        for (size_t i = 0; i < rows.size(); i++)
            if (pos[i] != BG_SA_NONE) res[i] = (size_t)pos[i];
        return res;
    }
    // Seed-and-extend in one call (bg_seed_extend_batch): the loop rust-bio's callers write from backward_search,
    // Interval::occ and Aligner::semiglobal (src/lib.rs:129-165, benches/fmindex.rs:20-38); definition in biogpu.h.
    // Needs attach_text() and a suffix array.
    void attach_text(const Text& text) { check(bg_fm_set_text(h_, text.data(), text.size()), "bg_fm_set_text"); }
    struct SeedHit {
        std::optional<alignment::Alignment> alignment;  // Aligner::semiglobal(read, window) of the best candidate
        size_t ref_start = 0, ref_end = 0;              // text coordinates of its y span
        uint32_t n_candidates = 0;
    };
    std::vector<SeedHit> seed_extend_batch(const alignment::pairwise::Scoring& scoring, const std::vector<Text>& reads, uint32_t seed_len = 20,
                                           uint32_t stride = 10, uint32_t max_occ = 16, uint32_t pad = 25) const {
        std::vector<int32_t> table;
        const bg_scoring_t sc = scoring.to_c(table);
        const bg_seed_params_t prm = {seed_len, stride, max_occ, pad};
        Text buf;
        std::vector<uint64_t> off{0};
        for (auto& r : reads) {
            buf.insert(buf.end(), r.begin(), r.end());
            off.push_back(buf.size());
        }
        std::vector<bg_seed_hit_t> hits(reads.size());
        std::vector<uint8_t> ops(2 * buf.size() + (2 * (size_t)pad + 4) * reads.size() + 8);
        uint64_t used = 0;
        const int rc = bg_seed_extend_batch(h_, &sc, &prm, reads.size(), buf.data(), off.data(), hits.data(), ops.data(), ops.size(), &used);
        if (rc == BG_ERR_OUT_OF_ALPHABET) throw Panic("index out of bounds: a seed holds a byte outside the index's alphabet");
        check(rc, "bg_seed_extend_batch");
        std::vector<SeedHit> res(reads.size());
        for (size_t r = 0; r < reads.size(); r++) {
            if (hits[r].aln.score != BG_MIN_SCORE) res[r].alignment = alignment::pairwise::detail::to_alignment(hits[r].aln, ops.data());
            res[r].ref_start = (size_t)hits[r].ref_start;
            res[r].ref_end = (size_t)hits[r].ref_end;
            res[r].n_candidates = hits[r].n_candidates;
        }
        return res;
    }
    bg_fm* raw() const { return h_; }
    size_t len() const { return n_; }

    // Serialize / Deserialize (the reference derives them for FMIndex, Occ, SampledSuffixArray: fmindex.rs:214, bwt.rs:76,
    // suffix_array.rs:124): bg_fm_save / bg_fm_load — BWT, less, alphabet, k, the attached suffix array, an owned text.
    void save(const std::string& path) const { check(bg_fm_save(h_, path.c_str()), "FMIndex::serialize"); }
    static std::unique_ptr<FMIndex> load(const std::string& path, std::shared_ptr<Context> ctx = nullptr) {
        std::unique_ptr<FMIndex> fm(new FMIndex(ctx ? std::move(ctx) : Context::shared_default()));
        check(bg_fm_load(fm->ctx_->raw(), path.c_str(), &fm->h_), "FMIndex::deserialize");
        uint64_t n = 0;
        check(bg_fm_len(fm->h_, &n), "bg_fm_len");
        fm->n_ = (size_t)n;
        return fm;
    }
    // the BWT the index was built over, read back out of the handle (bg_fm_bwt): a deserialized FMIndex has no other
    bwt::BWT bwt() const {
        bwt::BWT b(n_);
        check(bg_fm_bwt(h_, b.data()), "bg_fm_bwt");
        return b;
    }

private:
    explicit FMIndex(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)), n_(0) {}
    std::shared_ptr<Context> ctx_;
    bg_fm* h_ = nullptr;
    size_t n_;
};

inline std::vector<size_t> Interval::occ(const FMIndex& fm) const { return fm.occ_batch({*this})[0]; }

// BiInterval (fmindex.rs:250-283)
struct BiInterval {
    size_t lower = 0, lower_rev = 0, size = 0, match_size = 0;
    Interval forward() const { return {lower, lower + size}; }
    Interval revcomp() const { return {lower_rev, lower_rev + size}; }
    bool operator==(const BiInterval& o) const {
        return lower == o.lower && lower_rev == o.lower_rev && size == o.size && match_size == o.match_size;
    }
};

// FMDIndex::from(fmindex) (fmindex.rs:311-329); smems / all_smems (363-501) run on the device
class FMDIndex {
public:
    struct Smem {  // the reference's tuple (BiInterval, position on the pattern, SMEM length)
        BiInterval interval;
        size_t position, length;
    };
    explicit FMDIndex(const FMIndex& fm, const bwt::BWT& b) : fm_(&fm) { check_alphabet(b); }
    // FMDIndex::from(fmindex) proper: the BWT is the index's own (also for one that came from FMIndex::load)
    explicit FMDIndex(const FMIndex& fm) : fm_(&fm) { check_alphabet(fm.bwt()); }
    std::vector<Smem> smems(const Text& pattern, size_t i, size_t l) const { return run({pattern}, {(uint32_t)i}, l, false)[0]; }
    std::vector<Smem> all_smems(const Text& pattern, size_t l) const { return run({pattern}, {}, l, true)[0]; }
    std::vector<std::vector<Smem>> smems_batch(const std::vector<Text>& patterns, const std::vector<uint32_t>& positions,
                                               size_t l) const {
        return run(patterns, positions, l, false);
    }
    std::vector<std::vector<Smem>> all_smems_batch(const std::vector<Text>& patterns, size_t l) const {
        return run(patterns, {}, l, true);
    }

private:
    static void check_alphabet(const bwt::BWT& b) {  // fmindex.rs:323-327
        for (uint8_t c : b)
            if (!c || !std::char_traits<char>::find("ACGTNacgtn$", 11, (char)c))
                throw Panic("Expecting BWT over the DNA alphabet (including N) with the sentinel $.");
    }
    std::vector<std::vector<Smem>> run(const std::vector<Text>& patterns, const std::vector<uint32_t>& positions, size_t l,
                                       bool all) const {
        Text pat;
        std::vector<uint64_t> off{0};
        size_t cap = 1;
        for (auto& p : patterns) {
            pat.insert(pat.end(), p.begin(), p.end());
            off.push_back(pat.size());
            cap = std::max(cap, p.size() + 1);
        }
        const size_t n = patterns.size();
        std::vector<uint32_t> count(n);
        std::vector<uint64_t> out(n * cap * 6);  // (usize records, fmindex.rs:254-259: the flavour both index layouts answer)
        const int rc = bg_fmd_smems_batch64(fm_->raw(), all ? 1 : 0, n, pat.data(), off.data(), all ? nullptr : positions.data(),
                                            (uint32_t)l, (uint32_t)cap, count.data(), out.data());
        if (rc == BG_ERR_OUT_OF_ALPHABET) throw Panic("index out of bounds");
        check(rc, "FMDIndex::smems");
        std::vector<std::vector<Smem>> res(n);
        for (size_t q = 0; q < n; q++)
            for (uint32_t t = 0; t < count[q]; t++) {
                const uint64_t* r = &out[(q * cap + t) * 6];
                res[q].push_back({{(size_t)r[0], (size_t)r[1], (size_t)r[2], (size_t)r[3]}, (size_t)r[4], (size_t)r[5]});
            }
        return res;
    }
    const FMIndex* fm_;
};

}  // namespace fmindex

namespace suffix_array {

// RawSuffixArray::sample (suffix_array.rs:86-120) + SampledSuffixArray::get (157-184, on the device)
class SampledSuffixArray {
public:
    SampledSuffixArray(const RawSuffixArray& sa, const Text& t, const bwt::BWT& b, size_t sampling_rate, fmindex::FMIndex& fm)
        : s_(sampling_rate), fm_(&fm), n_(sa.size()) {
        const uint8_t sentinel = t.back();
        for (size_t i = 0; i < sa.size(); i++) {
            if (i % s_ == 0)
                sample_.push_back(sa[i]);
            else if (b[i] == sentinel) {
                extra_rows_.push_back(i);
                extra_pos_.push_back(sa[i]);
            }
        }
        check(bg_fm_set_sampled_suffix_array(fm.raw(), sample_.data(), sample_.size(), (uint32_t)s_, sentinel, extra_rows_.data(),
                                             extra_pos_.data(), extra_rows_.size()),
              "SampledSuffixArray");
    }
    std::optional<size_t> get(size_t index) const { return fm_->sa_get_batch({(uint64_t)index})[0]; }
    size_t len() const { return n_; }
    size_t sampling_rate() const { return s_; }

private:
    size_t s_;
    fmindex::FMIndex* fm_;
    size_t n_;
    std::vector<uint64_t> sample_, extra_rows_, extra_pos_;
};

inline SampledSuffixArray sample(const RawSuffixArray& sa, const Text& t, const bwt::BWT& b, size_t sampling_rate,
                                 fmindex::FMIndex& fm) {
    return SampledSuffixArray(sa, t, b, sampling_rate, fm);
}

}  // namespace suffix_array
}  // namespace data_structures

// bio::io::fastq, reading side (io/fastq.rs:153-527), over a text held in memory; parsed on the device
namespace io {
namespace fastq {
struct ReadError : Panic {  // fastq.rs:113-126
    enum Kind { MissingAt = BG_FASTQ_MISSING_AT, IncompleteRecord = BG_FASTQ_INCOMPLETE, Io = BG_FASTQ_IO } kind;
    uint64_t pos;
    ReadError(Kind k, uint64_t p) : Panic(k == MissingAt ? "MissingAt" : k == IncompleteRecord ? "IncompleteRecord" : "Io"), kind(k), pos(p) {}
};
enum class CheckError { Ok = 0, EmptyId, NonAsciiSequence, InvalidSequence, NonAsciiQualities, UnequalLength };  // fastq.rs:129-150

class Record {  // fastq.rs:309-452
public:
    Record() = default;
    Record(std::string id, std::optional<std::string> desc, Text seq, Text qual, CheckError chk = CheckError::Ok)
        : id_(std::move(id)), desc_(std::move(desc)), seq_(std::move(seq)), qual_(std::move(qual)), check_(chk) {}
    bool is_empty() const { return id_.empty() && !desc_ && seq_.empty() && qual_.empty(); }
    CheckError check() const { return check_; }  // Ok(()) == CheckError::Ok; evaluated on the device with the parse
    const std::string& id() const { return id_; }
    const std::optional<std::string>& desc() const { return desc_; }
    const Text& seq() const { return seq_; }
    const Text& qual() const { return qual_; }
    bool operator==(const Record& o) const { return id_ == o.id_ && desc_ == o.desc_ && seq_ == o.seq_ && qual_ == o.qual_; }

private:
    std::string id_;
    std::optional<std::string> desc_;
    Text seq_, qual_;
    CheckError check_ = CheckError::Ok;
};

// `Reader::new(&[u8])` + `read()` / `records()`: all records are parsed by one device call; read() hands them
// out one by one, an empty Record at the end (fastq.rs:228), ReadError where the reference returns Err
class Reader {
public:
    explicit Reader(const Text& t, std::shared_ptr<Context> ctx = nullptr) {
        if (!ctx) ctx = Context::shared_default();
        const uint64_t cap = t.size() / 4 + 2;
        std::vector<bg_fastq_record_t> recs(cap);
        std::vector<uint8_t> seq(t.size() + 1), qual(t.size() + 1);
        std::vector<uint64_t> so(cap + 1), qo(cap + 1);
        uint64_t n = 0, err = 0;
        int32_t st = 0;
        check(bg_fastq_parse(ctx->raw(), t.data(), t.size(), recs.data(), cap, seq.data(), so.data(), qual.data(), qo.data(), &n, &st, &err),
              "bg_fastq_parse");
        for (uint64_t k = 0; k < n; k++) {
            const bg_fastq_record_t& r = recs[k];
            std::optional<std::string> d;
            if (r.has_desc) d = std::string(t.begin() + r.desc_off, t.begin() + r.desc_off + r.desc_len);
            records_.emplace_back(std::string(t.begin() + r.id_off, t.begin() + r.id_off + r.id_len), std::move(d),
                                  Text(seq.begin() + so[k], seq.begin() + so[k + 1]), Text(qual.begin() + qo[k], qual.begin() + qo[k + 1]),
                                  (CheckError)r.check);
        }
        status_ = st;
        err_pos_ = err;
    }
    void read(Record& record) {
        if (next_ < records_.size()) {
            record = records_[next_++];
        } else if (status_ != BG_FASTQ_OK && !raised_) {
            raised_ = true;
            throw ReadError((ReadError::Kind)status_, err_pos_);
        } else {
            record = Record();
        }
    }
    // the records read before the first error; `status()` tells whether the iterator would end with Err
    const std::vector<Record>& records() const { return records_; }
    int status() const { return status_; }

private:
    std::vector<Record> records_;
    size_t next_ = 0;
    int status_ = 0;
    uint64_t err_pos_ = 0;
    bool raised_ = false;
};
}  // namespace fastq
}  // namespace io
}  // namespace bio
#endif
