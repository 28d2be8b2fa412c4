// The reference's known-answer tests, written against the C++ host mirror (include/biogpu.hpp) the way
// the reference writes them against its Rust API — plus the panics the reference asserts on.
// Needs a gfx950 device (the engine has no CPU fallback).  Usage: run_kats [filter-substring]
#include <cstdio>
#include <cstring>
#include <sstream>

#include "biogpu.hpp"

using namespace bio;
using namespace bio::alignment;
using namespace bio::alignment::pairwise;
using namespace bio::data_structures;

static int g_failed = 0;
static const char* g_current = "";
#define CHECK(cond)                                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            std::fprintf(stderr, "FAIL %s: %s (%s:%d)\n", g_current, #cond, __FILE__, __LINE__); \
            g_failed++;                                                                 \
        }                                                                               \
    } while (0)
#define CHECK_EQ(a, b)                                                                  \
    do {                                                                                \
        const auto va = (a);                                                            \
        const auto vb = (b);                                                            \
        if (!(va == vb)) {                                                              \
            std::ostringstream os;                                                      \
            os << va << " != " << vb;                                                   \
            std::fprintf(stderr, "FAIL %s: %s == %s: %s (%s:%d)\n", g_current, #a, #b, os.str().c_str(), __FILE__, __LINE__); \
            g_failed++;                                                                 \
        }                                                                               \
    } while (0)
static std::string ops_str(const std::vector<AlignmentOperation>& v) {
    static const char* n[] = {"M", "S", "D", "I", "X", "Y"};
    std::string s;
    for (auto& o : v) {
        s += n[o.kind];
        if (o.kind >= AlignmentOperation::Xclip) s += std::to_string(o.len);
        s += ' ';
    }
    return s;
}
#define CHECK_OPS(a, b)                                                                 \
    do {                                                                                \
        if (!((a) == (b))) {                                                            \
            std::fprintf(stderr, "FAIL %s: operations [%s] != [%s] (%s:%d)\n", g_current, ops_str(a).c_str(), ops_str(b).c_str(), __FILE__, __LINE__); \
            g_failed++;                                                                 \
        }                                                                               \
    } while (0)
template <typename F>
static bool panics(F&& f) {
    try {
        f();
    } catch (const Panic&) {
        return true;
    }
    return false;
}

static int32_t blosum62(uint8_t a, uint8_t b);
#include "kats_generated.inc"

// scores::blosum62 (src/scores/blosum62.rs) through the pairs the golden fixture holds
static int32_t blosum62(uint8_t a, uint8_t b) {
    static int8_t tab[256][256];
    static bool init = false;
    if (!init) {
        std::memset(tab, 0, sizeof(tab));
        for (auto& p : kBlosum62Pairs) tab[(uint8_t)p.a][(uint8_t)p.b] = p.v;
        init = true;
    }
    return tab[a][b];
}

// ---- the reference's asserts (mod.rs:265-266, 322 ff.; suffix_array.rs:431-437; fmindex.rs:229)
static void kat_panics() {
    CHECK(panics([] { Scoring::from_scores(1, -1, 1, -1); }));   // gap_open can't be positive
    CHECK(panics([] { Scoring::from_scores(-5, 1, 1, -1); }));   // gap_extend can't be positive
    CHECK(panics([] { Scoring::from_scores(-5, -1, 1, -1).xclip(1); }));
    CHECK(panics([] { suffix_array::suffix_array(text("ACGT")); }));   // no sentinel
    CHECK(panics([] { suffix_array::suffix_array(text("AC#GT$")); }));  // '#' < '$'
    CHECK(panics([] {  // pattern byte outside the alphabet reaches Occ::get
        const Text t = text("GCCTTAACATTATTACGCCTA$");
        const auto alphabet = alphabets::dna::n_alphabet();
        const auto sa = suffix_array::suffix_array(t);
        const auto b = bwt::bwt(t, sa);
        fmindex::FMIndex fm(b, bwt::less(b, alphabet), bwt::Occ(b, 3, alphabet));
        fm.backward_search(text("TT~"));
    }));
    CHECK(panics([] {  // Interval out of range of suffix array (fmindex.rs:77)
        const Text t = text("GCCTTAACATTATTACGCCTA$");
        const auto sa = suffix_array::suffix_array(t);
        fmindex::Interval{0, sa.size() + 1}.occ(sa);
    }));
}

// ---- batch entry points: a batch equals its single calls
static void kat_batches_equal_single_calls() {
    auto aligner = pairwise::Aligner::with_scoring(Scoring::from_scores(-5, -1, 1, -1));
    std::vector<std::pair<Text, Text>> pairs = {{text("ACCGTGGAT"), text("AAAAACCGTTGAT")},
                                                {text("ACGTACGT"), text("ACGGTACGT")},
                                                {text(""), text("ACGT")},
                                                {text("TTTT"), text("")}};
    const auto batch = aligner.align_batch(AlignmentMode::Local, pairs);
    for (size_t p = 0; p < pairs.size(); p++) CHECK(batch[p] == aligner.local(pairs[p].first, pairs[p].second));
}

// ---- banded entry points that take matches / a prehash (banded.rs:294-401; doctest banded.rs:46-54, test 1464-1466)
static void kat_banded_with_matches_and_prehash() {
    const Text x = text("AGCACACGTGTGCGCTATACAGTAAGTAGTAGTACACGTGTCACAGTTGTACTAGCATGAC");
    const Text y = text("AGCACACGTGTGCGCTATACAGTACACGTGTCACAGTTGTACTAGCATGAC");
    auto score = [](uint8_t a, uint8_t b) { return a == b ? 1 : -1; };
    const size_t k = 8, w = 6;
    auto aligner = pairwise::banded::Aligner::new_(-5, -1, score, k, w);
    const auto y_kmers_hash = sparse::hash_kmers(y, k);
    CHECK(aligner.semiglobal_with_prehash(x, y, y_kmers_hash) == aligner.semiglobal(x, y));
    CHECK(aligner.custom_with_prehash(x, y, y_kmers_hash) == aligner.custom(x, y));
    const auto matches = sparse::find_kmer_matches(x, y, k);
    CHECK(aligner.custom_with_matches(x, y, matches) == aligner.custom(x, y));
    // no matches -> the full matrix (banded.rs:1309-1313; what the fuzz target compares against)
    auto full = pairwise::Aligner::new_(-5, -1, score);
    CHECK(aligner.custom_with_matches(x, y, {}) == full.custom(x, y));
    CHECK(aligner.custom_with_expanded_matches(x, y, matches, std::optional<size_t>(1), true).score ==
          aligner.custom_with_expanded_matches(x, y, matches, std::nullopt, false).score);
    CHECK(panics([&] { aligner.custom_with_matches(x, y, {{5, 5}, {1, 1}}); }));  // unsorted
}

// ---- FMDIndex (fmindex.rs:704-780: test_smems, test_all_smems)
static void kat_fmdindex_smems() {
    const Text orig = text("GCCTTAACAT"), t = text("GCCTTAACAT$ATGTTAAGGC$");
    const auto alphabet = alphabets::dna::n_alphabet();
    const auto sa = suffix_array::suffix_array(t);
    const auto b = bwt::bwt(t, sa);
    const auto less = bwt::less(b, alphabet);
    const bwt::Occ occ(b, 3, alphabet);
    fmindex::FMIndex fmindex(b, less, occ);
    fmindex::FMDIndex fmdindex(fmindex, b);
    {
        const auto intervals = fmdindex.smems(text("AA"), 0, 0);
        CHECK(intervals[0].interval.forward().occ(sa) == (std::vector<size_t>{5, 16}));
        CHECK(intervals[0].interval.revcomp().occ(sa) == (std::vector<size_t>{3, 14}));
        CHECK_EQ(intervals[0].position, (size_t)0);
        CHECK_EQ(intervals[0].length, (size_t)2);
    }
    {
        const auto intervals = fmdindex.smems(text("CTTAA"), 1, 0);
        CHECK(intervals[0].interval.forward().occ(sa) == (std::vector<size_t>{2}));
        CHECK(intervals[0].interval.revcomp().occ(sa) == (std::vector<size_t>{14}));
        CHECK_EQ(intervals[0].position, (size_t)0);
        CHECK_EQ(intervals[0].length, (size_t)5);
        CHECK_EQ(intervals[0].interval.match_size, (size_t)5);
    }
    CHECK(fmdindex.smems(text("CTTAA"), 1, 7).empty());
    {
        const Text t2 = text("ATTCGGGG$CCCCGAAT$");
        const auto sa2 = suffix_array::suffix_array(t2);
        const auto b2 = bwt::bwt(t2, sa2);
        fmindex::FMIndex fm2(b2, bwt::less(b2, alphabet), bwt::Occ(b2, 3, alphabet));
        fmindex::FMDIndex fmd2(fm2, b2);
        const auto intervals = fmd2.all_smems(text("ATTGGGG"), 0);
        CHECK_EQ(intervals.size(), (size_t)2);
        const size_t solutions[2][4] = {{0, 14, 0, 3}, {4, 9, 3, 4}};
        for (size_t i = 0; i < intervals.size() && i < 2; i++) {
            CHECK_EQ(intervals[i].interval.forward().occ(sa2)[0], solutions[i][0]);
            CHECK_EQ(intervals[i].interval.revcomp().occ(sa2)[0], solutions[i][1]);
            CHECK_EQ(intervals[i].position, solutions[i][2]);
            CHECK_EQ(intervals[i].length, solutions[i][3]);
        }
    }
}

// io/fastq.rs:606-866 and the bio-types cigar documentation example
static void kat_fastq_reader_and_cigar() {
    using namespace bio::io;
    {
        fastq::Reader reader(text("@id desc\nACCGTAGGCTGA\n+\nIIIIIIJJJJJJ\n"));
        CHECK_EQ(reader.records().size(), (size_t)1);
        const auto& record = reader.records()[0];
        CHECK(record.check() == fastq::CheckError::Ok);
        CHECK_EQ(record.id(), std::string("id"));
        CHECK_EQ(*record.desc(), std::string("desc"));
        CHECK(record.seq() == text("ACCGTAGGCTGA"));
        CHECK(record.qual() == text("IIIIIIJJJJJJ"));
    }
    {  // test_read_header_does_not_start_with_correct_char_raises_err / test_read_quality_is_empty_raises_err
        fastq::Reader r1(text(">id description\nACGT\n+\n!!!!\n"));
        fastq::Record rec;
        bool missing_at = false, incomplete = false;
        try { r1.read(rec); } catch (const fastq::ReadError& e) { missing_at = e.kind == fastq::ReadError::MissingAt; }
        CHECK(missing_at);
        fastq::Reader r2(text("@id description\nACGT\n+\n"));
        try { r2.read(rec); } catch (const fastq::ReadError& e) { incomplete = e.kind == fastq::ReadError::IncompleteRecord; }
        CHECK(incomplete);
    }
    {  // test_read_sequence_and_quality_are_wrapped_is_handled_with_three_sequences
        fastq::Reader reader(text("@id description\nACGT\nGGGG\nC\n+\n@@@@\n!!!!\n$\n@id2 description\nACGT\nGGGG\nC\n+\n@@@@\n!!!!\n$\n"
                                  "@id3 desc1 desc2\nAAA\nAAA\nAA\n+\n^^^\n^^^\n^^\n"));
        fastq::Record actual;
        reader.read(actual);
        CHECK(actual == fastq::Record("id", std::string("description"), text("ACGTGGGGC"), text("@@@@!!!!$")));
        reader.read(actual);
        CHECK(actual == fastq::Record("id2", std::string("description"), text("ACGTGGGGC"), text("@@@@!!!!$")));
        reader.read(actual);
        CHECK(actual == fastq::Record("id3", std::string("desc1 desc2"), text("AAAAAAAA"), text("^^^^^^^^")));
        reader.read(actual);
        CHECK(actual.is_empty());
    }
    {  // test_read_wrapped_record_with_inconsistent_wrapping_errors
        fastq::Reader reader(text("@id description\nACGT\nGGGG\nC\n+\n@@@@\n!!!!$\n@id2 description\nACGT\nGGGG\nC\n+\n@@@@\n!!!!\n$\n"));
        fastq::Record record;
        reader.read(record);
        bool missing_at = false;
        try { reader.read(record); } catch (const fastq::ReadError& e) { missing_at = e.kind == fastq::ReadError::MissingAt; }
        CHECK(missing_at);
    }
    {  // Record::check (fastq.rs:731-788)
        CHECK(fastq::Reader(text("@\nACGT\n+\n!!!!\n")).records()[0].check() == fastq::CheckError::EmptyId);
        CHECK(fastq::Reader(text("@id\nATGC1234\n+\nQQQQQQQQ\n")).records()[0].check() == fastq::CheckError::InvalidSequence);
        CHECK(fastq::Reader(text("@id\nATGC\n+\nQQ\n")).records()[0].check() == fastq::CheckError::UnequalLength);
        CHECK(fastq::Reader(text("@id_str desc\nATGCGGG\n+\nQQQQQQQ\n")).records()[0].check() == fastq::CheckError::Ok);
    }
    {  // bio-types: Alignment::cigar documentation example
        Alignment alignment;
        alignment.score = 5;
        alignment.xstart = 3, alignment.xend = 9, alignment.ystart = 0, alignment.yend = 10, alignment.ylen = 10, alignment.xlen = 10;
        alignment.operations = {Match, Match, Match, Subst, Ins, Ins, Del, Del};
        alignment.mode = AlignmentMode::Semiglobal;
        CHECK_EQ(alignment.cigar(false), std::string("3S3=1X2I2D1S"));
        CHECK_EQ(alignment.cigar(true), std::string("3H3=1X2I2D1H"));
        alignment.mode = AlignmentMode::Custom;
        CHECK(panics([&] { alignment.cigar(false); }));
    }
}

// Alignment::pretty on the bio-types documentation pair, and the seed-and-extend loop of src/lib.rs:129-165 in one call
static void kat_pretty_and_seed_extend() {
    using namespace bio::alignment;
    const Text x = text("CCGTCCGGCAAGGG"), y = text("AAAAACCGTTGACGGCCAA");
    auto aligner = pairwise::Aligner::new_(-5, -1, [](uint8_t a, uint8_t b) { return a == b ? 1 : -1; });
    CHECK_EQ(aligner.local(x, y).pretty(x, y, 100), std::string("     CCGTCCGGCAAGGG          \n"
                                                                "     ||||                    \n"
                                                                "AAAAACCGT          TGACGGCCAA\n\n\n"));
    CHECK_EQ(aligner.global(x, y).pretty(x, y, 100), std::string("-----CCGTCCGGCAAGGG\n"
                                                                 "xxxxx||||\\\\\\\\\\\\\\\\\\\\\n"
                                                                 "AAAAACCGTTGACGGCCAA\n\n\n"));
    {
        // a pseudo-random text without long repeats; reads are substrings (one with a substitution) or foreign
        Text g;
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 4000; i++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            g.push_back("ACGT"[(s >> 33) & 3]);
        }
        Text t = g;
        t.push_back('$');
        const auto alphabet = alphabets::dna::n_alphabet();
        const auto sa = suffix_array::suffix_array(t);
        const auto b = bwt::bwt(t, sa);
        fmindex::FMIndex fm(b, bwt::less(b, alphabet), bwt::Occ(b, 3, alphabet));
        fm.attach(sa);
        fm.attach_text(t);
        Text r0(g.begin() + 1000, g.begin() + 1100), r1(g.begin() + 2500, g.begin() + 2600), r2(100, 'A');
        r1[50] = r1[50] == 'A' ? 'C' : 'A';
        for (size_t i = 0; i < r2.size(); i += 2) r2[i] = 'C';
        const auto scoring = pairwise::Scoring::from_scores(-5, -1, 1, -1);
        const auto hits = fm.seed_extend_batch(scoring, {r0, r1, r2});
        CHECK_EQ(hits.size(), (size_t)3);
        CHECK(hits[0].alignment && hits[0].alignment->score == 100);
        CHECK_EQ(hits[0].ref_start, (size_t)1000);
        CHECK_EQ(hits[0].ref_end, (size_t)1100);
        CHECK(hits[1].alignment && hits[1].alignment->score == 98);
        CHECK_EQ(hits[1].ref_start, (size_t)2500);
        CHECK(!hits[2].alignment && hits[2].n_candidates == 0);
        // the same placement through the reference's own loop: backward_search -> occ -> semiglobal on the window
        const auto iv = fm.backward_search(Text(r0.begin(), r0.begin() + 20));
        CHECK(iv.kind == fmindex::BackwardSearchResult::Complete);
        const auto pos = iv.interval.occ(sa);
        CHECK(pos == (std::vector<size_t>{1000}));
        auto semi = pairwise::Aligner::with_scoring(scoring);
        const Text window(g.begin() + 975, g.begin() + 1125);
        const auto a = semi.semiglobal(r0, window);
        CHECK_EQ(a.score, hits[0].alignment->score);
        CHECK(a.operations == hits[0].alignment->operations);
        CHECK_EQ(975 + a.ystart, hits[0].ref_start);
    }
}

// ---- Serialize / Deserialize (fmindex.rs:214, bwt.rs:76, suffix_array.rs:124: derived in the reference): a saved and
// reloaded index answers backward_search, Interval::occ and seed-and-extend like the one that was saved
static void kat_fmindex_save_and_load() {
    Text g;
    uint64_t s = 0x2545F4914F6CDD1Dull;
    for (int i = 0; i < 6000; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        g.push_back("ACGT"[(s >> 33) & 3]);
    }
    g[1234] = 'N';
    Text t = g;
    t.push_back('$');
    const auto alphabet = alphabets::dna::n_alphabet();
    const auto sa = suffix_array::suffix_array(t);
    const auto b = bwt::bwt(t, sa);
    fmindex::FMIndex fm(b, bwt::less(b, alphabet), bwt::Occ(b, 3, alphabet));
    fm.attach(sa);
    fm.attach_text(t);
    const std::string path = "/tmp/biogpu_kat_index.bgfm";
    fm.save(path);
    const auto fm2 = fmindex::FMIndex::load(path);
    std::vector<Text> pats;
    for (int q = 0; q < 200; q++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const size_t at = (s >> 20) % 5900, len = 1 + (s >> 50) % 40;
        Text p(g.begin() + at, g.begin() + at + len);
        if (q % 3 == 0) p[len / 2] = p[len / 2] == 'A' ? 'C' : 'A';
        pats.push_back(p);
    }
    const auto r1 = fm.backward_search_batch(pats), r2 = fm2->backward_search_batch(pats);
    CHECK(r1 == r2);
    std::vector<fmindex::Interval> ivs;
    for (auto& r : r2)
        if (r.kind == fmindex::BackwardSearchResult::Complete) ivs.push_back(r.interval);
    CHECK(!ivs.empty());
    CHECK(fm.occ_batch(ivs) == fm2->occ_batch(ivs));
    for (auto& iv : ivs) CHECK(iv.occ(*fm2) == iv.occ(sa));
    const auto scoring = pairwise::Scoring::from_scores(-5, -1, 1, -1);
    const Text r0(g.begin() + 2000, g.begin() + 2100);
    const auto h1 = fm.seed_extend_batch(scoring, {r0}), h2 = fm2->seed_extend_batch(scoring, {r0});
    CHECK(h2[0].alignment && h1[0].alignment && h2[0].alignment->score == h1[0].alignment->score);
    CHECK_EQ(h2[0].ref_start, (size_t)2000);
    CHECK(panics([] { fmindex::FMIndex::load("/tmp/biogpu_kat_index.missing"); }));
    // the loaded handle knows its length and its BWT (bg_fm_len / bg_fm_bwt): FMDIndex::from works on it when the text is
    // DNA + '$' (fmindex.rs:311-329)
    CHECK_EQ(fm2->len(), fm.len());
    CHECK(fm2->bwt() == b);
    fmindex::FMDIndex fmd(*fm2);
    (void)fmd.all_smems(Text(g.begin() + 100, g.begin() + 160), 10);  // (not T$R$: only that the call goes through)
    std::remove(path.c_str());
}

int main(int argc, char** argv) {
    const char* filter = argc > 1 ? argv[1] : "";
    int ran = 0;
    auto run = [&](const char* name, void (*fn)()) {
        if (*filter && !std::strstr(name, filter)) return;
        g_current = name;
        const int before = g_failed;
        try {
            fn();
        } catch (const std::exception& e) {
            std::fprintf(stderr, "FAIL %s: unexpected panic: %s\n", name, e.what());
            g_failed++;
        }
        ran++;
        if (g_failed != before) std::fprintf(stderr, "  ^ in %s\n", name);
    };
    for (auto& k : kKats) run(k.name, k.fn);
    run("kat_panics", kat_panics);
    run("kat_batches_equal_single_calls", kat_batches_equal_single_calls);
    run("kat_banded_with_matches_and_prehash", kat_banded_with_matches_and_prehash);
    run("kat_fmdindex_smems", kat_fmdindex_smems);
    run("kat_fastq_reader_and_cigar", kat_fastq_reader_and_cigar);
    run("kat_pretty_and_seed_extend", kat_pretty_and_seed_extend);
    run("kat_fmindex_save_and_load", kat_fmindex_save_and_load);
    std::printf("%d tests, %d failed\n", ran, g_failed);
    return g_failed ? 1 : 0;
}
