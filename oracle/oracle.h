/*
 * oracle.h — C interface of the CPU ORACLE.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a line-faithful CPU
 * restatement (C++17, g++) of rust-bio 4.0.1's algorithms on the hot path
 * (pairwise::Aligner, pairwise::banded::Aligner, sparse::sdpkpp, bwt::{bwt,less,Occ},
 * fmindex::backward_search, suffix_array).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it — as the checker / the timed CPU baseline,
 * never as the thing shipped.  Nothing under rust-bio_amd/ links or calls it.
 *
 * Parity pinning: rust-bio itself cannot be compiled in this environment (no rustc /
 * cargo, un-vendored dependencies), so the oracle is pinned against every known-answer
 * test the reference holds for this path (tests/golden/ JSON files, transcribed from the
 * reference's own #[test]s and doctests with file:line citations; see
 * tests/test_oracle_*.py).
 */
#ifndef BIOGPU_ORACLE_H
#define BIOGPU_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MIN_SCORE (-858993459) /* pairwise/mod.rs:174 */

/* AlignmentMode values (bio-types) used in the result record */
enum { ORC_MODE_CUSTOM = 0, ORC_MODE_GLOBAL = 1, ORC_MODE_SEMIGLOBAL = 2, ORC_MODE_LOCAL = 3 };

/* AlignmentOperation kinds (bio-types); Xclip/Yclip carry a length */
enum { ORC_OP_MATCH = 0, ORC_OP_SUBST = 1, ORC_OP_DEL = 2, ORC_OP_INS = 3, ORC_OP_XCLIP = 4, ORC_OP_YCLIP = 5 };

/* Scoring<F> (pairwise/mod.rs:238-247). match_fn is either MatchParams
 * (matrix == NULL) or a tabulated closure matrix[a*256+b]. match_scores_some mirrors
 * `match_scores: Option<(i32,i32)>` which only the banded aligner looks at. */
typedef struct {
    int32_t gap_open, gap_extend;
    int32_t xclip_prefix, xclip_suffix, yclip_prefix, yclip_suffix;
    int32_t match_score, mismatch_score;
    int32_t match_scores_some;
    const int32_t* matrix;
} orc_scoring_t;

/* bio_types::alignment::Alignment (constructed at pairwise/mod.rs:911-921) */
typedef struct {
    int32_t score;
    uint64_t ystart, xstart, yend, xend, ylen, xlen;
    uint64_t n_ops; /* number of operations written to the ops buffer */
    int32_t mode;
} orc_alignment_t;

/* ops: each entry = kind | (len << 8) (len only for Xclip/Yclip).
 * Returns 0, or -1 if ops_cap is too small (n_ops still reports the needed size). */
int orc_align(const orc_scoring_t* sc, int mode, const uint8_t* x, uint64_t m, const uint8_t* y,
              uint64_t n, orc_alignment_t* out, uint64_t* ops, uint64_t ops_cap);

/* Batch of independent pairs over `threads` host threads, one Aligner per thread (the
 * reference's &mut self API forces exactly that). ops for pair p are written at
 * ops + p*ops_stride. Used for differential tests and as the timed CPU baseline. */
/* test hook (pairwise.cpp): custom() without the x-suffix-clip fold of the columns before n — what the engine's LF kernel
 * leaves out; a traceback that asks for such an Lx[j] fails (-2) and is counted */
void orc_test_lf_hook(int on);
uint64_t orc_test_lf_lx_reads(void);
int orc_align_batch(const orc_scoring_t* sc, int mode, uint64_t n_pairs, const uint8_t* x,
                    const uint64_t* x_off, const uint8_t* y, const uint64_t* y_off,
                    orc_alignment_t* out, uint64_t* ops, uint64_t ops_stride, int threads);

/* ---- banded aligner (pairwise/banded.rs) ---- */
int orc_banded_align(const orc_scoring_t* sc, int mode, uint32_t k, uint32_t w, const uint8_t* x,
                     uint64_t m, const uint8_t* y, uint64_t n, orc_alignment_t* out, uint64_t* ops,
                     uint64_t ops_cap, uint64_t* band_cells);
int orc_banded_align_batch(const orc_scoring_t* sc, int mode, uint32_t k, uint32_t w,
                           uint64_t n_pairs, const uint8_t* x, const uint64_t* x_off,
                           const uint8_t* y, const uint64_t* y_off, orc_alignment_t* out,
                           uint64_t* ops, uint64_t ops_stride, uint64_t* band_cells, int threads);
/* Band::create (banded.rs:1278): writes n+1 half-open row ranges; returns num_cells */
uint64_t orc_band_create(const orc_scoring_t* sc, uint32_t k, uint32_t w, const uint8_t* x,
                         uint64_t m, const uint8_t* y, uint64_t n, uint32_t* start, uint32_t* end);
/* sparse::find_kmer_matches (sparse.rs:337) → pairs (x_pos,y_pos); returns count (≤cap written) */
uint64_t orc_find_kmer_matches(const uint8_t* x, uint64_t m, const uint8_t* y, uint64_t n,
                               uint32_t k, uint32_t* out_xy, uint64_t cap);
/* sparse::sdpkpp (sparse.rs:188) / lcskpp (sparse.rs:67); path = indices into matches */
uint64_t orc_sdpkpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k,
                    uint32_t match_score, int32_t gap_open, int32_t gap_extend, uint32_t* path,
                    uint64_t cap, uint32_t* score);
uint64_t orc_lcskpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k, uint32_t* path,
                    uint64_t cap, uint32_t* score);

/* ---- FM index (data_structures/{suffix_array,bwt,fmindex}.rs) ---- */
/* suffix_array(text) (suffix_array.rs:264-284): text must end with a sentinel that is <=
 * every byte; multiple sentinels are ordered by position (transform_text 444-466).
 * Returns 0 or -1 when the sentinel precondition (the reference's assert) fails. */
int orc_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa);
/* bwt(text, sa) (bwt.rs:39-49) */
void orc_bwt(const uint8_t* text, const uint64_t* sa, uint64_t n, uint8_t* bwt);
/* less(bwt, alphabet) (bwt.rs:186-199): out has max_symbol+2 entries; returns that length */
uint64_t orc_less(const uint8_t* bwt, uint64_t n, const uint8_t* alphabet, uint64_t n_sym,
                  uint64_t* less_out);

typedef struct orc_occ orc_occ; /* Occ { occ: Vec<Vec<usize>>, k } (bwt.rs:77-80) */
orc_occ* orc_occ_new(const uint8_t* bwt, uint64_t n, uint32_t k, const uint8_t* alphabet,
                     uint64_t n_sym);
void orc_occ_free(orc_occ*);
/* row a of the sampled table (len written to *len); NULL/0 for non-alphabet symbols */
const uint64_t* orc_occ_row(const orc_occ*, uint32_t a, uint64_t* len);
/* Occ::get (bwt.rs:129-182). Returns -1 where the reference would panic (index out of
 * bounds on a non-alphabet symbol), else 0 and *out = count. */
int orc_occ_get(const orc_occ*, const uint8_t* bwt, uint64_t n, uint64_t r, uint8_t a,
                uint64_t* out);

enum { ORC_BS_COMPLETE = 0, ORC_BS_PARTIAL = 1, ORC_BS_ABSENT = 2, ORC_BS_PANIC = 3 };
/* FMIndexable::backward_search (fmindex.rs:144-208). tag ORC_BS_PANIC marks where the
 * reference would panic on an out-of-alphabet byte (fmindex.rs:229 / bwt.rs:158). */
int orc_backward_search(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                        const orc_occ* occ, const uint8_t* pattern, uint64_t plen,
                        uint64_t* lower, uint64_t* upper, uint64_t* matched_len);
void orc_backward_search_batch(const uint8_t* bwt, uint64_t n, const uint64_t* less,
                               uint64_t less_len, const orc_occ* occ, uint64_t n_q,
                               const uint8_t* pat, const uint64_t* pat_off, uint8_t* tag,
                               uint64_t* lower, uint64_t* upper, uint64_t* matched_len,
                               int threads);
/* The interval of a pattern by definition, without a suffix array (fmindex.rs:63-79,100-102: the suffixes that start with
 * it; suffix_array.rs:264-284: in plain byte order) — lower = #suffixes < P, upper = lower + #suffixes with prefix P — and
 * up to pos_cap occurrence positions per pattern.  For texts too large to sort in test time (tools/exp/fm_wide_big.py). */
void orc_intervals_by_scan(const uint8_t* text, uint64_t n, uint64_t n_pat, const uint8_t* pat, const uint64_t* pat_off,
                           uint64_t* lower, uint64_t* upper, uint64_t* pos_out, uint64_t pos_cap, uint64_t* n_pos, int threads);


/* SampledSuffixArray (suffix_array.rs:86-184): sample() and get().  orc_sampled_sa_get returns 0,
 * -1 for None (index out of range) or -2 where the reference would panic. */
typedef struct orc_sampled_sa orc_sampled_sa;
orc_sampled_sa* orc_sa_sample(const uint64_t* sa, uint64_t n, const uint8_t* text, const uint8_t* bwt,
                              uint64_t sampling_rate);
void orc_sa_sample_free(orc_sampled_sa*);
uint64_t orc_sa_sample_counts(const orc_sampled_sa*, uint64_t* n_extra);
void orc_sa_sample_export(const orc_sampled_sa*, uint64_t* sample, uint64_t* extra_row, uint64_t* extra_pos);
int orc_sampled_sa_get(const orc_sampled_sa*, const uint8_t* bwt, uint64_t n, const uint64_t* less,
                       uint64_t less_len, const orc_occ* occ, uint64_t index, uint64_t* out);


/* FMDIndex (fmindex.rs:250-576).  orc_fmd_check: FMDIndex::from's assert (BWT over n_alphabet + '$').
 * orc_fmd_smems: smems(pattern, i, l) (all == 0) or all_smems(pattern, l); records of 6 uint64
 * {lower, lower_rev, size, match_size, pattern position, length}; returns their number or -1 (panic).
 * orc_fmd_interval: op 0 init_interval, 1 init_interval_with(a), 2 backward_ext(iv, a), 3 forward_ext. */
int orc_fmd_check(const uint8_t* bwt, uint64_t n);
int64_t orc_fmd_smems(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                      const orc_occ* occ, const uint8_t* pattern, uint64_t plen, uint64_t i, uint64_t l,
                      int all, uint64_t* out, uint64_t cap);
int orc_fmd_interval(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                     const orc_occ* occ, int op, const uint64_t* iv, uint8_t a, uint64_t* out);

/* ---- bio::io::fastq (io/fastq.rs) on a byte buffer, and bio-types Alignment::cigar ---- */
enum { ORC_FASTQ_OK = 0, ORC_FASTQ_MISSING_AT = 1, ORC_FASTQ_INCOMPLETE = 2, ORC_FASTQ_IO = 3 }; /* ReadError, fastq.rs:113-126 */
enum { ORC_FQCHECK_OK = 0, ORC_FQCHECK_EMPTY_ID = 1, ORC_FQCHECK_NONASCII_SEQ = 2, ORC_FQCHECK_INVALID_SEQ = 3,
       ORC_FQCHECK_NONASCII_QUAL = 4, ORC_FQCHECK_UNEQUAL = 5 }; /* CheckError, fastq.rs:129-150 */
typedef struct {
    uint64_t id_off, id_len;     /* into the text */
    uint64_t desc_off, desc_len; /* into the text; has_desc == 0: None */
    uint64_t seq_off, seq_len;   /* into the concatenated sequence output */
    uint64_t qual_off, qual_len; /* into the concatenated quality output */
    int32_t has_desc;
    int32_t check;               /* Record::check() of this record */
} orc_fastq_rec_t;
/* Parses records until the end of the text or the first ReadError (status / err_pos = byte offset of the line
 * that raised it); n_records = records read before it.  seq / qual must hold `len` bytes (or be NULL). */
int orc_fastq_parse(const uint8_t* text, uint64_t len, orc_fastq_rec_t* recs, uint64_t rec_cap, uint8_t* seq, uint8_t* qual,
                    uint64_t* n_records, int32_t* status, uint64_t* err_pos);
/* Alignment::cigar(hard_clip): returns the length written, -1 if cap is too small, -2 for AlignmentMode::Custom */
int64_t orc_cigar(const orc_alignment_t* a, const uint64_t* ops, int hard_clip, char* out, uint64_t cap);
/* Alignment::pretty(x, y, ncol): returns the length written, -1 if cap is too small, -2 where the crate panics */
int64_t orc_pretty(const orc_alignment_t* a, const uint64_t* ops, const uint8_t* x, uint64_t xl, const uint8_t* y, uint64_t yl,
                   uint64_t ncol, char* out, uint64_t cap);

/* ---- seed-and-extend composition (oracle/pipeline.cpp): backward_search -> Interval::occ -> Aligner::semiglobal,
 * the caller pattern of src/lib.rs:129-165 / benches/fmindex.rs:20-38; the definition is stated in pipeline.cpp ---- */
typedef struct {
    orc_alignment_t aln;    /* the winning semiglobal alignment (y = the candidate window); score MIN_SCORE: unmapped */
    uint64_t window_start;  /* text offset of the winning window */
    uint64_t ref_start, ref_end; /* window_start + ystart / yend; UINT64_MAX: unmapped */
    uint32_t n_candidates, n_seed_hits;
} orc_seed_hit_t;
int orc_seed_extend_batch(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                          const orc_occ* occ, const uint64_t* sa, const uint8_t* text, uint64_t n_text,
                          const orc_scoring_t* sc, uint64_t n_reads, const uint8_t* reads,
                          const uint64_t* read_off, uint32_t seed_len, uint32_t stride, uint32_t max_occ,
                          uint32_t pad, orc_seed_hit_t* out, uint64_t* ops, uint64_t ops_stride, int threads);
/* the same with a 32-bit suffix array (texts below 2^32 symbols: half the host memory) */
int orc_seed_extend_batch_sa32(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                               const orc_occ* occ, const uint32_t* sa, const uint8_t* text, uint64_t n_text,
                               const orc_scoring_t* sc, uint64_t n_reads, const uint8_t* reads,
                               const uint64_t* read_off, uint32_t seed_len, uint32_t stride, uint32_t max_occ,
                               uint32_t pad, orc_seed_hit_t* out, uint64_t* ops, uint64_t ops_stride, int threads);

#ifdef __cplusplus
}
#endif
#endif
