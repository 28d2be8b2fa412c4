/*
 * biogpu.h — C ABI of the MI355X-native engine that sits behind rust-bio's
 *   bio::alignment::pairwise::Aligner            (src/alignment/pairwise/mod.rs:472-1016)
 *   bio::alignment::pairwise::banded::Aligner    (src/alignment/pairwise/banded.rs:122-1004)
 *   bio::data_structures::fmindex::FMIndex       (src/data_structures/fmindex.rs:98-248)
 * and the host-side table builders they are fed from
 *   suffix_array / bwt / less / Occ::new         (suffix_array.rs:264, bwt.rs:39,186,94).
 *
 * rust-bio has no FFI of its own; these are the entry points a Rust shim
 * inside `bio` binds with `extern "C"` (INTEGRATION.md shows the binding).  Plain pointers
 * and sizes only.  Every function returns BG_OK (0) or a negative bg_status; nothing panics
 * or throws across the boundary.  Where the reference would `panic!` (assert on positive
 * penalties, index-out-of-bounds on a byte outside the alphabet, missing sentinel) the
 * matching error code is returned and documented at the function.
 *
 * Two flavours per batched op:
 *   bg_*_batch      caller passes HOST buffers (the drop-in boundary; includes PCIe copies)
 *   bg_*_batch_dev  caller passes DEVICE pointers + a hipStream_t (as void*): inputs already
 *                   resident in HBM, results left in HBM, asynchronous on that stream.
 *
 * Streams and threads.  A bg_ctx owns ONE set of device scratch (traceback words, aux records, scoring table).
 * *_dev calls that use it may be issued on different streams: a call arriving on another stream than the
 * previous one first waits, on the device, for that call's last kernel (an event per ctx), so two calls in
 * flight never share scratch — they serialise.  For real overlap use one bg_ctx per stream.  A bg_ctx must
 * not be used from two host threads at once (like `&mut Aligner`).  A bg_fm is immutable after bg_fm_build /
 * bg_fm_set_*; bg_fm_backward_search_batch_dev, bg_sa_get_batch_dev and bg_interval_occ_batch_dev use no ctx
 * scratch and may be called from several threads on their own streams (with bg_enable_timing off — the timing
 * events belong to the ctx); the host-buffer flavours and bg_fmd_smems_* go through the handle's ctx and share
 * its single-thread rule.
 *
 * Several GPUs.  One process (and one bg_ctx, one replica of a bg_fm) per device; pairs / queries / reads are split in
 * contiguous ranges and nothing is exchanged during a call.  What a caller gathers afterwards (one all-gather) are the
 * FIXED-SIZE results: bg_alignment_t headers (score + coordinates), search intervals + tags, bg_seed_hit_t headers.
 * Operation lists are variable-length and ops_off indexes the buffer of the process that made them: they stay where
 * they are unless the caller gathers byte counts (ops_used) and bytes itself and re-bases ops_off.
 *
 * Text positions are 64-bit at the boundary, like the reference's usize (fmindex.rs:70-71, bwt.rs:94, suffix_array.rs:
 * 264).  Inside, an index below 2^32 - 1 symbols keeps the uint32 layout of rounds 1-4 (and its speed); from 2^32 - 1
 * symbols on — T$R$ of a human genome for an FMD index is 6.2 G — bg_fm_build / bg_fm_build_dev lay the SAME rank blocks
 * out with superblock-relative counters and 64-bit bases (csrc/fm_wide.hip) and every entry point that takes a bg_fm works
 * on them: backward search with byte or 2-bit packed patterns (2-step blocks included), bg_fm_set_[sampled_]suffix_array,
 * bg_sa_get_batch[_dev], bg_interval_occ_batch[_dev], bg_fm_set_text + bg_seed_extend_batch[_dev], bg_fm_save / bg_fm_load,
 * and the FMD kernels through their *64 entry points (bg_fmd_smems_batch64[_dev], bg_fmd_interval_batch64: uint64 records;
 * the uint32-record flavours answer BG_ERR_UNSUPPORTED on such an index, the *64 flavours serve both layouts).
 * bg_suffix_array_dev64 / bg_bwt_dev64 / bg_sa_sample_dev64 build the 64-bit suffix array in HBM.  BG_ERR_TOO_LARGE only
 * beyond 2^40 symbols.  The one thing a 64-bit index does NOT offer (BG_ERR_UNSUPPORTED at build time): BWTs with more
 * than 1024 positions outside their four most frequent bytes (protein texts, genomes with long N runs: they would need
 * rank bit vectors with 64-bit bases); bg_fm_backward_search_count_lines_dev is a measurement aid of the 32-bit kernels.
 * A sequence of an aligner call may have up to 2^24 symbols.
 */
#ifndef BIOGPU_H
#define BIOGPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    BG_OK = 0,
    BG_ERR_INVALID_ARG = -1,
    BG_ERR_NO_DEVICE = -2,       /* no usable HIP device / extension cannot run */
    BG_ERR_HIP = -3,             /* a HIP runtime call failed (see bg_last_error) */
    BG_ERR_OOM = -4,
    BG_ERR_SENTINEL = -5,        /* suffix_array.rs:431-437 assert: last byte must be the smallest */
    BG_ERR_POSITIVE_PENALTY = -6,/* pairwise/mod.rs:265-266,292-293,554-571 asserts */
    BG_ERR_OUT_OF_ALPHABET = -7, /* fmindex.rs:229 / bwt.rs:114,158 index-out-of-bounds panics */
    BG_ERR_TOO_LARGE = -8,       /* text > 2^40 symbols (>= 2^32 - 1 for the uint32 suffix-array flavours), or sequence too long for the engine */
    BG_ERR_OPS_CAP = -9,         /* caller's ops buffer too small (ops_used reports the need) */
    BG_ERR_TRACEBACK = -10,      /* traceback did not terminate (reference would loop forever) */
    BG_ERR_UNSUPPORTED = -11,    /* legal for rust-bio, not yet covered by the device layout */
    BG_ERR_IO = -12              /* bg_fm_save / bg_fm_load: the file cannot be opened, is truncated, or fails its checksum */
} bg_status;

#define BG_MIN_SCORE (-858993459) /* pairwise::MIN_SCORE, mod.rs:174 */

typedef struct bg_ctx bg_ctx; /* one per device; not thread-safe (like `&mut Aligner`) */
typedef struct bg_fm bg_fm;   /* device-resident FM index; immutable after construction (like Arc<FMIndex>): see "Streams and threads" */

int bg_device_count(void);
int bg_init(int device, bg_ctx** out);
int bg_free(bg_ctx* ctx);
const char* bg_strerror(int status);
const char* bg_last_error(void); /* text of the last HIP failure on this thread */
/* Tunables (0 keeps the default) and the switches the tests use to reach every kernel variant:
 *   chunk_pairs        pairs per sub-batch of bg_align_batch_dev (default 2^20) and of the banded pipeline (16384)
 *   host_chunk_pairs   pairs per stage of bg_align_batch's pipelined host path (122880)
 *   seed_chunk_reads   reads per pass of bg_seed_extend_batch[_dev] (0: equal passes of at most 2^21 reads)
 *   force_wide = 1     scores kept as plain int32 even where they fit the 24-bit keys of the fast kernels
 *   no_pk16 = 1        no packed-int16 fill (K1p): the int32 kernel K1 runs for every batch
 *   no_couples = 1     K1p without the (m, n) slot order on ragged batches
 *   no_local_fast = 1  Aligner::local batches on the general K1p instead of its LF flavour (tests, A/B)
 *   fm_host_bytes = 1  bg_fm_backward_search_batch stages the pattern bytes as they are (default: packed to 2-bit codes by
 *                      the host threads where the index takes packed patterns; tests, A/B)
 *   band_on_host = 1   bands built by the host threads instead of the device builder
 *   band_fill_v1       1: banded fill with one pair per wavefront (K3) always; -1: eight pairs per wavefront (K3v2)
 *                      always; 0 (default): K3 for sub-batches of at most 2048 pairs (latency), K3v2 above (throughput)
 *   band_interior_off = 1  banded fill (K3v2) with its general step in every strip (tests, A/B: no reduced interior step)
 *   band_packed_off = 1    interior runs on the int32 kernel (K3i) only: no packed-int16 kernel (K3p) in front of it
 *   band_packed_thresh     K3p's detect-and-recompute threshold in key units (score * 16); 0 = derived from the scoring,
 *                          65535 = every pair is flagged and recomputed by the int32 kernels (tests)
 *   band_tail_last / band_window / band_raster_late = 1  A/B switches of the banded pipeline's order (round-3 behaviour)
 *   band_join_serial = 1   the k-mer join of a sub-batch on the builder's stream instead of its own (A/B)
 *   band_pre_serial = 1    a banded fill's preparation (pair table, waits, first strips) on the fill stream instead of its own (A/B)
 *   band_chain_global  chaining tree placement: 0 LDS, 1 global scratch, -1 by batch size (default)
 *   band_join_global = 1  k-mer join with its table in global memory even where the LDS flavour applies
 *   band_join_late = 1    the k-mer join of a sub-batch waits for the chaining of the one before it (A/B: measured slower)
 *   band_p_block512 = 1   K3p in blocks of eight wavefronts compiled for 168 VGPRs instead of four at 187 (A/B: measured slower)
 *   band_chain_rows    global-tree chaining with four pairs per wavefront (chain_rows_kernel; default 1) or one (0: A/B, tests)
 *   band_host_sync = 1 the chaining of a sub-batch is launched after a host wait for K4 of two sub-batches ago (rounds 2-4)
 *                      instead of a stream wait (A/B)
 *   fm_wide_from       texts of this many symbols or more get the 64-bit index layout (default 2^32 - 1; tests lower it so
 *                      that small texts exercise csrc/fm_wide.hip); 0 restores the default
 *   fm_wide_sb_shift   log2 of the rank blocks per superblock of the 64-bit layout (default 17; 0 .. 24; tests use small
 *                      values so that short texts span many superblocks)
 *   fq_no_fused = 1    bg_fastq_parse[_dev] through its general multi-pass kernels only, without the one-pass kernel that serves
 *                      four-line ASCII records in front (tests, A/B)
 *   sa_chunk_symbols   suffixes sorted per pass of round 0 of bg_suffix_array_dev[64] (0 = derived from free device memory;
 *                      tests use small values so that short texts take several passes)
 *   band_budget_gb     traceback + aux bytes per scratch set of the banded pipeline, in GB (0 = default: 40, and never more
 *                      than a third of the device's free memory); a sub-batch that does not fit is cut
 * Unknown keys return BG_ERR_INVALID_ARG. */
int bg_set_option(bg_ctx* ctx, const char* key, int64_t value);

/* ------------------------------------------------------------------ host table builders
 * Same contracts as the reference functions; pure host code (usable without a GPU). */

/* suffix_array(text) — suffix_array.rs:264-284.  text must end in a sentinel <= every other
 * byte, else BG_ERR_SENTINEL (the reference asserts).  Several sentinels are ordered by
 * position: the first occurrence sorts last among them (transform_text, 444-466). */
int bg_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa_out);
/* bwt(text, pos) — bwt.rs:39-49 */
int bg_bwt(const uint8_t* text, const uint64_t* sa, uint64_t n, uint8_t* bwt_out);
/* less(bwt, alphabet) — bwt.rs:186-199.  less_out must hold max_symbol+2 entries;
 * *less_len receives that length (call with less_out == NULL to query it). */
int bg_less(const uint8_t* bwt, uint64_t n, const uint8_t* alphabet, uint32_t n_sym,
            uint64_t* less_out, uint32_t* less_len);

/* Device flavours for a text that already lives in HBM (sa_build.hip): suffix array by prefix doubling (radix sorts
 * of (rank, rank) keys), n uint32 entries; BWT gather; RawSuffixArray::sample (suffix_array.rs:86-120) with the
 * samples and the sentinel rows returned to host arrays (what bg_fm_set_sampled_suffix_array takes: sample holds
 * ceil(n / rate) entries, extra rows come back sorted, *n_extra says how many; BG_ERR_OPS_CAP beyond extra_cap).
 * The results equal the host functions' (the suffix array of the transformed text is unique), also for texts whose
 * sentinel byte occurs several times — several sequences, or T$R$ for an FMD index (fmindex.rs:312-340): the sentinels
 * rank by position, the last occurrence smallest (transform_text, suffix_array.rs:444-466).  The uint32 flavours take
 * texts up to 2^32 - 2 symbols (BG_ERR_TOO_LARGE beyond) with about 29 bytes of device scratch per symbol; the *64
 * flavours are the same algorithm on uint64 positions (suffix_array.rs:264: usize) for texts up to 2^40 symbols, 49 bytes
 * of scratch per symbol (a 4.4 G-symbol text: 216 GB next to its 35 GB suffix array on the 288 GB part).  Synchronous. */
int bg_suffix_array_dev(bg_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* stream);
int bg_bwt_dev(bg_ctx* ctx, const uint8_t* d_text, const uint32_t* d_sa, uint64_t n, uint8_t* d_bwt, void* stream);
int bg_sa_sample_dev(bg_ctx* ctx, const uint32_t* d_sa, const uint8_t* d_bwt, uint64_t n, uint32_t sampling_rate,
                     uint8_t sentinel, uint64_t* sample, uint64_t* extra_rows, uint64_t* extra_pos, uint64_t extra_cap,
                     uint64_t* n_extra, void* stream);
int bg_suffix_array_dev64(bg_ctx* ctx, const uint8_t* d_text, uint64_t n, uint64_t* d_sa, void* stream);
int bg_bwt_dev64(bg_ctx* ctx, const uint8_t* d_text, const uint64_t* d_sa, uint64_t n, uint8_t* d_bwt, void* stream);
int bg_sa_sample_dev64(bg_ctx* ctx, const uint64_t* d_sa, const uint8_t* d_bwt, uint64_t n, uint32_t sampling_rate,
                       uint8_t sentinel, uint64_t* sample, uint64_t* extra_rows, uint64_t* extra_pos, uint64_t extra_cap,
                       uint64_t* n_extra, void* stream);

/* ------------------------------------------------------------------ FM index
 * bg_fm_build replaces `Occ::new(&bwt, k, &alphabet)` + `FMIndex::new(bwt, less, occ)`
 * (bwt.rs:94-125, fmindex.rs:245-247): the sampled-Occ table layout on the device is the
 * engine's own, `occ_k` is accepted for API fidelity and only validated (>= 1):
 * Occ::get's result does not depend on k.  The reference's third argument, the host-built `Occ` table itself,
 * is deliberately NOT part of this signature: the engine ranks on its own packed blocks built from `bwt`, so a
 * Rust shim drops its `Occ` (or never builds it) and passes (bwt, less, k, alphabet) — INTEGRATION.md.  BG_ERR_OUT_OF_ALPHABET when a BWT byte exceeds
 * the alphabet's max symbol (Occ::new would panic, bwt.rs:114). */
int bg_fm_build(bg_ctx* ctx, const uint8_t* bwt, uint64_t n, const uint64_t* less,
                uint32_t less_len, uint32_t occ_k, const uint8_t* alphabet, uint32_t n_sym,
                bg_fm** out);
/* The same index from a BWT that lives in HBM (with bg_suffix_array_dev / bg_bwt_dev: text to searchable index
 * without a host copy of anything text-sized).  `less(bwt, alphabet)` (bwt.rs:186-199) falls out of the byte
 * histogram the layout needs anyway: it is computed here and, if less_out is given (max_symbol + 2 entries),
 * returned.  Identical handle to bg_fm_build's on the same BWT.  Synchronous. */
int bg_fm_build_dev(bg_ctx* ctx, const uint8_t* d_bwt, uint64_t n, uint32_t occ_k, const uint8_t* alphabet,
                    uint32_t n_sym, uint64_t* less_out, bg_fm** out, void* stream);
int bg_fm_free(bg_fm* fm);
uint64_t bg_fm_device_bytes(const bg_fm* fm);
/* Index persistence.  The reference derives Serialize / Deserialize for Occ (bwt.rs:76), FMIndex (fmindex.rs:214) and
 * SampledSuffixArray (suffix_array.rs:124): an index is built once per genome and loaded afterwards.  bg_fm_save writes what
 * the reference's FMIndex holds — the BWT (read back out of the rank blocks), less, the alphabet and k — plus the suffix
 * array attached to the handle (raw, or sampled with its extra rows) and the text if the handle owns a copy
 * (bg_fm_set_text); bg_fm_load lays the index out again (the rank blocks are cheap: csrc/fm_persist.hip) and attaches
 * them: the loaded handle answers every call like the saved one.  The file is the engine's own format (little-endian,
 * checksummed), not serde's.  BG_ERR_IO: cannot open / truncated / altered. */
int bg_fm_save(const bg_fm* fm, const char* path);
int bg_fm_load(bg_ctx* ctx, const char* path, bg_fm** out);
/* What a handle says about itself — a loaded handle comes without the caller-side arrays it was built from, and the
 * reference's FMDIndex::from(fmindex) (fmindex.rs:311-329) reads the BWT of whatever FMIndex it is given, deserialized
 * or not.  bg_fm_len: the text length n (FMIndex's bwt.len()).  bg_fm_less: the `less` array the index answers with
 * (less_out may be NULL to query *less_len = max_symbol + 2).  bg_fm_bwt / bg_fm_bwt_dev: the n BWT bytes, read back
 * out of the rank blocks (2-bit codes -> bytes, listed exceptions put back) into host / device memory; the device
 * flavour is asynchronous on `stream`. */
int bg_fm_len(const bg_fm* fm, uint64_t* n);
int bg_fm_less(const bg_fm* fm, uint64_t* less_out, uint32_t* less_len);
int bg_fm_bwt(const bg_fm* fm, uint8_t* bwt);
int bg_fm_bwt_dev(const bg_fm* fm, uint8_t* d_bwt, void* stream);
/* Bytes of the index's 2-step rank blocks (128-byte lines: 16 pair counters + 128 four-bit pair codes per 128 BWT
 * positions; built behind DNA-like indexes whose `less` is the BWT's own, fm_step2.hip) that the searches take two
 * pattern symbols per block access from; 0: single steps (no such blocks, or bg_fm_set_option "no_step2" = 1). */
uint64_t bg_fm_step2_bytes(const bg_fm* fm);

/* Result tags of FMIndexable::backward_search (fmindex.rs:92-96). */
enum { BG_FM_COMPLETE = 0, BG_FM_PARTIAL = 1, BG_FM_ABSENT = 2,
       BG_FM_PANIC = 3 /* this query reached a byte outside the alphabet */ };

/* Options of an index handle: "jump_min_queries" — batch size from which backward search builds (once,
 * 256 MB, synchronously on the first such call's stream, under a lock) and uses a table of the search state
 * after a pattern's last 12 symbols; < 0 disables it.  OFF by default: it buys 2.5 % on an index that sits in
 * the Infinity Cache and nothing on one that does not.  Results do not depend on it.
 * "ilp" — queries a quad of lanes walks at once in backward search: 2 (default: fm_search_fast2x_kernel on the 2-step blocks
 * of a DNA-like index — on either position width — fmw_search2x_kernel on a 64-bit index without such blocks) or 1 (the
 * round-4 kernels; A/B, tests).  "no_step2" = 1 —
 * single LF steps even where the index has 2-step blocks; "no_fast" = 1 — every search through the generic kernel (tests).
 * None of them changes a result. */
int bg_fm_set_option(bg_fm* fm, const char* key, int64_t value);

/* backward_search for n_q patterns (fmindex.rs:144-208).  Pattern q is
 * pat[pat_off[q] .. pat_off[q+1]).  Outputs per query: tag, Interval{lower,upper} (for
 * Partial: the interval of the maximal matching suffix), matched_len (Complete: |P|).
 * Returns BG_ERR_OUT_OF_ALPHABET if any query has tag BG_FM_PANIC (all other queries are
 * still valid). */
int bg_fm_backward_search_batch(bg_fm* fm, uint64_t n_q, const uint8_t* pat,
                                const uint64_t* pat_off, uint8_t* tag, uint64_t* lower,
                                uint64_t* upper, uint32_t* matched_len);
/* Same with device pointers; asynchronous on `stream`; no panic scan (tags tell). */
int bg_fm_backward_search_batch_dev(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat,
                                    const uint64_t* d_pat_off, uint8_t* d_tag, uint64_t* d_lower,
                                    uint64_t* d_upper, uint32_t* d_matched_len, void* stream);

/* ---- 2-bit packed sequences (pack2.hip) ---------------------------------------------------------
 * rust-bio's Aligner and FMIndex take `&[u8]` (pairwise/mod.rs:591, fmindex.rs:144); the engine also accepts its own
 * 2-bit wire format, what BASELINE's north_star calls "packed 2-bit reads".  A byte buffer — any concatenation of
 * sequences, e.g. the `seq` buffer bg_fastq_parse_dev fills — is packed as ONE stream: symbol s sits in bits
 * 2 (s % 16) .. + 1 of little-endian dword s / 16, `codes[c]` is the byte value of code c (four distinct bytes).
 * Sequence boundaries do not matter to the packing: the offset arrays of the byte flavours (x_off / pat_off /
 * seq_off, in symbols) address the packed stream unchanged.  d_packed must hold (n + 15) / 16 + 1 dwords (consumers
 * may read one dword past the last symbol; its content is never interpreted).  A byte that is none of the four
 * codes is stored as code 0 and counted in *d_n_invalid (a device counter the caller zeroes; may be NULL): a buffer
 * with a non-zero count must take the byte entry points.  Asynchronous on `stream`. */
int bg_pack2_dev(bg_ctx* ctx, const uint8_t* d_bytes, uint64_t n, const uint8_t* codes, uint32_t* d_packed,
                 uint64_t* d_n_invalid, void* stream);
int bg_unpack2_dev(bg_ctx* ctx, const uint32_t* d_packed, uint64_t n, const uint8_t* codes, uint8_t* d_bytes,
                   void* stream);
/* The same packing on the host (no GPU involved; AVX2 where the CPU has it): `packed` must hold (n + 15) / 16 dwords.
 * Returns 1 if every byte was one of the four codes, 0 if not (such bytes are stored as whatever their low bits say:
 * take the byte entry points), a negative BG_ERR_* for bad arguments.  bg_fm_backward_search_batch packs its stages
 * with it on the worker threads. */
int bg_pack2_host(const uint8_t* bytes, uint64_t n, const uint8_t* codes, uint32_t* packed);
/* The byte values of the index's four 2-bit codes — the `codes` to pack its patterns with.  BG_ERR_UNSUPPORTED when
 * the text has fewer than four frequent letters or symbols ranked by bit vectors: such an
 * index takes byte patterns only.  (Both position widths.) */
int bg_fm_pattern_codes(const bg_fm* fm, uint8_t* codes);
/* bg_fm_backward_search_batch_dev on patterns packed with bg_fm_pattern_codes' codes; d_sym_off[q] is the SYMBOL
 * offset of pattern q in the stream (n_q + 1 entries).  Same outputs.  (A dword load per 16 steps instead of a byte
 * load per step, no symbol-class lookups: measured against the byte flavour in bench.py's `fm.packed2`.) */
int bg_fm_backward_search_packed_dev(bg_fm* fm, uint64_t n_q, const uint32_t* d_packed, const uint64_t* d_sym_off,
                                     uint8_t* d_tag, uint64_t* d_lower, uint64_t* d_upper, uint32_t* d_matched_len,
                                     void* stream);
/* Measurement aid: bg_fm_backward_search_batch_dev with the 64-byte block loads it issues counted (*lines_out, host
 * memory; synchronous).  bench.py sets the count against the chip's measured ceiling for such gathers
 * (tools/microbench/ub_gather64.hip). */
int bg_fm_backward_search_count_lines_dev(bg_fm* fm, uint64_t n_q, const uint8_t* d_pat, const uint64_t* d_pat_off,
                                          uint8_t* d_tag, uint64_t* d_lower, uint64_t* d_upper, uint32_t* d_matched_len,
                                          uint64_t* lines_out, void* stream);

/* ---- suffix-array lookups for FM-index hits (kernel K6) ------------------------------------
 * Attach the suffix array the FM index was built from, then resolve rows to text positions.
 * The index handle must come from bg_fm_build over the same text (n rows). */
#define BG_SA_NONE 0xFFFFFFFFFFFFFFFFull  /* SuffixArray::get -> None (row >= len) */
#define BG_SA_PANIC 0xFFFFFFFFFFFFFFFEull /* the reference would panic (missing extra row / non-alphabet byte) */

/* RawSuffixArray (suffix_array.rs:25, get 134-141): the full SA, n entries */
int bg_fm_set_suffix_array(bg_fm* fm, const uint64_t* sa, uint64_t n);
/* SampledSuffixArray as RawSuffixArray::sample builds it (suffix_array.rs:86-120): sample[i] =
 * SA[i * sampling_rate], n_sample = ceil(n / rate); sentinel = last byte of the text; the rows
 * whose BWT byte is the sentinel and that are not sampled, sorted by row, with their positions
 * (`extra_rows`, a hash map in the reference). */
int bg_fm_set_sampled_suffix_array(bg_fm* fm, const uint64_t* sample, uint64_t n_sample,
                                   uint32_t sampling_rate, uint8_t sentinel,
                                   const uint64_t* extra_rows, const uint64_t* extra_pos,
                                   uint64_t n_extra);
/* SuffixArray::get for a batch of rows (suffix_array.rs:134-141 raw, 157-184 sampled):
 * pos[i] = SA[index[i]], BG_SA_NONE if index[i] >= n.  Returns BG_ERR_OUT_OF_ALPHABET if any
 * row hit BG_SA_PANIC. */
int bg_sa_get_batch(bg_fm* fm, uint64_t n_idx, const uint64_t* index, uint64_t* pos);
int bg_sa_get_batch_dev(bg_fm* fm, uint64_t n_idx, const uint64_t* d_index, uint64_t* d_pos,
                        void* stream);
/* Interval::occ for a batch (fmindex.rs:75-79): positions of interval v are written to
 * pos[out_off[v] .. out_off[v+1]) in row order; out_off (n_iv + 1 entries) is filled by the call.
 * BG_ERR_INVALID_ARG if an interval exceeds the suffix array (the reference's expect() panic),
 * BG_ERR_OPS_CAP if pos_cap is too small (out_off is still filled). */
int bg_interval_occ_batch(bg_fm* fm, uint64_t n_iv, const uint64_t* lower, const uint64_t* upper,
                          uint64_t* out_off, uint64_t* pos, uint64_t pos_cap);
/* Device flavour: the caller supplies the prefix offsets (exclusive scan of upper - lower) and
 * their total; asynchronous on `stream`. */
int bg_interval_occ_batch_dev(bg_fm* fm, uint64_t n_iv, const uint64_t* d_lower,
                              const uint64_t* d_out_off, uint64_t total, uint64_t* d_pos,
                              void* stream);

/* ---- FMD index: supermaximal exact matches (kernel K7) ---------------------------------------
 * FMDIndex::smems(pattern, i, l) (fmindex.rs:363-434; all == 0, i_pos[q] = i) or
 * FMDIndex::all_smems(pattern, l) (479-501; all != 0, i_pos may be NULL) for a batch of patterns.
 * The index must have been built (bg_fm_build) over T$R$-style text whose BWT is a word over
 * dna::n_alphabet() + '$' — FMDIndex::from's assert (323-327), else BG_ERR_UNSUPPORTED.
 * Pattern q's matches are the records out[(q*cap + t)*6 ..], t < min(count[q], cap), six uint32:
 * BiInterval {lower, lower_rev, size, match_size}, position on the pattern, SMEM length — in the
 * order the reference pushes them.  count[q] == 0xFFFFFFFF where the reference would panic
 * (BG_ERR_OUT_OF_ALPHABET); BG_ERR_OPS_CAP if some count exceeds cap. */
int bg_fmd_smems_batch(bg_fm* fm, int all, uint64_t n_p, const uint8_t* pat, const uint64_t* pat_off,
                       const uint32_t* i_pos, uint32_t min_len, uint32_t cap, uint32_t* count,
                       uint32_t* out);
/* Single bi-interval steps for a batch of requests: op[q] = 0 init_interval (fmindex.rs:517-524),
 * 1 init_interval_with(sym[q]) (504-514), 2 backward_ext(iv_in[q], sym[q]) (527-558), 3 forward_ext
 * (560-564).  Intervals are four uint32 {lower, lower_rev, size, match_size}. */
int bg_fmd_interval_batch(bg_fm* fm, uint64_t n_req, const uint8_t* op, const uint32_t* iv_in,
                          const uint8_t* sym, uint32_t* iv_out);
int bg_fmd_smems_batch_dev(bg_fm* fm, int all, uint64_t n_p, const uint8_t* d_pat,
                           const uint64_t* d_pat_off, const uint32_t* d_i_pos, uint32_t min_len,
                           uint32_t max_pattern_len, uint32_t cap, uint32_t* d_count, uint32_t* d_out,
                           void* stream);
/* The same three with uint64 records — the reference's BiInterval is usize throughout (fmindex.rs:254-259) — for any
 * index, and the only flavour an index with 64-bit positions answers (T$R$ of a human genome: 6.2 G symbols): six uint64
 * per match {lower, lower_rev, size, match_size, position, length}, four per interval. */
int bg_fmd_smems_batch64(bg_fm* fm, int all, uint64_t n_p, const uint8_t* pat, const uint64_t* pat_off,
                         const uint32_t* i_pos, uint32_t min_len, uint32_t cap, uint32_t* count,
                         uint64_t* out);
int bg_fmd_interval_batch64(bg_fm* fm, uint64_t n_req, const uint8_t* op, const uint64_t* iv_in,
                            const uint8_t* sym, uint64_t* iv_out);
int bg_fmd_smems_batch64_dev(bg_fm* fm, int all, uint64_t n_p, const uint8_t* d_pat,
                             const uint64_t* d_pat_off, const uint32_t* d_i_pos, uint32_t min_len,
                             uint32_t max_pattern_len, uint32_t cap, uint32_t* d_count, uint64_t* d_out,
                             void* stream);

/* ------------------------------------------------------------------ pairwise alignment */

/* Scoring<F> (pairwise/mod.rs:238-247) as the *effective* values `custom` sees.  match_fn is
 * MatchParams{match_score,mismatch_score} when matrix == NULL, else the closure tabulated by
 * the host: matrix[a*256+b] = F(a,b) (int32[65536]).  match_scores_some mirrors
 * `match_scores: Option<(i32,i32)>`, read only by the banded aligner (banded.rs:1315-1318). */
typedef struct {
    int32_t gap_open, gap_extend;
    int32_t xclip_prefix, xclip_suffix, yclip_prefix, yclip_suffix;
    int32_t match_score, mismatch_score;
    int32_t match_scores_some;
    const int32_t* matrix;
} bg_scoring_t;

/* AlignmentMode (bio-types) */
enum { BG_MODE_CUSTOM = 0, BG_MODE_GLOBAL = 1, BG_MODE_SEMIGLOBAL = 2, BG_MODE_LOCAL = 3 };
/* AlignmentOperation, one byte each in the ops buffer.  Xclip/Yclip lengths are the
 * entries of bg_alignment_t.clip_len, in the order the clip ops appear. */
enum { BG_OP_MATCH = 0, BG_OP_SUBST = 1, BG_OP_DEL = 2, BG_OP_INS = 3, BG_OP_XCLIP = 4,
       BG_OP_YCLIP = 5 };

/* bio_types::alignment::Alignment (fields as constructed at pairwise/mod.rs:911-921) */
typedef struct {
    int32_t score;
    uint32_t xstart, xend, ystart, yend, xlen, ylen;
    uint32_t n_ops;
    uint64_t ops_off;     /* offset of this alignment's first op in the ops buffer */
    uint32_t clip_len[4]; /* lengths of the Xclip/Yclip ops, in order of appearance */
    uint8_t n_clips;
    uint8_t mode;         /* BG_MODE_* */
    int8_t status;        /* BG_OK, BG_ERR_TRACEBACK, or BG_ERR_INVALID_ARG (longer than the stated bounds) for this pair */
    uint8_t _pad;
    uint32_t _reserved;   /* 0: the record is 64 bytes, every one of them defined */
} bg_alignment_t;

/* Aligner::{custom,global,semiglobal,local} for n_pairs independent pairs
 * (mod.rs:591,925,954,986).  `mode` selects the wrapper: GLOBAL/SEMIGLOBAL/LOCAL overwrite
 * the four clip penalties exactly as the reference does (and SEMIGLOBAL/LOCAL drop the clip
 * ops, mod.rs:974,1006); CUSTOM uses sc as given.  x/y are concatenated sequences with
 * n_pairs+1 offsets each.  Returns BG_ERR_POSITIVE_PENALTY when a penalty is > 0.
 * ops_buf may be NULL (scores/coordinates only). */
int bg_align_batch(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs,
                   const uint8_t* x, const uint64_t* x_off, const uint8_t* y,
                   const uint64_t* y_off, bg_alignment_t* out, uint8_t* ops_buf,
                   uint64_t ops_cap, uint64_t* ops_used);
/* Device-resident flavour.  d_ops must hold n_pairs * ops_stride bytes where
 * ops_stride >= max(xlen+ylen)+4 over the batch; alignment p's ops end at
 * d_ops + (p+1)*ops_stride and ops_off points at its first op.  sc->matrix (if any) is a
 * HOST pointer (it is compacted and uploaded by the call, which then synchronises `stream` once).
 * max_xlen/max_ylen are upper bounds on the sequence lengths in the batch: a pair that exceeds them is not
 * aligned and gets status BG_ERR_INVALID_ARG in its record (xlen/ylen filled in, no operations). */
int bg_align_batch_dev(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs,
                       const uint8_t* d_x, const uint64_t* d_x_off, const uint8_t* d_y,
                       const uint64_t* d_y_off, uint32_t max_xlen, uint32_t max_ylen,
                       bg_alignment_t* d_out, uint8_t* d_ops, uint64_t ops_stride, void* stream);
/* The same on 2-bit streams (bg_pack2_dev above): d_x / d_y hold 16 symbols per dword, the offsets count symbols,
 * `codes` are the bytes the four codes stand for.  Short reads under MatchParams scoring (the packed-int16 kernel) read
 * the codes as they are and the call stays asynchronous like bg_align_batch_dev.  Every other case (long reads, wide
 * scores, a matrix) is a FALLBACK that is not: it reads the two stream lengths d_x_off[n_pairs] / d_y_off[n_pairs] back
 * (two blocking copies + one synchronisation of `stream`), may grow the ctx's unpack scratch (hipMalloc), unpacks both
 * streams there and runs the byte kernels — same results either way.  That scratch belongs to the ctx: calls on one ctx
 * are serialised by the ctx's scratch guard (an event), so use one ctx per stream for concurrent batches. */
int bg_align_batch_packed_dev(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs, const uint32_t* d_x,
                              const uint64_t* d_x_off, const uint32_t* d_y, const uint64_t* d_y_off,
                              const uint8_t* codes, uint32_t max_xlen, uint32_t max_ylen, bg_alignment_t* d_out,
                              uint8_t* d_ops, uint64_t ops_stride, void* stream);

/* banded::Aligner::{custom,global,semiglobal,local} (banded.rs:282,872,901,972) with k-mer
 * length k and window w.  The band (k-mer matching, sparse DP chaining, Band construction,
 * banded.rs:1278-1367) is built on the device by this call (band_device.hip; the few pairs its fixed-size
 * tables cannot hold — > 4095 matches, > 32 occurrences of one k-mer — are rebuilt by the host builder,
 * band_host.cpp, with identical results); pairs whose band exceeds MAX_CELLS
 * (banded.rs:104) get the reference's sentinel alignment {score: MIN_SCORE, all zero, no ops,
 * mode Custom} (banded.rs:407-420).  band_cells (optional) receives Band::num_cells. */
int bg_align_banded_batch(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w,
                          uint64_t n_pairs, const uint8_t* x, const uint64_t* x_off,
                          const uint8_t* y, const uint64_t* y_off, bg_alignment_t* out,
                          uint8_t* ops_buf, uint64_t ops_cap, uint64_t* ops_used,
                          uint64_t* band_cells);

/* Same with device pointers for the sequences, offsets, records and operation slots (bg_align_batch_dev
 * conventions: record p keeps ops_off = (p + 1) * ops_stride - n_ops, its operations right-aligned in
 * slot p; ops_stride >= longest x + longest y + 4).  The call is synchronous (it drives the engine's own
 * streams); work queued on `stream` before it is waited for.  band_cells is a host array (optional). */
int bg_align_banded_batch_dev(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w,
                              uint64_t n_pairs, const uint8_t* d_x, const uint64_t* d_x_off,
                              const uint8_t* d_y, const uint64_t* d_y_off, bg_alignment_t* d_out,
                              uint8_t* d_ops, uint64_t ops_stride, uint64_t* band_cells, void* stream);

/* Band::create (banded.rs:1278-1367) for a batch, on host threads: k-mer matching, sparse DP
 * chaining (sparse.rs:188-295) and band rasterisation under the clip penalties `mode` implies.
 * Pair p's n_p+1 half-open row ranges [start, end) are written at band_off[p]; band_cells
 * (optional) receives Band::num_cells.  Pure host code. */
int bg_band_create_batch(const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w, uint64_t n_pairs,
                         const uint8_t* x, const uint64_t* x_off, const uint8_t* y, const uint64_t* y_off,
                         const uint64_t* band_off, uint32_t* start, uint32_t* end, uint64_t* band_cells);

/* compute_alignment (banded.rs:406-869) over caller-supplied bands — the common tail of
 * custom_with_prehash / custom_with_matches / custom_with_expanded_matches / custom_with_match_path
 * / semiglobal_with_prehash (banded.rs:294-401, 938-970) once their band exists.  Pair p's band is
 * the n_p + 1 half-open row ranges band_start/band_end[band_off[p] ..]; `mode` applies the same
 * clip overrides as bg_align_banded_batch.  A band whose column ranges are not monotone gets
 * status BG_ERR_UNSUPPORTED for that pair. */
int bg_align_banded_bands_batch(bg_ctx* ctx, const bg_scoring_t* sc, int mode, uint64_t n_pairs,
                                const uint8_t* x, const uint64_t* x_off, const uint8_t* y,
                                const uint64_t* y_off, const uint64_t* band_off,
                                const uint32_t* band_start, const uint32_t* band_end,
                                bg_alignment_t* out, uint8_t* ops_buf, uint64_t ops_cap,
                                uint64_t* ops_used, uint64_t* band_cells);

/* Band::create_with_matches (banded.rs:1301-1328; path == NULL) or Band::create_from_match_path
 * (banded.rs:1330-1367) for a batch, on host threads.  Matches are (x, y) uint32 pairs, sorted;
 * pair p owns matches_xy[2*match_off[p] .. 2*match_off[p+1]) and, if given, path[path_off[p] ..
 * path_off[p+1]) (indices into its matches).  BG_ERR_INVALID_ARG where the reference asserts
 * (unsorted matches) or indexes out of bounds. */
int bg_band_from_matches_batch(const bg_scoring_t* sc, int mode, uint32_t k, uint32_t w,
                               uint64_t n_pairs, const uint64_t* x_off, const uint64_t* y_off,
                               const uint32_t* matches_xy, const uint64_t* match_off,
                               const uint32_t* path, const uint64_t* path_off,
                               const uint64_t* band_off, uint32_t* start, uint32_t* end,
                               uint64_t* band_cells);

/* sparse.rs on the host: find_kmer_matches (337-348), sdpkpp path (188-295), lcskpp path + score
 * (67-143), sdpkpp_union_lcskpp_path (297-329), expand_kmer_matches (404-500).  Each returns the
 * length of its result (call again with a larger buffer if it exceeds `cap`), or UINT64_MAX where
 * the reference asserts ("incoming matches must be sorted"). */
uint64_t bg_sparse_find_kmer_matches(const uint8_t* x, uint64_t m, const uint8_t* y, uint64_t n,
                                     uint32_t k, uint32_t* out_xy, uint64_t cap);
uint64_t bg_sparse_sdpkpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k,
                          uint32_t match_score, int32_t gap_open, int32_t gap_extend,
                          uint32_t* path, uint64_t cap);
uint64_t bg_sparse_lcskpp(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k, uint32_t* path,
                          uint64_t cap, uint32_t* score);
uint64_t bg_sparse_sdpkpp_union_lcskpp_path(const uint32_t* matches_xy, uint64_t n_matches, uint32_t k,
                                            uint32_t match_score, int32_t gap_open,
                                            int32_t gap_extend, uint32_t* path, uint64_t cap);
uint64_t bg_sparse_expand_kmer_matches(const uint8_t* x, uint64_t m, const uint8_t* y, uint64_t n,
                                       uint32_t k, const uint32_t* matches_xy, uint64_t n_matches,
                                       uint32_t allowed_mismatches, uint32_t* out_xy, uint64_t cap);

/* ---- seed-and-extend read mapping (BASELINE configs[4]) ----------------------------------------------------
 * rust-bio has no read mapper; its callers compose one from FMIndex::backward_search (fmindex.rs:144-208),
 * Interval::occ (fmindex.rs:75-79) and Aligner::semiglobal (pairwise/mod.rs:954) — the pattern of src/lib.rs:129-165
 * and benches/fmindex.rs:20-38.  bg_seed_extend_batch is that composition for a batch of reads with every
 * intermediate in HBM; its definition (stated on the CPU by oracle/pipeline.cpp out of the oracle's three calls):
 *   seeds        read[o .. o + seed_len) for o = 0, stride, 2 stride, ... while the window fits in the read;
 *   votes        a seed votes when its search is Complete and its interval holds 1 ..= max_occ rows;
 *   proposals    hit position p of the seed at offset o proposes the read start s = p - o; s < 0 or s >= n_text
 *                (the text without its final sentinel) is dropped, equal (read, s) proposals are merged; of the sorted starts
 *                of a read, one within pad / 2 of the last start kept is merged into it as well (the seeds either side of an
 *                indel propose the same locus a few bases apart, and the +- pad window of the first holds both alignments;
 *                pad / 2 = 0: only equal starts merge — the definition of rounds 3-5);
 *   extension    Aligner::semiglobal(x = read, y = text[max(0, s - pad) .. min(n_text, s + read_len + pad)));
 *   best hit     per read the highest score, the smallest s among equal scores; a read without candidates
 *                reports score BG_MIN_SCORE, ref positions UINT64_MAX and no operations.
 * A seed that reaches a byte outside the index's alphabet (where the reference's backward_search panics, fmindex.rs:229)
 * does not vote; the call then returns BG_ERR_OUT_OF_ALPHABET with every read still answered.
 * The index handle needs the text (bg_fm_set_text[_dev]: all n bytes the index was built from, final sentinel
 * included) and a suffix array (bg_fm_set_suffix_array / bg_fm_set_sampled_suffix_array). */
int bg_fm_set_text(bg_fm* fm, const uint8_t* text, uint64_t n);         /* host text, copied to the device */
int bg_fm_set_text_dev(bg_fm* fm, const uint8_t* d_text, uint64_t n);   /* device text, borrowed: must outlive the handle's use */
typedef struct {
    uint32_t seed_len, stride; /* benches/fmindex.rs:21-25 searches 20-mers */
    uint32_t max_occ;          /* a seed with more occurrences does not vote */
    uint32_t pad;              /* text taken on both sides of the proposed placement */
} bg_seed_params_t;
typedef struct {
    bg_alignment_t aln;          /* Aligner::semiglobal(read, window) of the best candidate (y coordinates inside the window) */
    uint64_t window_start;       /* text offset of that window */
    uint64_t ref_start, ref_end; /* window_start + ystart / yend */
    uint32_t n_candidates;       /* distinct proposed starts of this read (all were aligned) */
    uint32_t n_seed_hits;        /* suffix-array rows its voting seeds resolved */
} bg_seed_hit_t;
/* reads: concatenated, n_reads + 1 offsets (reads up to 65535 bases; (seed slots) x max_occ <= 1024 per read).
 * ops_buf (optional) receives the winners' operations back to back in read order, hits[r].aln.ops_off points there. */
int bg_seed_extend_batch(bg_fm* fm, const bg_scoring_t* sc, const bg_seed_params_t* prm, uint64_t n_reads,
                         const uint8_t* reads, const uint64_t* read_off, bg_seed_hit_t* hits, uint8_t* ops_buf,
                         uint64_t ops_cap, uint64_t* ops_used);
/* Device flavour: reads, offsets, hits and (optional) operation slots in HBM; read r's operations end at
 * d_ops + (r + 1) * ops_stride (ops_stride >= 2 * max_read_len + 2 * pad + 4), hits[r].aln.ops_off points at the
 * first.  totals (optional, host, 2 entries): suffix-array rows resolved, candidates aligned.  The call waits twice
 * per 2^20 reads for a few counters that size the next stage; everything else is asynchronous on `stream`.
 * It goes through the handle's ctx (scratch, aligner): the ctx's single-thread rule applies. */
int bg_seed_extend_batch_dev(bg_fm* fm, const bg_scoring_t* sc, const bg_seed_params_t* prm, uint64_t n_reads,
                             const uint8_t* d_reads, const uint64_t* d_read_off, uint32_t max_read_len,
                             bg_seed_hit_t* d_hits, uint8_t* d_ops, uint64_t ops_stride, uint64_t* totals,
                             void* stream);

/* ---- FASTQ ingest and CIGAR emission --------------------------------------
 * bio::io::fastq::Reader::read / Records on a text that is in memory (io/fastq.rs:266-303, 508-527: header
 * line '@id desc', sequence lines up to a line that starts with '+', then as many quality lines as there were
 * sequence lines; every line trimmed with str::trim_end) and Record::check (fastq.rs:388-410).  Records are
 * read until the end of the text or the first ReadError: *status is that error (BG_FASTQ_*), *err_pos the byte
 * offset of the line that raised it (the header line for IncompleteRecord), *n_records the records before it.
 * seq/qual are the concatenated trimmed lines; seq_off/qual_off have n_records+1 entries (seq_off is directly
 * the x_off of bg_align_batch_dev); capacity of seq/qual: `len` bytes, of the offsets: rec_cap+1.
 * Returns BG_ERR_TOO_LARGE if there are more than rec_cap records (n_records says how many). */
enum { BG_FASTQ_OK = 0, BG_FASTQ_MISSING_AT = 1, BG_FASTQ_INCOMPLETE = 2, BG_FASTQ_IO = 3 };   /* ReadError, fastq.rs:113-126 */
enum { BG_FQCHECK_OK = 0, BG_FQCHECK_EMPTY_ID = 1, BG_FQCHECK_NONASCII_SEQ = 2, BG_FQCHECK_INVALID_SEQ = 3,
       BG_FQCHECK_NONASCII_QUAL = 4, BG_FQCHECK_UNEQUAL = 5 };                                   /* CheckError, fastq.rs:129-150 */
typedef struct {
    uint64_t id_off, desc_off;   /* into the text */
    uint64_t seq_off, qual_off;  /* into seq / qual */
    uint32_t id_len, desc_len, seq_len, qual_len;
    int32_t has_desc;            /* 0: Record::desc() is None */
    int32_t check;               /* Record::check(): BG_FQCHECK_* (first failing rule) */
} bg_fastq_record_t;
int bg_fastq_parse(bg_ctx* ctx, const uint8_t* text, uint64_t len, bg_fastq_record_t* recs, uint64_t rec_cap,
                   uint8_t* seq, uint64_t* seq_off, uint8_t* qual, uint64_t* qual_off, uint64_t* n_records,
                   int32_t* status, uint64_t* err_pos);
/* the same with text, records, sequences, qualities and offsets in device memory (n_records/status/err_pos are
 * host pointers; the call synchronises `stream`) */
int bg_fastq_parse_dev(bg_ctx* ctx, const uint8_t* d_text, uint64_t len, bg_fastq_record_t* d_recs,
                       uint64_t rec_cap, uint8_t* d_seq, uint64_t* d_seq_off, uint8_t* d_qual,
                       uint64_t* d_qual_off, uint64_t* n_records, int32_t* status, uint64_t* err_pos,
                       void* stream);
/* bio_types::alignment::Alignment::cigar(hard_clip) (bio-types 1.0, a dependency that is not in the reference
 * tree: restated from its documentation, parity unpinned) for n alignments as returned by bg_align_batch
 * (ops_off into `ops`): xstart as a leading soft/hard clip, runs of '=' 'X' 'D' 'I', xlen - xend as the
 * trailing clip, "" without operations.  out_off has n+1 entries.  BG_ERR_UNSUPPORTED if some alignment has
 * AlignmentMode::Custom (the crate panics; its string is empty), BG_ERR_OPS_CAP if out_cap is too small. */
int bg_cigar_batch(bg_ctx* ctx, uint64_t n, const bg_alignment_t* aln, const uint8_t* ops, uint64_t ops_bytes,
                   int hard_clip, char* out, uint64_t out_cap, uint64_t* out_off);
/* device flavour: one slot of `stride` chars (>= 2 * max n_ops + 24) per alignment, d_len[p] = chars written
 * or a negative bg_status */
int bg_cigar_batch_dev(bg_ctx* ctx, uint64_t n, const bg_alignment_t* d_aln, const uint8_t* d_ops, int hard_clip,
                       char* d_out, uint64_t stride, int32_t* d_len, void* stream);

/* bio_types::alignment::Alignment::pretty(x, y, ncol) (bio-types, not in the reference tree: restated from the crate's
 * source as documented — parity unpinned; rust-bio's tests only print it, e.g. pairwise/banded.rs:1805) for n
 * alignments and the sequences they were computed from: rows x / marks / y ('|' match, '\\' mismatch, '+' insertion,
 * 'x' deletion, ' ' clipped, '-' gap), cut into blocks of ncol columns, each block "x\nmarks\ny\n\n\n".
 * out_off has n + 1 entries.  BG_ERR_UNSUPPORTED where the crate panics (a non-ASCII byte breaks its row-length
 * assert) or the sequences do not have the alignment's xlen / ylen; BG_ERR_OPS_CAP if out_cap is too small. */
int bg_pretty_batch(bg_ctx* ctx, uint64_t n, const bg_alignment_t* aln, const uint8_t* ops, uint64_t ops_bytes,
                    const uint8_t* x, const uint64_t* x_off, const uint8_t* y, const uint64_t* y_off, uint32_t ncol,
                    char* out, uint64_t out_cap, uint64_t* out_off);

/* ------------------------------------------------------------------ several GPUs (comm.hip)
 * north_star: "query batches shard embarrassingly across the 8 GPUs of one node with a single RCCL all-gather over xGMI
 * only to collect per-query scores/intervals".  One process and one bg_ctx per GPU.  rust-bio has no counterpart (a
 * single-process library); the shim's align_batch_sharded / backward_search_sharded (rust/bio-gpu-shim) are built on these.
 *   bg_shard_range     rank's contiguous slice [rank * N / W, (rank + 1) * N / W) of N units
 *   bg_shard_balanced  world + 1 boundaries of contiguous slices of (nearly) equal total cost (sum of DP cells of mixed-
 *                      length pairs, of pattern lengths): boundary r = first unit where the running cost passes r/W of it
 *   bg_comm_unique_id  (one rank) the 128-byte id every rank hands to bg_comm_init — distribute it however the job
 *                      talks (a file, an environment variable, MPI)
 *   bg_comm_init       RCCL communicator of this rank's ctx (ncclCommInitRank; librccl.so is opened at run time:
 *                      BG_ERR_UNSUPPORTED if it is not there)
 *   bg_comm_init_host  host-staged communicator for the ranks of ONE node, named `name` (POSIX shared memory): moves the
 *                      records through host memory.  For what RCCL cannot do — several ranks on one GPU (tests) — and,
 *                      with ctx == NULL, for plain host pointers (no GPU at all)
 *   bg_gather_records  every rank contributes n_local records of rec_bytes bytes (device pointers; host pointers for
 *                      a ctx-less host communicator) and receives all of them in rank order in `all` (capacity: the sum
 *                      of the counts); counts_out (optional, host, world entries) says how many each rank brought.  RCCL:
 *                      one ncclAllGather on `stream` when the shards are equal, grouped broadcasts when they are ragged;
 *                      the call returns when the collective is queued (the counts cost one stream synchronisation).
 *   bg_gather_records_cap  the same with the size of `all` stated in records: the counts (and every rank's all_cap) travel
 *                      first, and when the ranks' records together exceed the smallest all_cap EVERY rank returns
 *                      BG_ERR_OPS_CAP before a single record has moved.
 * The host-staged flavour never leaves a rank behind: a local failure (segment, mapping, copy) is published in the control
 * block and all ranks return it together after the call's last barrier; a barrier that is not completed within 120 s (a
 * rank is gone) returns BG_ERR_HIP with the data segment unmapped and unlinked.  bg_comm_init_host survives a control
 * segment of the same name left behind by a crashed or earlier run (ranks confirm with a nonce that the segment they
 * mapped is the one this run's rank 0 created, and attach again otherwise). */
#define BG_COMM_ID_BYTES 128
typedef struct bg_comm bg_comm;
int bg_shard_range(uint64_t n_units, int rank, int world, uint64_t* lo, uint64_t* hi);
int bg_shard_balanced(const uint64_t* costs, uint64_t n, int world, uint64_t* bounds);
int bg_comm_unique_id(uint8_t* id /* BG_COMM_ID_BYTES */);
int bg_comm_init(bg_ctx* ctx, int rank, int world, const uint8_t* id /* BG_COMM_ID_BYTES */, bg_comm** out);
int bg_comm_init_host(bg_ctx* ctx, int rank, int world, const char* name, bg_comm** out);
int bg_gather_records(bg_comm* comm, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t* counts_out,
                      void* stream);
int bg_gather_records_cap(bg_comm* comm, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t all_cap,
                          uint64_t* counts_out, void* stream);
/* ... for records in HOST memory (the results of the host-buffer entry points): staged through device scratch (sized from
 * the gathered counts) for an RCCL communicator, through the shared segment for a host-staged one; all_cap = records `all`
 * can hold: BG_ERR_OPS_CAP on every rank, nothing written, if the ranks bring more.  Synchronous. */
int bg_gather_records_host(bg_comm* comm, const void* local, uint64_t n_local, uint32_t rec_bytes, void* all, uint64_t all_cap,
                           uint64_t* counts_out);
/* What the communicator is, read back from the library that runs it — so that a multi-GPU line can prove "RCCL saw N
 * ranks" instead of repeating what the caller asked for.  info (5 entries): [0] world as given to bg_comm_init*,
 * [1] ncclCommCount() of the RCCL communicator (0 for a host-staged one: no RCCL involved), [2] ncclCommUserRank() (-1
 * host-staged), [3] path of the last gather: 0 none yet, 1 one ncclAllGather, 2 grouped ncclBroadcasts (ragged shards),
 * 3 host-staged through shared memory, [4] gathers done on this communicator. */
int bg_comm_world(bg_comm* comm, int64_t* info);
int bg_comm_free(bg_comm* comm);

/* Timing of the last *_dev / batch call's kernels on this ctx, measured with HIP events on
 * the stream the kernels ran on (used by bench.py for the roofline line). */
typedef struct {
    float fill_ms, traceback_ms, fm_ms;
    uint32_t fill_launches, traceback_launches, fm_launches;
} bg_timing_t;
int bg_get_timing(bg_ctx* ctx, bg_timing_t* out);
/* Pairs of the last banded call on this ctx that the packed-int16 fill flagged (a band cell below the floor of its strip's
 * 16-bit range) and the int32 kernels recomputed.  Waits for the call's kernels. */
int bg_band_redo_pairs(bg_ctx* ctx, uint64_t* out);
int bg_enable_timing(bg_ctx* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif
