// ORACLE (test infrastructure, not product code) — see oracle.h.
//
// Seed-and-extend read mapping composed from the oracle's restatements of the three reference calls a
// rust-bio caller strings together for approximate matching (the caller pattern of
// /root/reference/src/lib.rs:129-165 and benches/fmindex.rs:20-38; BASELINE configs[4]):
//     FMIndex::backward_search(seed)            fmindex.rs:144-208   (orc_backward_search)
//     Interval::occ(&suffix_array)              fmindex.rs:75-79     (a slice of the raw suffix array)
//     Aligner::semiglobal(read, window)         pairwise/mod.rs:954  (orc::Aligner::semiglobal)
// The reference has no such function, so the *definition* of the composition is this repository's
// (include/biogpu.h, bg_seed_extend_batch); the oracle states it with nothing but those three calls:
//   * seeds: read[o .. o + seed_len) for o = 0, stride, 2*stride, ... while the window fits in the read;
//   * a seed votes when its search is Complete and its interval holds 1 ..= max_occ rows;
//   * hit position p of the seed at offset o proposes the read start s = p - o; proposals with s < 0 or
//     s >= n_text are dropped, equal (read, s) proposals are merged, and so are, in ascending order, starts within pad / 2
//     of the last start kept (the seeds either side of an indel propose one locus a few bases apart);
//   * candidate window = text[max(0, s - pad) .. min(n_text, s + read_len + pad)), n_text = the text without
//     its final sentinel; the read is x, the window is y of Aligner::semiglobal;
//   * per read the candidate with the highest score wins, the smallest s among equal scores; a read without
//     candidates reports score MIN_SCORE and ref positions UINT64_MAX.
#include <algorithm>
#include <thread>
#include <vector>

#include "oracle.h"
#include "pairwise_impl.h"

// SaT: the suffix array's element type (uint64_t as `RawSuffixArray = Vec<usize>`, or uint32_t so that a 3 Gbp
// array downloaded from the device need not be widened to 24 GB of host memory)
template <typename SaT>
static int seed_extend_impl(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                            const orc_occ* occ, const SaT* sa, const uint8_t* text, uint64_t n_text,
                            const orc_scoring_t* sc, uint64_t n_reads, const uint8_t* reads,
                            const uint64_t* read_off, uint32_t seed_len, uint32_t stride, uint32_t max_occ,
                            uint32_t pad, orc_seed_hit_t* out, uint64_t* ops, uint64_t ops_stride,
                            int threads) {
    if (threads < 1) threads = 1;
    std::vector<int> rc(threads, 0);
    auto work = [&](int t) {
        orc::Aligner al(orc::scoring_from_c(sc));  // one Aligner per thread (&mut self)
        std::vector<uint64_t> cand;
        const uint64_t lo_r = n_reads * t / threads, hi_r = n_reads * (t + 1) / threads;
        for (uint64_t r = lo_r; r < hi_r; r++) {
            const uint8_t* x = reads + read_off[r];
            const uint64_t L = read_off[r + 1] - read_off[r];
            cand.clear();
            uint64_t seed_hits = 0;
            for (uint64_t o = 0; seed_len > 0 && o + seed_len <= L; o += stride) {
                uint64_t lo, hi, ml;
                int tag = orc_backward_search(bwt, n, less, less_len, occ, x + o, seed_len, &lo, &hi, &ml);
                if (tag == ORC_BS_PANIC) {
                    rc[t] = -2;
                    continue;
                }
                if (tag != ORC_BS_COMPLETE || hi <= lo || hi - lo > max_occ) continue;
                for (uint64_t row = lo; row < hi; row++) {  // Interval::occ
                    const uint64_t p = sa[row];
                    seed_hits++;
                    if (p < o) continue;
                    const uint64_t s = p - o;
                    if (s >= n_text) continue;
                    cand.push_back(s);
                }
            }
            std::sort(cand.begin(), cand.end());
            cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
            {  // starts within pad / 2 of the last one kept are the same locus (its window holds both alignments): merged, in order
                size_t kept = 0;
                for (size_t i = 0; i < cand.size(); i++)
                    if (kept == 0 || cand[i] - cand[kept - 1] > pad / 2) cand[kept++] = cand[i];
                cand.resize(kept);
            }
            orc_seed_hit_t h{};
            h.aln.score = ORC_MIN_SCORE;
            h.ref_start = h.ref_end = h.window_start = UINT64_MAX;
            h.n_candidates = (uint32_t)cand.size();
            h.n_seed_hits = (uint32_t)seed_hits;
            orc::Alignment best;
            bool have = false;
            for (uint64_t s : cand) {
                const uint64_t a = s > pad ? s - pad : 0;
                const uint64_t e = std::min<uint64_t>(n_text, s + L + pad);
                orc::Alignment aln = al.semiglobal(x, L, text + a, e - a);
                if (!have || aln.score > best.score) {
                    best = aln;
                    have = true;
                    h.window_start = a;
                    h.ref_start = a + aln.ystart;
                    h.ref_end = a + aln.yend;
                }
            }
            if (have) {
                if (orc::export_alignment(best, &h.aln, ops ? ops + r * ops_stride : nullptr, ops ? ops_stride : 0) && ops)
                    rc[t] = -1;
            }
            out[r] = h;
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (int r : rc)
        if (r) return r;
    return 0;
}

extern "C" int orc_seed_extend_batch(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                                     const orc_occ* occ, const uint64_t* sa, const uint8_t* text, uint64_t n_text,
                                     const orc_scoring_t* sc, uint64_t n_reads, const uint8_t* reads,
                                     const uint64_t* read_off, uint32_t seed_len, uint32_t stride, uint32_t max_occ,
                                     uint32_t pad, orc_seed_hit_t* out, uint64_t* ops, uint64_t ops_stride,
                                     int threads) {
    return seed_extend_impl<uint64_t>(bwt, n, less, less_len, occ, sa, text, n_text, sc, n_reads, reads, read_off, seed_len, stride,
                                      max_occ, pad, out, ops, ops_stride, threads);
}

extern "C" int orc_seed_extend_batch_sa32(const uint8_t* bwt, uint64_t n, const uint64_t* less, uint64_t less_len,
                                          const orc_occ* occ, const uint32_t* sa, const uint8_t* text, uint64_t n_text,
                                          const orc_scoring_t* sc, uint64_t n_reads, const uint8_t* reads,
                                          const uint64_t* read_off, uint32_t seed_len, uint32_t stride, uint32_t max_occ,
                                          uint32_t pad, orc_seed_hit_t* out, uint64_t* ops, uint64_t ops_stride,
                                          int threads) {
    return seed_extend_impl<uint32_t>(bwt, n, less, less_len, occ, sa, text, n_text, sc, n_reads, reads, read_off, seed_len, stride,
                                      max_occ, pad, out, ops, ops_stride, threads);
}
