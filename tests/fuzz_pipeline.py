"""Differential fuzz of the seed-and-extend composition (run by hand on a GPU box: python tests/fuzz_pipeline.py SEED SECONDS):
random genomes with planted repeats (a copied segment, a tandem repeat of a short period — runs of proposals a few bases apart),
reads drawn from them with substitutions and indels (fixed and ragged lengths, some unmappable, some hanging over the text's
ends), random seed parameters (seed length, stride, max_occ, pad 0 .. 40: the candidate-merging distance pad / 2 takes every
small value), raw and sampled suffix arrays, one or several passes (ctx option seed_chunk_reads) — bg_seed_extend_batch against
oracle/pipeline.cpp read by read: candidate and hit counts, window, reference span, the winner's record and every operation."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle_py as orc  # noqa: E402
from rust_bio_amd import _lib, synth  # noqa: E402
from rust_bio_amd.bwt import Occ, bwt, less  # noqa: E402
from rust_bio_amd.fmindex import FMIndex  # noqa: E402
from rust_bio_amd.pairwise import MIN_SCORE, Scoring  # noqa: E402
from rust_bio_amd.pipeline import SeedParams, attach_text, seed_extend_arrays  # noqa: E402
from rust_bio_amd.suffix_array import RawSuffixArray, SampledSuffixArray, suffix_array  # noqa: E402

ALPHA = b"ACGTNacgtn$"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0 = time.time()
rounds = n_reads = n_cand = n_fail = 0
while time.time() - t0 < budget and n_fail == 0:
    rounds += 1
    n_text = int(rng.choice([3_000, 40_000, 150_000]))
    L = int(rng.choice([40, 100, 150, 250]))
    if n_text < 4 * L:
        continue
    g = synth.random_dna(n_text, seed=int(rng.integers(1, 1 << 30))).copy()
    if n_text > 20_000:
        a, b_ = int(rng.integers(0, n_text // 2 - 600)), int(rng.integers(n_text // 2, n_text - 600))
        g[b_:b_ + 500] = g[a:a + 500]  # a copied segment: two candidates far apart
    per = int(rng.integers(3, 30))
    t_at = int(rng.integers(0, n_text - 400))
    g[t_at:t_at + 300] = np.tile(g[t_at:t_at + per], 300 // per + 1)[:300]  # a tandem repeat: runs of proposals `per` apart
    text = np.append(g, np.uint8(ord("$")))
    R = int(rng.choice([50, 400, 1500]))
    starts = rng.integers(0, n_text - L, size=R)
    starts[:R // 10] = t_at - L // 2 + rng.integers(0, 300, size=R // 10)  # reads over the tandem repeat
    starts = np.clip(starts, 0, n_text - L)
    starts[:3] = [0, 1, n_text - L]
    refs = np.stack([g[s:s + L] for s in starts])
    sub, ind = float(rng.choice([0.0, 0.02, 0.06])), float(rng.choice([0.0, 0.01, 0.03]))
    reads, _ = synth.mutate_fixed(refs, int(rng.integers(1, 1 << 30)), sub, ind, ind)
    nbad = R // 20
    if nbad:
        reads[-nbad:] = synth.random_dna(nbad * L, seed=int(rng.integers(1, 1 << 30))).reshape(nbad, L)
    if rng.random() < 0.4:
        lens = rng.integers(max(5, L // 4), L + 1, size=R)
        flat = np.concatenate([reads[r, :lens[r]] for r in range(R)])
        off = np.zeros(R + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
    else:
        flat = np.ascontiguousarray(reads.reshape(-1))
        off = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    seed_len = int(rng.choice([12, 16, 20, 24]))
    prm = SeedParams(seed_len=seed_len, stride=int(rng.integers(1, seed_len + 4)), max_occ=int(rng.choice([1, 4, 16])), pad=int(rng.integers(0, 41)))
    if ((L - seed_len) // prm.stride + 1) > 64 or ((L - seed_len) // prm.stride + 1) * prm.max_occ > 1024:
        continue  # (more seed slots / proposals than the engine takes: BG_ERR_UNSUPPORTED by design)
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, ALPHA)
    ctx = _lib.Context(0)
    if rng.random() < 0.5:
        ctx.set_option("seed_chunk_reads", int(rng.integers(1, R + 1)))
    fm = FMIndex(b, ls, Occ(b, 64, ALPHA), ctx=ctx)
    sampled = int(rng.choice([0, 0, 4, 32]))
    if sampled:
        SampledSuffixArray(sa, text, b, sampled, fmindex=fm)
    else:
        RawSuffixArray(sa, fm)
    attach_text(fm, text)
    scs = [(-5, -1, 1, -1), (-4, -2, 2, -3), (-2, -1, 3, -2)][int(rng.integers(0, 3))]
    hits, ops = seed_extend_arrays(fm, Scoring.from_scores(*scs), flat, off, params=prm)
    occ = orc.Occ(b, 64, ALPHA)
    ohits, oops, ostride = orc.seed_extend_batch(b, ls, occ, sa, text, n_text, orc.make_scoring(*scs), flat, off, seed_len=prm.seed_len,
                                                 stride=prm.stride, max_occ=prm.max_occ, pad=prm.pad, threads=8)
    ok = all((hits[f] == ohits[f]).all() for f in ("n_candidates", "n_seed_hits", "window_start", "ref_start", "ref_end"))
    ok = ok and all((hits["aln"][f].astype(np.int64) == ohits["aln"][f].astype(np.int64)).all()
                    for f in ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen", "n_ops"))
    if ok:
        for r in np.nonzero(ohits["aln"]["score"] > MIN_SCORE)[0]:
            k, o = int(hits["aln"]["n_ops"][r]), int(hits["aln"]["ops_off"][r])
            if not (ops[o:o + k] == (oops[r * ostride:r * ostride + k] & np.uint64(0xFF)).astype(np.uint8)).all():
                ok = False
                break
    if not ok:
        n_fail += 1
        print("MISMATCH round", rounds, "n_text", n_text, "L", L, "R", R, "prm", (prm.seed_len, prm.stride, prm.max_occ, prm.pad), "sampled", sampled, "scoring", scs, flush=True)
    n_reads += R
    n_cand += int(ohits["n_candidates"].sum())
    del fm
    ctx.close()
print("rounds", rounds, "reads", n_reads, "candidates", n_cand, "failures", n_fail, flush=True)
sys.exit(1 if n_fail else 0)
