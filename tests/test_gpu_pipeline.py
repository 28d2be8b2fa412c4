"""Seed-and-extend (BASELINE configs[4] shape, scaled down): the device pipeline of
rust_bio_amd.pipeline against the same composition made of CPU-oracle calls
(backward_search -> Interval::occ -> Aligner::semiglobal, the caller pattern of
/root/reference/src/lib.rs:129-165)."""
import numpy as np
import pytest
import torch

import oracle_py as orc
from rust_bio_amd import synth
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.pairwise import MIN_SCORE, Aligner, Scoring
from rust_bio_amd.pipeline import seed_and_extend
from rust_bio_amd.suffix_array import RawSuffixArray, SampledSuffixArray, suffix_array

pytestmark = pytest.mark.gpu
ALPHA = b"ACGTNacgtn$"


def oracle_pipeline(text, n_text, sa, b, ls, reads, L, seed_len, stride, max_occ, pad):
    R = len(reads) // L
    occ = orc.Occ(b, 64, ALPHA)
    offs = list(range(0, L - seed_len + 1, stride))
    seeds = np.concatenate([reads[r * L + o: r * L + o + seed_len] for r in range(R) for o in offs])
    poff = np.arange(R * len(offs) + 1, dtype=np.uint64) * seed_len
    tag, lo, hi, ml = orc.backward_search_batch(b, ls, occ, seeds, poff, threads=4)
    cands = []
    for q in range(R * len(offs)):
        if tag[q] == 0 and 0 < hi[q] - lo[q] <= max_occ:
            for p in sa[int(lo[q]):int(hi[q])]:
                s = int(p) - offs[q % len(offs)]
                if 0 <= s < n_text:
                    cands.append((q // len(offs), s))
    cands = sorted(set(cands))
    xs, ys, wl = [], [], []
    for r, s in cands:
        a, e = max(0, s - pad), min(n_text, s + L + pad)
        xs.append(reads[r * L:(r + 1) * L])
        ys.append(text[a:e])
        wl.append(a)
    score = np.full(R, MIN_SCORE, dtype=np.int64)
    rs = np.full(R, -1, dtype=np.int64)
    re = np.full(R, -1, dtype=np.int64)
    if cands:
        xo = np.arange(len(cands) + 1, dtype=np.uint64) * L
        yo = np.zeros(len(cands) + 1, dtype=np.uint64)
        yo[1:] = np.cumsum([len(y) for y in ys])
        out, _, _ = orc.align_batch(orc.make_scoring(-5, -1, 1, -1), "semiglobal", np.concatenate(xs), xo,
                                    np.concatenate(ys), yo, threads=4, want_ops=False)
        for c, (r, s) in enumerate(cands):
            if out["score"][c] > score[r]:
                score[r] = out["score"][c]
                rs[r] = wl[c] + int(out["ystart"][c])
                re[r] = wl[c] + int(out["yend"][c])
    return score, rs, re, len(cands)


@pytest.mark.parametrize("sampled", [0, 8])
def test_seed_and_extend_matches_oracle_composition(sampled):
    n_text, R, L = 200_000, 1500, 150
    g = synth.random_dna(n_text, seed=31).copy()
    g[50_000:50_400] = g[10_000:10_400]  # a repeat: seeds with several occurrences
    text = g.tobytes() + b"$"
    rng = np.random.default_rng(3)
    starts = rng.integers(0, n_text - L, size=R)
    starts[:20] = np.arange(20) * 3           # reads hanging over the left end of the windows
    starts[20:40] = n_text - L - np.arange(20)  # ... and the right end
    refs = np.stack([g[s:s + L] for s in starts])
    reads, _ = synth.mutate_fixed(refs, 77, 0.04, 0.01, 0.01)
    reads[-50:] = synth.random_dna(50 * L, seed=5).reshape(50, L)  # unmappable reads
    reads = np.ascontiguousarray(reads.reshape(-1))

    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, ALPHA)
    fm = FMIndex(b, ls, Occ(b, 64, ALPHA))
    if sampled:
        SampledSuffixArray(sa, text, b, sampled, fmindex=fm)
    else:
        RawSuffixArray(sa, fm)
    dev = torch.device("cuda:0")
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    res = seed_and_extend(fm, al, torch.from_numpy(g).to(dev), n_text, torch.from_numpy(reads).to(dev), R, L)
    torch.cuda.synchronize()
    o_score, o_rs, o_re, o_c = oracle_pipeline(g, n_text, sa, b, ls, reads, L, 20, 10, 16, 25)
    assert res.n_candidates == o_c
    assert (res.score.cpu().numpy().astype(np.int64) == o_score).all()
    assert (res.ref_start.cpu().numpy() == o_rs).all()
    assert (res.ref_end.cpu().numpy() == o_re).all()
    mapped = o_score > MIN_SCORE
    assert mapped[:-50].mean() > 0.95 and not mapped[-50:].any()
    # most reads land where they were drawn from
    assert (np.abs(o_rs[:-50][mapped[:-50]] - starts[:-50][mapped[:-50]]) <= 8).mean() > 0.9
