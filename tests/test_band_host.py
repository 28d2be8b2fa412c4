"""CPU tests of the product's host band builder (rust-bio_amd/csrc/band_host.cpp through
bg_band_create_batch) against the oracle's Band::create: identical per-column ranges."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load
from rust_bio_amd import _lib, synth
from rust_bio_amd.banded import Aligner
from rust_bio_amd.pairwise import MIN_SCORE, Scoring

CMP = load("banded_compare.json")


def check_bands(scoring, okw, mode, k, w, xs, ys):
    al = Aligner.with_scoring(scoring, k, w, ctx=False)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    boff, st, en, cells = al.band_create_arrays(mode, x, xo, y, yo)
    xp, xs_, yp, ys_ = {0: (okw["xclip_prefix"], okw["xclip_suffix"], okw["yclip_prefix"], okw["yclip_suffix"]),
                        1: (MIN_SCORE,) * 4, 2: (MIN_SCORE, MIN_SCORE, 0, 0), 3: (0, 0, 0, 0)}[mode]
    osc = orc.make_scoring(okw["gap_open"], okw["gap_extend"], okw["match"], okw["mismatch"], xclip_prefix=xp,
                           xclip_suffix=xs_, yclip_prefix=yp, yclip_suffix=ys_,
                           match_scores_some=okw.get("match_scores_some", 1))
    for p in range(len(xs)):
        ost, oen, ocells = orc.band_create(osc, k, w, xs[p], ys[p])
        lo, hi = int(boff[p]), int(boff[p + 1])
        assert (st[lo:hi] == ost).all() and (en[lo:hi] == oen).all(), (p, k, w, xs[p], ys[p])
        assert int(cells[p]) == ocells


BASE = dict(gap_open=-5, gap_extend=-1, match=1, mismatch=-1, xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE,
            yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)


def test_reference_compare_inputs_all_modes():
    xs = [t["x"].encode() for t in CMP["tests"]]
    ys = [t["y"].encode() for t in CMP["tests"]]
    sc = Scoring.new(-5, -1, lambda a, b: 1 if a == b else -1)  # closure: match_scores None
    sc._matrix = None
    for mode in (1, 2, 3):
        check_bands(sc, dict(BASE, match_scores_some=0), mode, 10, 10, xs, ys)


def test_random_pairs_random_k_w():
    rng = np.random.default_rng(17)
    for trial in range(12):
        xs, ys = synth.ragged_pairs(40, 300, seed=500 + trial, min_len=20)
        k, w = int(rng.integers(4, 14)), int(rng.integers(2, 16))
        for mode in (1, 2, 3):
            check_bands(Scoring.from_scores(-5, -1, 1, -1), BASE, mode, k, w, xs, ys)
        kw = dict(BASE, xclip_prefix=-3, xclip_suffix=0, yclip_prefix=0, yclip_suffix=-7, gap_open=-2, gap_extend=-2, match=3)
        s = Scoring.from_scores(-2, -2, 3, -1)
        s.xclip_prefix, s.xclip_suffix, s.yclip_prefix, s.yclip_suffix = -3, 0, 0, -7
        check_bands(s, kw, 0, k, w, xs, ys)


def test_long_reads_10kb_band_geometry():
    # config-4 style pair: 10 kb reference, read with 6 % sub / 2 % ins / 2 % del, k = 16, w = 32
    y = synth.random_dna(10_000, 4)
    xm, lens = synth.mutate_fixed(y.reshape(1, -1), 5, 0.06, 0.02, 0.02)
    x = xm[0][:int(lens[0])]
    check_bands(Scoring.from_scores(-5, -1, 1, -1), BASE, 2, 16, 32, [x.tobytes()], [y.tobytes()])


# ---- sparse.rs helpers of the product (host code behind the custom_with_* entry points) vs the oracle
def _rand_pair(rng, n=300, nsub=25):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    y = acgt[rng.integers(0, 4, size=n)]
    x = y.copy()
    x[rng.integers(0, n, size=nsub)] = acgt[rng.integers(0, 4, size=nsub)]
    cut = int(rng.integers(0, n // 4))
    return np.delete(x, np.arange(cut, cut + int(rng.integers(0, 6)))).tobytes(), y.tobytes()


def test_sparse_helpers_match_oracle():
    from rust_bio_amd import sparse
    rng = np.random.default_rng(17)
    for it in range(60):
        x, y = _rand_pair(rng)
        k = int(rng.integers(4, 9))
        mm = sparse.find_kmer_matches(x, y, k)
        assert mm == [tuple(m) for m in orc.find_kmer_matches(x, y, k).tolist()]
        if not mm:
            continue
        assert sparse.sdpkpp_path(mm, k, 1, -5, -1) == orc.sdpkpp(mm, k, 1, -5, -1)[0]
        assert sparse.lcskpp(mm, k) == orc.lcskpp(mm, k)
        assert sparse.sdpkpp_union_lcskpp_path(mm, k, 1, -5, -1) == orc.sdpkpp_union_lcskpp_path(mm, k, 1, -5, -1)
        for allowed in (0, 1, 2):
            assert sparse.expand_kmer_matches(x, y, k, mm, allowed) == orc.expand_kmer_matches(x, y, k, mm, allowed)


def test_expand_kmer_matches_reference_kat():
    from kat_util import load
    from rust_bio_amd import sparse
    g = load("sparse_kats.json")["expand_kmer_matches"]
    for c in g["cases"]:
        got = sparse.expand_kmer_matches(c["x"].encode(), c["y"].encode(), g["k"], c["matches"], g["allowed_mismatches"])
        assert got == [tuple(m) for m in c["expanded"]]


def test_unsorted_matches_panic():
    from rust_bio_amd import sparse
    with pytest.raises(AssertionError):
        sparse.sdpkpp_path([(5, 5), (1, 1)], 3, 1, -5, -1)
    with pytest.raises(AssertionError):
        sparse.expand_kmer_matches(b"ACGTACGT", b"ACGTACGT", 3, [(5, 5), (1, 1)], 1)


def test_bands_from_matches_match_oracle():
    from rust_bio_amd import sparse
    from rust_bio_amd.banded import Aligner
    from rust_bio_amd.pairwise import Scoring
    rng = np.random.default_rng(23)
    sc = Scoring.from_scores(-5, -1, 1, -1).xclip(-4).yclip_prefix_(-3)
    osc = orc.make_scoring(-5, -1, 1, -1, xclip_prefix=-4, xclip_suffix=-4, yclip_prefix=-3)
    al = Aligner.with_scoring(sc, 6, 5, ctx=False)
    for it in range(25):
        x, y = _rand_pair(rng)
        mm = sparse.find_kmer_matches(x, y, 6)
        xo, yo = [0, len(x)], [0, len(y)]
        _, bs, be, cells = al.bands_from_matches(0, xo, yo, mm, [0, len(mm)])
        ref = orc.banded_align_with(osc, "custom", 6, 5, x, y, mm, want_band=True)
        assert (bs == ref["band"][0]).all() and (be == ref["band"][1]).all() and int(cells[0]) == ref["band_cells"]
        if mm:
            path = sparse.sdpkpp_union_lcskpp_path(mm, 6, 1, -5, -1)
            _, bs, be, cells = al.bands_from_matches(0, xo, yo, mm, [0, len(mm)], path, [0, len(path)])
            ref = orc.banded_align_with(osc, "custom", 6, 5, x, y, mm, path=path, want_band=True)
            assert (bs == ref["band"][0]).all() and (be == ref["band"][1]).all()
