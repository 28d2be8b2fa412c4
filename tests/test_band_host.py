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
