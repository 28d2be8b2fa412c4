"""`bg_pretty_batch` (bio-types `Alignment::pretty`, parity unpinned) against the oracle's restatement, on the
alignments of the reference's own known-answer tests and on random pairs in every mode."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load, scoring_kwargs
from rust_bio_amd import _lib, synth
from rust_bio_amd.pairwise import Aligner, MatchParams, Scoring, pretty_batch

pytestmark = pytest.mark.gpu
KIND = {"M": 0, "S": 1, "D": 2, "I": 3, "X": 4, "Y": 5}
MODES = {"custom": 0, "global": 1, "semiglobal": 2, "local": 3}


def ops_u64(tokens):
    return [KIND[t[0]] | ((int(t[1:]) if len(t) > 1 else 0) << 8) for t in tokens]


def test_reference_kat_alignments_pretty():
    for case in load("pairwise_kats.json")["cases"]:
        kw = scoring_kwargs(case["scoring"])
        if "matrix" in kw:
            continue
        s = Scoring.from_scores(kw["gap_open"], kw["gap_extend"], kw["match"], kw["mismatch"])
        for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
            setattr(s, c, kw[c])
        x, y = case["x"].encode(), case["y"].encode()
        al = Aligner.with_scoring(s)
        for mode in case.get("modes", [case.get("mode", "custom")]):
            a = al.align_batch(MODES[mode], [x], [y])[0]
            want = orc.pretty({"xstart": a.xstart, "ystart": a.ystart, "mode": mode}, ops_u64(a.operations), x, y, 60)
            assert a.pretty(x, y, 60) == want, (case.get("name"), mode)


@pytest.mark.parametrize("mode", ["custom", "global", "semiglobal", "local"])
def test_random_pairs_batch(mode):
    xs, ys = synth.ragged_pairs(300, 120, seed=7)
    s = Scoring.from_scores(-3, -1, 2, -2)
    if mode == "custom":
        s.xclip(-4).yclip(-2)
    al = Aligner.with_scoring(s)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    out, ops = al.align_arrays(MODES[mode], x, xo, y, yo)
    got = pretty_batch(out, ops, xs, ys, 50)
    for p in range(len(xs)):
        k, o = int(out["n_ops"][p]), int(out["ops_off"][p])
        clip = iter(int(v) for v in out["clip_len"][p])
        toks = [int(v) | ((next(clip) if v >= 4 else 0) << 8) for v in ops[o:o + k]]
        want = orc.pretty({"xstart": int(out["xstart"][p]), "ystart": int(out["ystart"][p]), "mode": mode}, toks, xs[p], ys[p], 50)
        assert got[p] == want, p


def test_non_ascii_byte_is_the_crates_panic():
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    x, y = b"AC\xc3GT", b"AC\xc3GT"
    a = al.align_batch(3, [x], [y])[0]
    with pytest.raises(AssertionError):
        a.pretty(x, y, 80)
    with pytest.raises(AssertionError):
        orc.pretty({"xstart": a.xstart, "ystart": a.ystart, "mode": "local"}, ops_u64(a.operations), x, y, 80)
