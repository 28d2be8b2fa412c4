"""GPU parity tests of the suffix-array lookups (kernel K6 through the C ABI): `Interval::occ`
over a raw and a sampled suffix array — against the reference's own sampled-SA test
(suffix_array.rs:912-964), the CPU oracle, and the raw SA itself."""
import json
import os

import numpy as np
import pytest

import oracle_py as orc
from rust_bio_amd import synth
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMIndex, Interval
from rust_bio_amd.suffix_array import NONE, RawSuffixArray, SampledSuffixArray, suffix_array

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampled_sa_kats.json")))
ALPHA = b"ACGTNacgtn$"


def build(text, k=3):
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, ALPHA)
    return sa, b, ls, FMIndex(b, ls, Occ(b, k, ALPHA))


@pytest.mark.parametrize("case", GOLD["texts"], ids=[c["name"] for c in GOLD["texts"]])
def test_reference_sampled_matches(case):
    text = case["text"].encode()
    sa, b, ls, fm = build(text, GOLD["occ_k"])
    rows = np.arange(len(sa) + 2, dtype=np.uint64)  # two rows past the end -> None
    for rate in GOLD["rates"]:
        ssa = SampledSuffixArray(sa, text, b, rate, fmindex=fm)
        got = ssa.get_batch(rows)
        assert (got[:len(sa)] == sa).all(), (case["name"], rate)
        assert (got[len(sa):] == NONE).all()
        assert ssa.get(len(sa)) is None and ssa.get(0) == int(sa[0])
    raw = RawSuffixArray(sa, fm)
    got = raw.get_batch(rows)
    assert (got[:len(sa)] == sa).all() and (got[len(sa):] == NONE).all()


def test_sample_layout_matches_oracle():
    text = GOLD["texts"][-1]["text"].encode()
    sa, b, ls, fm = build(text)
    occ = orc.Occ(b, 3, ALPHA)
    for rate in (2, 5, 16):
        ours = SampledSuffixArray(sa, text, b, rate)
        sample, erow, epos = orc.SampledSuffixArray(sa, text, b, ls, occ, rate).arrays()
        assert (ours.sample == sample).all() and (ours.extra_rows == erow).all() and (ours.extra_pos == epos).all()


def test_interval_occ_kat():
    # fmindex.rs:125-142: TTA in GCCTTAACATTATTACGCCTA$ -> positions [3, 12, 9] (any order in the doc; SA order here)
    text = b"GCCTTAACATTATTACGCCTA$"
    sa, b, ls, fm = build(text)
    res = fm.backward_search(b"TTA")
    host = res.interval.occ(sa)
    assert sorted(host) == [3, 9, 12]
    assert res.interval.occ(RawSuffixArray(sa, fm)) == host
    assert res.interval.occ(SampledSuffixArray(sa, text, b, 4, fmindex=fm)) == host
    with pytest.raises(Exception):  # "Interval out of range of suffix array"
        Interval(0, len(sa) + 1).occ(RawSuffixArray(sa, fm))
    assert Interval(5, 5).occ(RawSuffixArray(sa, fm)) == []


@pytest.mark.parametrize("rate", [1, 7, 32, 64])
def test_genome_locate_vs_oracle(rate):
    # 300 kbp genome with stray N's (exceptions in the 2-bit stream) and three sentinels
    g = synth.random_dna(300_000, seed=21).copy()
    rng = np.random.default_rng(5)
    g[rng.integers(0, len(g), size=40)] = ord("N")
    g[100_000] = g[200_000] = ord("$")
    text = g.tobytes() + b"$"
    sa, b, ls, fm = build(text, k=32)
    ssa = SampledSuffixArray(sa, text, b, rate, fmindex=fm)
    rows = rng.integers(0, len(sa), size=50_000).astype(np.uint64)
    got = ssa.get_batch(rows)
    assert (got == sa[rows.astype(np.intp)]).all()
    # a sample of the same rows through the oracle's restatement of SampledSuffixArray::get
    occ = orc.Occ(b, 32, ALPHA)
    ossa = orc.SampledSuffixArray(sa, text, b, ls, occ, rate)
    for r in rows[:300]:
        assert ossa.get(int(r)) == int(sa[int(r)])
    # intervals of real patterns: occ over sampled == raw
    pats = [text[p:p + 12] for p in rng.integers(0, len(text) - 13, size=2000)]
    res = fm.backward_search_batch([p for p in pats if b"$" not in p])
    lo = [r.interval.lower for r in res if r.kind == "Complete"]
    hi = [r.interval.upper for r in res if r.kind == "Complete"]
    off, pos = fm.interval_occ_arrays(lo, hi)
    want = np.concatenate([sa[a:b_] for a, b_ in zip(lo, hi)])
    assert (pos == want).all() and int(off[-1]) == len(want)
