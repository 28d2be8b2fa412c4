"""Oracle pin for SampledSuffixArray (suffix_array.rs:86-184) against the reference's own test
(suffix_array.rs:912-964): for every text and sampling rate, sampled.get(i) == sa[i] for all rows."""
import json
import os

import numpy as np
import pytest

import oracle_py as orc

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampled_sa_kats.json")))
N_ALPHABET = b"ACGTNacgtn"  # alphabets/dna.rs:23-35 n_alphabet()


def build(text, occ_k):
    sa = orc.suffix_array(text)
    b = orc.bwt(text, sa)
    less = orc.less(b, N_ALPHABET + b"$")
    occ = orc.Occ(b, occ_k, N_ALPHABET + b"$")
    return sa, b, less, occ


@pytest.mark.parametrize("case", GOLD["texts"], ids=[c["name"] for c in GOLD["texts"]])
def test_sampled_matches(case):
    text = case["text"].encode()
    sa, b, less, occ = build(text, GOLD["occ_k"])
    for rate in GOLD["rates"]:
        ssa = orc.SampledSuffixArray(sa, text, b, less, occ, rate)
        got = [ssa.get(i) for i in range(len(sa))]
        assert got == [int(v) for v in sa], (case["name"], rate)
        assert ssa.get(len(sa)) is None


def test_doctest():
    d = GOLD["doctest"]
    text = d["text"].encode()
    sa, b, less, occ = build(text, d["occ_k"])
    ssa = orc.SampledSuffixArray(sa, text, b, less, occ, d["rate"])
    assert [ssa.get(i) for i in range(len(sa))] == [int(v) for v in sa]


def test_random_multi_sentinel():
    # the reference's rand_seqs cases: several '$'-terminated random sequences
    rng = np.random.default_rng(11)
    for case in range(40):
        parts = []
        for _ in range(int(rng.integers(1, 8))):
            parts.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(1, 30)))].tobytes() + b"$")
        text = b"".join(parts)
        sa, b, less, occ = build(text, 3)
        for rate in (2, 3, 5, 16):
            ssa = orc.SampledSuffixArray(sa, text, b, less, occ, rate)
            assert [ssa.get(i) for i in range(len(sa))] == [int(v) for v in sa]


def test_interval_occ():
    text = b"GCCTTAACATTATTACGCCTA$"
    sa, b, less, occ = build(text, 3)
    tag, lo, hi, ml = orc.backward_search(b, less, occ, b"TTA")
    assert sorted(orc.interval_occ(lo, hi, sa)) == [3, 9, 12]  # fmindex.rs:125-142
    ssa = orc.SampledSuffixArray(sa, text, b, less, occ, 4)
    assert orc.interval_occ(lo, hi, ssa) == orc.interval_occ(lo, hi, sa)
