"""Oracle pin for FMDIndex (fmindex.rs:250-576) against the reference's own tests
(fmindex.rs:704-880 and the doctests)."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load

G = load("fmd_kats.json")
ALPHA = b"ACGTNacgtn"  # dna::n_alphabet(); Occ tabulates '$' by itself (bwt.rs:101-104)
COMP = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


def revcomp(s):
    return s.translate(COMP)[::-1]


def build(text):
    sa = orc.suffix_array(text)
    b = orc.bwt(text, sa)
    ls = orc.less(b, ALPHA)
    occ = orc.Occ(b, G["occ_k"], ALPHA)
    return sa, orc.FMDIndex(b, ls, occ)


def occ_of(iv, sa, rev=False):
    lo = iv[1] if rev else iv[0]
    return [int(sa[p]) for p in range(lo, lo + iv[2])]


def test_smems():
    orig = G["smems"]["text_orig"].encode()
    text = orig + b"$" + revcomp(orig) + b"$"
    sa, fmd = build(text)
    for c in G["smems"]["cases"]:
        r = fmd.smems(c["pattern"].encode(), c["i"], c["l"])
        if c.get("empty"):
            assert r == []
            continue
        iv, pos, ln = r[0]
        f = c["first"]
        assert occ_of(iv, sa) == f["forward"] and occ_of(iv, sa, True) == f["revcomp"]
        assert (pos, ln) == (f["pos"], f["len"])
        if "match_size" in f:
            assert iv[3] == f["match_size"]


def test_all_smems_and_doctest():
    c = G["all_smems"]
    sa, fmd = build(c["text"].encode())
    r = fmd.all_smems(c["pattern"].encode(), c["l"])
    assert len(r) == len(c["solutions"])
    for (iv, pos, ln), sol in zip(r, c["solutions"]):
        assert [occ_of(iv, sa)[0], occ_of(iv, sa, True)[0], pos, ln] == sol
    d = G["doctest_smems"]
    sa, fmd = build(d["text"].encode())
    iv, pos, ln = fmd.smems(d["pattern"].encode(), d["i"], d["l"])[0]
    assert occ_of(iv, sa) == d["forward"] and occ_of(iv, sa, True) == d["revcomp"] and (pos, ln) == (d["pos"], d["len"])


def test_init_interval():
    c = G["init_interval"]
    sa, fmd = build(c["text"].encode())
    iv = fmd.init_interval_with(ord(c["a"]))
    assert occ_of(iv, sa) == c["forward"] and occ_of(iv, sa, True) == c["revcomp"]
    empty = fmd.init_interval()
    assert fmd.backward_ext(empty, ord(c["a"])) == iv
    assert fmd.forward_ext(empty, ord(c["a"])) == iv


def test_issue39():
    c = G["issue39"]
    sa, fmd = build(c["text"].encode())
    read = c["read"].encode()
    for i in range(len(read)):
        matches = [p for iv, _, _ in fmd.smems(read, i, 0) for p in occ_of(iv, sa)]
        assert matches == [c["read_pos"]], i


def test_fmd_requires_dna_alphabet():
    text = b"ACGX$"
    sa = orc.suffix_array(text)
    b = orc.bwt(text, sa)
    with pytest.raises(AssertionError):
        orc.FMDIndex(b, orc.less(b, b"ACGTXN"), orc.Occ(b, 3, b"ACGTXN"))
