"""GPU parity tests of the FMD-index SMEM kernel (K7 through the C ABI) against the reference's tests
(fmindex.rs:704-880, tests/golden/fmd_kats.json) and the CPU oracle on random reads."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load
from rust_bio_amd import synth
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMDIndex, FMIndex
from rust_bio_amd.suffix_array import suffix_array

pytestmark = pytest.mark.gpu
G = load("fmd_kats.json")
ALPHA = b"ACGTNacgtn"
COMP = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


def revcomp(s):
    return s.translate(COMP)[::-1]


def build(text, k=3):
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, ALPHA)
    return sa, b, ls, FMDIndex(FMIndex(b, ls, Occ(b, k, ALPHA)))


def test_smems_kats():
    orig = G["smems"]["text_orig"].encode()
    sa, b, ls, fmd = build(orig + b"$" + revcomp(orig) + b"$")
    for c in G["smems"]["cases"]:
        r = fmd.smems(c["pattern"].encode(), c["i"], c["l"])
        if c.get("empty"):
            assert r == []
            continue
        iv, pos, ln = r[0]
        f = c["first"]
        assert iv.forward().occ(sa) == f["forward"] and iv.revcomp().occ(sa) == f["revcomp"]
        assert (pos, ln) == (f["pos"], f["len"])
        if "match_size" in f:
            assert iv.match_size == f["match_size"]


def test_all_smems_doctest_and_issue39():
    c = G["all_smems"]
    sa, b, ls, fmd = build(c["text"].encode())
    r = fmd.all_smems(c["pattern"].encode(), c["l"])
    assert [[iv.forward().occ(sa)[0], iv.revcomp().occ(sa)[0], p, ln] for iv, p, ln in r] == c["solutions"]
    d = G["doctest_smems"]
    sa, b, ls, fmd = build(d["text"].encode())
    iv, pos, ln = fmd.smems(d["pattern"].encode(), d["i"], d["l"])[0]
    assert iv.forward().occ(sa) == d["forward"] and iv.revcomp().occ(sa) == d["revcomp"] and (pos, ln) == (d["pos"], d["len"])
    c = G["issue39"]
    sa, b, ls, fmd = build(c["text"].encode())
    read = c["read"].encode()
    res = fmd.smems_batch([read] * len(read), list(range(len(read))), 0)
    for i, r in enumerate(res):
        assert [p for iv, _, _ in r for p in iv.forward().occ(sa)] == [c["read_pos"]], i


@pytest.mark.parametrize("flavour", ["general", "plain_two_dollars", "plain_four_dollars"])
def test_random_reads_vs_oracle(flavour):
    """K7's three instantiations, chosen at launch from the index's symbol classes (bg_fm::h_class): with N in the text the
    general extension; an ACGT text as T$R$ the plain-DNA one with the two '$' rows in registers; two sequences (four '$') the
    plain-DNA one that ranks '$' in its list."""
    rng = np.random.default_rng(13)
    g = synth.random_dna(20_000, seed=8).copy()
    if flavour == "general":
        g[rng.integers(0, len(g), size=10)] = ord("N")
    fwd = g.tobytes()
    if flavour == "plain_four_dollars":
        fwd = fwd[:9_000] + b"$" + fwd[9_000:]
    text = fwd + b"$" + revcomp(fwd) + b"$"
    sa, b, ls, fmd = build(text, k=16)
    ofmd = orc.FMDIndex(b, ls, orc.Occ(b, 16, ALPHA))
    reads, pos = [], []
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for _ in range(400):
        s = int(rng.integers(0, len(fwd) - 120))
        r = np.frombuffer(fwd[s:s + int(rng.integers(20, 120))].replace(b"$", b"A"), dtype=np.uint8).copy()
        nm = int(rng.integers(0, 4))
        r[rng.integers(0, len(r), size=nm)] = acgt[rng.integers(0, 4, size=nm)]
        rb = r.tobytes()
        if rng.random() < 0.5:
            rb = revcomp(rb)
        reads.append(rb)
        pos.append(int(rng.integers(0, len(rb))))
    got = fmd.smems_batch(reads, pos, 5)
    got_all = fmd.all_smems_batch(reads, 8)
    for q, rb in enumerate(reads):
        want = ofmd.smems(rb, pos[q], 5)
        assert [((iv.lower, iv.lower_rev, iv.size, iv.match_size), p, ln) for iv, p, ln in got[q]] == want, q
        want = ofmd.all_smems(rb, 8)
        assert [((iv.lower, iv.lower_rev, iv.size, iv.match_size), p, ln) for iv, p, ln in got_all[q]] == want, q


def test_fmd_needs_dna_text():
    text = b"ACGX$"
    sa = suffix_array(text)
    b = bwt(text, sa)
    fm = FMIndex(b, less(b, b"ACGTXN"), Occ(b, 3, b"ACGTXN"))
    with pytest.raises(AssertionError):
        FMDIndex(fm)


def test_init_interval_and_extensions():
    # fmindex.rs:782-805 test_init_interval
    c = G["init_interval"]
    sa, b, ls, fmd = build(c["text"].encode())
    iv = fmd.init_interval_with(ord(c["a"]))
    assert iv.forward().occ(sa) == c["forward"] and iv.revcomp().occ(sa) == c["revcomp"]
    empty = fmd.init_interval()
    assert fmd.backward_ext(empty, ord(c["a"])) == iv
    assert fmd.forward_ext(empty, ord(c["a"])) == iv
    # random walks vs the oracle
    rng = np.random.default_rng(3)
    g = synth.random_dna(3000, seed=2).tobytes()
    sa, b, ls, fmd = build(g + b"$" + revcomp(g) + b"$", k=8)
    ofmd = orc.FMDIndex(b, ls, orc.Occ(b, 8, ALPHA))
    for _ in range(60):
        a0 = int(rng.choice(list(b"ACGT")))
        iv, oiv = fmd.init_interval_with(a0), ofmd.init_interval_with(a0)
        for _ in range(12):
            assert (iv.lower, iv.lower_rev, iv.size, iv.match_size) == oiv
            if iv.size == 0:
                break
            a1 = int(rng.choice(list(b"ACGTN$")))
            if rng.random() < 0.5:
                iv, oiv = fmd.backward_ext(iv, a1), ofmd.backward_ext(oiv, a1)
            else:
                iv, oiv = fmd.forward_ext(iv, a1), ofmd.forward_ext(oiv, a1)


def test_fmd_index_built_entirely_on_the_device():
    """T$R$ never leaves HBM: bg_suffix_array_dev (two sentinels, ranked by position as transform_text does,
    suffix_array.rs:444-466) -> bg_bwt_dev -> bg_fm_build_dev -> SMEMs, against the oracle on the host-built tables"""
    import torch
    from rust_bio_amd.suffix_array import bwt_dev, suffix_array_dev
    rng = np.random.default_rng(31)
    fwd = synth.random_dna(30_000, seed=12).tobytes()
    text = np.frombuffer(fwd + b"$" + revcomp(fwd) + b"$", dtype=np.uint8)
    d_text = torch.from_numpy(text.copy()).to("cuda:0")
    d_sa = suffix_array_dev(d_text)
    d_b = bwt_dev(d_text, d_sa)
    sa = suffix_array(text)
    b = bwt(text, sa)
    assert (d_sa.cpu().numpy().view(np.uint32).astype(np.uint64) == sa).all() and (d_b.cpu().numpy() == b).all()
    fm = FMIndex.from_device(d_b, 16, ALPHA)
    fmd = FMDIndex(fm)
    ls = less(b, ALPHA)
    ofmd = orc.FMDIndex(b, ls, orc.Occ(b, 16, ALPHA))
    reads = []
    for _ in range(200):
        s = int(rng.integers(0, len(fwd) - 100))
        rb = fwd[s:s + int(rng.integers(25, 100))]
        reads.append(revcomp(rb) if rng.random() < 0.5 else rb)
    got = fmd.all_smems_batch(reads, 10)
    for q, rb in enumerate(reads):
        assert [((iv.lower, iv.lower_rev, iv.size, iv.match_size), p, ln) for iv, p, ln in got[q]] == ofmd.all_smems(rb, 10), q
