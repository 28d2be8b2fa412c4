"""CPU tests of the product's host-side builders (bg_suffix_array / bg_bwt / bg_less in
rust-bio_amd/csrc/host_tables.cpp) against the reference's KATs and against the oracle."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load
from rust_bio_amd import _lib
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.suffix_array import suffix_array

K = load("fm_kats.json")


@pytest.mark.parametrize("case", K["suffix_array"], ids=lambda c: c["text"][:12])
def test_suffix_array_kat(case):
    assert suffix_array(case["text"].encode()).tolist() == case["sa"]


@pytest.mark.parametrize("text", K["sorts_lexically"]["texts"], ids=lambda t: t[:10])
def test_sais_equals_oracle_on_reference_texts(text):
    t = text.encode()
    assert suffix_array(t).tolist() == orc.suffix_array(t).tolist()


def test_sais_equals_oracle_random_multi_sentinel():
    # seeded version of rand_seqs (suffix_array.rs:851-866): reads over ATCGN joined by '$'
    rng = np.random.default_rng(5)
    for i in range(60):
        seqs = []
        for _ in range(int(rng.integers(1, 12))):
            ln = int(rng.integers(0, 10 * i // 6 + 2))
            seqs.append(bytes(rng.choice(list(b"ATCGN"), size=ln).astype(np.uint8)))
        t = b"$".join(seqs) + b"$"
        assert suffix_array(t).tolist() == orc.suffix_array(t).tolist(), t


def test_sais_large_alphabet_and_many_sentinels():
    rng = np.random.default_rng(9)
    # > 256 distinct transformed symbols forces the wide-symbol path
    body = rng.integers(1, 256, size=5000).astype(np.uint8)
    body[rng.random(5000) < 0.08] = 0
    t = bytes(body) + b"\x00"
    assert suffix_array(t).tolist() == orc.suffix_array(t).tolist()


def test_sais_200k_dna():
    from rust_bio_amd import synth
    g = synth.genome(200_000, 3)
    sa = suffix_array(g)
    assert sa.tolist() == orc.suffix_array(g).tolist()
    b = bwt(g, sa)
    assert bytes(b) == bytes(orc.bwt(g, sa))


def test_sentinel_assert():
    with pytest.raises(_lib.SentinelError):
        suffix_array(b"AC#GT$")


def test_bwt_and_less_kats():
    for c in K["bwt"]:
        t = c["text"].encode()
        assert bytes(bwt(t, suffix_array(t))) == c["bwt"].encode()
    t = b"GCCTTAACATTATTACGCCTA$"
    b = bwt(t, suffix_array(t))
    for alpha in (b"ACGTNacgtn", b"ACGT$", b"ACGTRYSWKMBDHVNacgtryswkmbdhvn"):
        assert less(b, alpha).tolist() == orc.less(b, alpha).tolist()
    with pytest.raises(_lib.AlphabetError):
        less(np.array([200], dtype=np.uint8), b"ACGT")


def test_occ_new_panics_on_symbol_beyond_alphabet():
    with pytest.raises(_lib.AlphabetError):
        Occ(np.array([65, 250], dtype=np.uint8), 3, b"ACGT")
