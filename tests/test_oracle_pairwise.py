"""Pins the pairwise ORACLE against every known-answer test the reference holds for
bio::alignment::pairwise::Aligner (tests/golden/pairwise_kats.json, transcribed from
/root/reference/src/alignment/pairwise/mod.rs:21-160 and 1176-1769)."""
import pytest

import oracle_py as orc
from kat_util import check_expect, load, scoring_kwargs

KATS = load("pairwise_kats.json")


@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_reference_kat(case):
    sc = orc.make_scoring(**scoring_kwargs(case["scoring"]))
    got = orc.align(sc, case["mode"], case["x"].encode(), case["y"].encode())
    check_expect(got, case["expect"], case["name"])
    assert got["xlen"] == len(case["x"]) and got["ylen"] == len(case["y"])


def test_aligner_reuse_across_modes():
    # mod.rs:1523-1553 — wrappers restore the clip penalties
    r = KATS["aligner_reuse"]
    sc = orc.make_scoring(**scoring_kwargs(r["scoring"]))
    for step in r["sequence"]:
        got = orc.align(sc, step["mode"], r["x"].encode(), r["y"].encode())
        check_expect(got, step["expect"], step["mode"])


def test_traceback_cell_packing():
    # mod.rs:1176-1200 — nibble layout I=bits0-3, D=4-7, S=8-11 is exercised indirectly by
    # every KAT; here the custom-mode clips check the full 9-code range end to end.
    sc = orc.make_scoring(-5, -1, 1, -1, xclip_prefix=-5, xclip_suffix=-5, yclip_prefix=0,
                          yclip_suffix=0)
    got = orc.align(sc, "custom", b"GAAAA", b"CG")
    assert got["ops"] == ["Y1", "M", "X4"]


def test_empty_inputs():
    sc = orc.make_scoring(-5, -1, 1, -1)
    g = orc.align(sc, "local", b"", b"")
    assert g["score"] == 0 and g["ops"] == []
    g = orc.align(sc, "global", b"ACG", b"")
    assert g["ops"] == ["I", "I", "I"] and g["score"] == -7
    g = orc.align(sc, "global", b"", b"ACG")
    assert g["ops"] == ["D", "D", "D"] and g["score"] == -7
    g = orc.align(sc, "semiglobal", b"", b"ACG")
    assert g["score"] == 0 and g["ops"] == []


def test_batch_matches_single():
    import numpy as np
    rng = np.random.default_rng(7)
    xs = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, 40))).astype(np.uint8))
          for _ in range(50)]
    ys = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, 40))).astype(np.uint8))
          for _ in range(50)]
    x = b"".join(xs)
    y = b"".join(ys)
    xo = np.cumsum([0] + [len(s) for s in xs])
    yo = np.cumsum([0] + [len(s) for s in ys])
    sc = orc.make_scoring(-5, -1, 1, -1)
    for mode in ("local", "semiglobal", "global"):
        out, ops, stride = orc.align_batch(sc, mode, x, xo, y, yo, threads=3)
        for p in range(50):
            one = orc.align(sc, mode, xs[p], ys[p])
            assert one["score"] == out["score"][p]
            assert one["ops"] == orc.decode_ops(ops[p * stride:p * stride + out["n_ops"][p]])
