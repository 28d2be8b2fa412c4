"""What K1p's LF flavour (rust-bio_amd/csrc/sw_fill_pk16.inc) relies on, checked on the CPU against the
restatement of the reference itself: with all four clip penalties 0, gap_open < 0 and mismatch < 0, the x-suffix-clip
fold of the columns before n (mod.rs:793-796 for j < n) changes nothing a caller can see — score, coordinates and
operations are the same without it, and no traceback ever reads Lx[j] of such a column.  The oracle has a test hook
that leaves exactly that out (oracle/pairwise.cpp: g_lf_hook)."""
import numpy as np
import pytest

import oracle_py as orc

ZERO = dict(xclip_prefix=0, xclip_suffix=0, yclip_prefix=0, yclip_suffix=0)


def _pairs(rng, n_pairs, max_len, nalpha):
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)[:nalpha]
    xs, ys = [], []
    for _ in range(n_pairs):
        m, n = int(rng.integers(1, max_len + 1)), int(rng.integers(1, max_len + 1))
        y = alpha[rng.integers(0, nalpha, size=n)]
        if rng.random() < 0.6 and n > 2:  # a mutated, shifted copy: long diagonals, repeats, ties
            x = np.resize(y, m + 5)[int(rng.integers(0, 4)):][:m].copy()
            k = int(rng.integers(0, max(1, m // 4)))
            x[rng.integers(0, len(x), size=k)] = alpha[rng.integers(0, nalpha, size=k)]
            x = np.resize(x, m)
        else:
            x = alpha[rng.integers(0, nalpha, size=m)]
        xs.append(x.astype(np.uint8))
        ys.append(y.astype(np.uint8))
    xo = np.concatenate([[0], np.cumsum([len(v) for v in xs])]).astype(np.uint64)
    yo = np.concatenate([[0], np.cumsum([len(v) for v in ys])]).astype(np.uint64)
    return np.concatenate(xs), xo, np.concatenate(ys), yo


def _run(kw, mode, x, xo, y, yo, hook):
    orc.lf_hook(hook)
    try:
        return orc.align_batch(orc.make_scoring(**kw), mode, x, xo, y, yo, threads=8)
    finally:
        orc.lf_hook(False)


@pytest.mark.parametrize("mode", ["local", "custom"])
def test_local_alignments_do_not_need_the_fold_before_column_n(mode):
    rng = np.random.default_rng(2024)
    reads0 = orc.lf_lx_reads()
    total = 0
    for trial in range(40):
        kw = dict(gap_open=-int(rng.integers(1, 7)), gap_extend=-int(rng.integers(0, 3)), match=int(rng.integers(0, 4)),
                  mismatch=-int(rng.integers(1, 5)), **ZERO)
        x, xo, y, yo = _pairs(rng, 600, int(rng.choice([8, 20, 45, 70])), int(rng.integers(1, 5)))
        full, fops, stride = _run(kw, mode, x, xo, y, yo, False)
        red, rops, _ = _run(kw, mode, x, xo, y, yo, True)  # raises if any traceback asked for an Lx[j < n]
        for f in full.dtype.names:
            assert (full[f] == red[f]).all(), (f, kw)
        assert (fops == rops).all(), kw
        total += len(xo) - 1
    assert total == 24000 and orc.lf_lx_reads() == reads0


@pytest.mark.parametrize("go,ge,ma,mi", [(0, -1, 1, -1), (0, 0, 1, -1), (0, 0, 2, -3), (-1, 0, 1, 0), (-2, -1, 1, 0)])
def test_scorings_outside_the_engines_condition_show_no_difference_either(go, ge, ma, mi):
    """The argument in DESIGN.md §4 needs gap_open < 0 (a gap on top of a fold's value makes it strictly smaller) and the
    kernel's floor-by-saturation needs mismatch < 0, so the engine keeps other scorings on the general kernel; the
    reduced algorithm itself has shown no difference there either (free gaps, free mismatches: seeded sweeps)."""
    rng = np.random.default_rng(11)
    kw = dict(gap_open=go, gap_extend=ge, match=ma, mismatch=mi, **ZERO)
    for trial in range(10):
        x, xo, y, yo = _pairs(rng, 400, int(rng.choice([6, 10, 16, 30])), int(rng.integers(1, 4)))
        full, fops, _ = _run(kw, "local", x, xo, y, yo, False)
        red, rops, _ = _run(kw, "local", x, xo, y, yo, True)
        for f in full.dtype.names:
            assert (full[f] == red[f]).all(), (f, kw)
        assert (fops == rops).all(), kw
