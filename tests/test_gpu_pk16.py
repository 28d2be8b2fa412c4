"""GPU parity of K1p (sw_fill_pk16.hip: Aligner::local on short reads, two pairs per lane in packed int16
halves) against the CPU oracle and against the one-pair-per-lane K1 — scores, coordinates and operations."""
import numpy as np
import pytest

import oracle_py as orc
from rust_bio_amd import _lib, synth
from rust_bio_amd.pairwise import MIN_SCORE, Aligner, Scoring, decode_ops

pytestmark = pytest.mark.gpu
CLIPS = dict(xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def related_pairs(rng, n_pairs, m_of, n_of, alpha=ALPHA):
    """x is a mutated window of y (substitutions, one indel) or unrelated; lengths come from the callables"""
    xs, ys = [], []
    for p in range(n_pairs):
        m, n = m_of(p), n_of(p)
        y = alpha[rng.integers(0, len(alpha), size=n)]
        if rng.random() < 0.7 and n >= 4 and m >= 4:
            src = np.resize(y, m + 12)[int(rng.integers(0, 4)):]
            x = src[:m + 8].copy()
            k = int(rng.integers(0, max(1, m // 6)))
            x[rng.integers(0, len(x), size=k)] = alpha[rng.integers(0, len(alpha), size=k)]
            if rng.random() < 0.6:
                c = int(rng.integers(0, len(x) - 1))
                x = np.delete(x, np.arange(c, min(len(x), c + int(rng.integers(1, 6)))))
            if rng.random() < 0.4:
                c = int(rng.integers(0, len(x)))
                x = np.insert(x, c, alpha[rng.integers(0, len(alpha), size=int(rng.integers(1, 5)))])
            x = np.resize(x, m)
        else:
            x = alpha[rng.integers(0, len(alpha), size=m)]
        xs.append(x.astype(np.uint8).tobytes())
        ys.append(y.astype(np.uint8).tobytes())
    return xs, ys


MODES = {"custom": 0, "global": 1, "semiglobal": 2, "local": 3}


def local_vs_oracle(kw, xs, ys, mode="local", clips=None):
    sc = Scoring.from_scores(kw["gap_open"], kw["gap_extend"], kw["match"], kw["mismatch"])
    for c, v in (clips or {}).items():
        setattr(sc, c, v)
    al = Aligner.with_scoring(sc)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    out, ops = al.align_arrays(MODES[mode], x, xo, y, yo)
    oout, oops, stride = orc.align_batch(orc.make_scoring(**kw, **dict(CLIPS, **(clips or {}))), mode, x, xo, y, yo, threads=8)
    for f in ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen", "n_ops"):
        bad = np.nonzero(out[f].astype(np.int64) != oout[f].astype(np.int64))[0]
        assert len(bad) == 0, (f, mode, clips, kw, bad[:5], xs[bad[0]], ys[bad[0]], out[f][bad[0]], oout[f][bad[0]])
    assert (out["status"] == 0).all()
    for p in range(len(xs)):
        want = orc.decode_ops(oops[p * stride:p * stride + int(oout["n_ops"][p])])
        assert decode_ops(out[p], ops) == want, (mode, clips, kw, p, xs[p], ys[p])


BASE = dict(gap_open=-5, gap_extend=-1, match=1, mismatch=-1)


@pytest.mark.parametrize("m", [2, 8, 20, 33, 35, 37, 64, 96, 100, 121, 143, 150, 160, 176, 191, 192])
def test_uniform_read_lengths_every_rows_per_lane(m):
    # equal lengths take the fast launch when some R in 2..12 divides m (rows of m on a lane's last row),
    # the second launch otherwise; 37 pairs leave a partial wavefront and a couple without a partner
    rng = np.random.default_rng(m)
    for n in sorted({m, m + 7, max(1, m // 2), 1, 211}):
        xs, ys = related_pairs(rng, 37, lambda p: m, lambda p: n)
        local_vs_oracle(BASE, xs, ys)


@pytest.mark.parametrize("m", [193, 200, 250, 256, 300, 301, 384])
def test_uniform_read_lengths_two_pairs_per_wavefront(m):
    # 192 < m <= 384: 32 lanes per pair
    rng = np.random.default_rng(m)
    for n in sorted({m, m + 7, m // 2, 1}):
        xs, ys = related_pairs(rng, 13, lambda p: m, lambda p: n)
        local_vs_oracle(BASE, xs, ys)
    xs, ys = related_pairs(rng, 21, lambda p: m - (p % 3) * 17, lambda p: m + (p % 2) * 5)
    local_vs_oracle(BASE, xs, ys)


def test_couples_with_different_lengths_take_two_passes():
    rng = np.random.default_rng(5)
    # pair p and pair p + 4 share lanes: make some couples agree and some not
    lens = [150, 150, 150, 150, 150, 140, 150, 150, 120, 150, 150, 150, 150, 150, 150, 77, 150, 150, 150]
    for n_pairs in (len(lens), 8, 5, 4, 3, 1):
        xs, ys = related_pairs(rng, n_pairs, lambda p: lens[p], lambda p: lens[(p * 7) % len(lens)] + 3)
        local_vs_oracle(BASE, xs, ys)
    xs, ys = synth.ragged_pairs(400, 150, seed=77, min_len=1)
    xs += [b"", b"ACGT", b"", b"A", b"C"]
    ys += [b"", b"", b"ACGT", b"A", b"G"]
    local_vs_oracle(BASE, xs, ys)


def test_scorings_up_to_the_12_bit_bound_and_past_it():
    rng = np.random.default_rng(9)
    xs, ys = related_pairs(rng, 90, lambda p: 150, lambda p: 150)
    xs2, ys2 = related_pairs(rng, 64, lambda p: 48 + (p % 3), lambda p: 60)
    for kw in (dict(gap_open=0, gap_extend=0, match=1, mismatch=-1), dict(gap_open=-1, gap_extend=0, match=2, mismatch=0),
               dict(gap_open=0, gap_extend=-1, match=1, mismatch=-3), dict(gap_open=-11, gap_extend=-4, match=13, mismatch=-12),
               dict(gap_open=-13, gap_extend=-13, match=13, mismatch=-13),  # 13 * 152 = 1976: the last magnitude that fits
               dict(gap_open=-14, gap_extend=-2, match=14, mismatch=-9),    # past the bound: K1 takes over
               dict(gap_open=-3, gap_extend=-1, match=0, mismatch=-2), dict(gap_open=-2, gap_extend=-2, match=5, mismatch=0)):
        local_vs_oracle(kw, xs, ys)
        local_vs_oracle(kw, xs2, ys2)


def test_other_alphabets_and_low_complexity():
    rng = np.random.default_rng(10)
    for alpha in (np.frombuffer(b"AC", dtype=np.uint8), np.frombuffer(b"A", dtype=np.uint8), np.arange(256, dtype=np.uint8)):
        xs, ys = related_pairs(rng, 50, lambda p: 100, lambda p: 130, alpha=alpha)
        local_vs_oracle(BASE, xs, ys)
        local_vs_oracle(dict(gap_open=-2, gap_extend=-1, match=2, mismatch=-1), xs, ys)


def test_same_answers_as_k1_at_scale():
    # 200k x 150 bp: every byte of the result equals the one-pair-per-lane kernel's (which the oracle pins)
    x, xo, y, yo = synth.sw_pairs(200_000, 150, seed=3)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    out, ops = al.align_arrays(3, x, xo, y, yo)
    al.ctx.set_option("no_pk16", 1)
    out1, ops1 = al.align_arrays(3, x, xo, y, yo)
    al.ctx.set_option("no_pk16", 0)
    assert out.tobytes() == out1.tobytes()
    assert (ops == ops1).all()
    # a batch that is mostly uniform with stragglers: both launches and the second pass contribute
    rng = np.random.default_rng(4)
    xs, ys = related_pairs(rng, 30_000, lambda p: 150 if p % 97 else 149 - (p % 40), lambda p: 150 if p % 89 else 163)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    out, ops = al.align_arrays(3, x, xo, y, yo)
    al.ctx.set_option("no_pk16", 1)
    out1, ops1 = al.align_arrays(3, x, xo, y, yo)
    al.ctx.set_option("no_pk16", 0)
    assert out.tobytes() == out1.tobytes()
    assert (ops == ops1).all()


@pytest.mark.parametrize("mode", ["semiglobal", "global"])
@pytest.mark.parametrize("m", [8, 37, 60, 100, 150, 192, 250, 384])
def test_other_modes_uniform_and_ragged(mode, m):
    rng = np.random.default_rng(m + len(mode))
    for n in sorted({m, m + 16, max(1, m - 9)}):
        xs, ys = related_pairs(rng, 29, lambda p: m, lambda p: n)
        local_vs_oracle(BASE, xs, ys, mode)
    xs, ys = related_pairs(rng, 40, lambda p: max(1, m - (p % 5) * 3), lambda p: m + 11 - (p % 4))
    xs += [b"", b"ACGT", b"", b"A"]
    ys += [b"", b"", b"ACGT", b"C"]
    local_vs_oracle(BASE, xs, ys, mode)
    local_vs_oracle(dict(gap_open=-3, gap_extend=-2, match=2, mismatch=-3), xs, ys, mode)


def test_custom_clip_patterns_uniform_reads():
    rng = np.random.default_rng(21)
    xs, ys = related_pairs(rng, 48, lambda p: 150, lambda p: 166)
    xs2, ys2 = related_pairs(rng, 40, lambda p: 96 - (p % 3), lambda p: 90 + (p % 7))
    M = MIN_SCORE
    for clips in (dict(xclip_prefix=-3, xclip_suffix=-4, yclip_prefix=-2, yclip_suffix=0),
                  dict(xclip_prefix=0, xclip_suffix=M, yclip_prefix=M, yclip_suffix=-1),
                  dict(xclip_prefix=M, xclip_suffix=0, yclip_prefix=0, yclip_suffix=M),
                  dict(xclip_prefix=M, xclip_suffix=-7, yclip_prefix=-9, yclip_suffix=M),
                  dict(xclip_prefix=-1, xclip_suffix=-1, yclip_prefix=M, yclip_suffix=M),
                  dict(xclip_prefix=0, xclip_suffix=0, yclip_prefix=-5, yclip_suffix=-5),
                  dict(xclip_prefix=-12, xclip_suffix=0, yclip_prefix=0, yclip_suffix=-12)):
        for kw in (BASE, dict(gap_open=0, gap_extend=-1, match=2, mismatch=-2), dict(gap_open=-4, gap_extend=0, match=1, mismatch=0)):
            local_vs_oracle(kw, xs, ys, "custom", clips)
            local_vs_oracle(kw, xs2, ys2, "custom", clips)


def test_semiglobal_same_answers_as_k1_at_scale():
    rng = np.random.default_rng(8)
    xs, ys = related_pairs(rng, 40_000, lambda p: 150 if p % 101 else 140, lambda p: 166 if p % 67 else 150)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    for mode in (2, 1):
        out, ops = al.align_arrays(mode, x, xo, y, yo)
        al.ctx.set_option("no_pk16", 1)
        out1, ops1 = al.align_arrays(mode, x, xo, y, yo)
        al.ctx.set_option("no_pk16", 0)
        assert out.tobytes() == out1.tobytes()
        assert (ops == ops1).all()


def test_ragged_batches_visit_pairs_in_length_order():
    # slots in (m, n) order (counting sort on the device), results per pair: same bytes with and without it
    rng = np.random.default_rng(31)
    xs, ys = related_pairs(rng, 20_000, lambda p: int(rng.integers(90, 151)), lambda p: int(rng.integers(100, 161)))
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    for mode in (3, 2):
        out, ops = al.align_arrays(mode, x, xo, y, yo)
        al.ctx.set_option("no_couples", 1)
        out1, ops1 = al.align_arrays(mode, x, xo, y, yo)
        al.ctx.set_option("no_couples", 0)
        assert out.tobytes() == out1.tobytes() and (ops == ops1).all()
    # device-resident entry (the device decides), small sub-batches
    import torch
    n = len(xs)
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    dxo, dyo = torch.from_numpy(xo.astype(np.int64)).cuda(), torch.from_numpy(yo.astype(np.int64)).cuda()
    stride = 150 + 160 + 4
    d_out = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
    d_ops = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
    al.ctx.set_option("chunk_pairs", 4096)
    al.align_dev(3, n, dx.data_ptr(), dxo.data_ptr(), dy.data_ptr(), dyo.data_ptr(), 150, 160, d_out.data_ptr(), d_ops.data_ptr(), stride,
                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    al.ctx.set_option("chunk_pairs", 0)
    out, _ = al.align_arrays(3, x, xo, y, yo)
    rec = d_out.cpu().numpy().view(_lib.ALN_DTYPE)
    for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"):
        assert (rec[f] == out[f]).all(), f


def test_raw_scorings_with_negative_match_stay_off_the_lf_flavour():
    """the LF cell's unsigned multiply-add needs match >= mismatch; MatchParams::new asserts match >= 0
    (mod.rs:199-200) and so do the wrappers, but a raw bg_scoring_t can say anything: such scorings must take the general
    K1p (same answers as the oracle, which does the reference's arithmetic on whatever it is given)."""
    rng = np.random.default_rng(77)
    xs, ys = related_pairs(rng, 300, lambda p: 40, lambda p: 44)
    for match, mismatch in ((-2, -1), (-1, -3), (-3, -3)):
        sc = Scoring.from_scores(-2, -1, 0, -1)
        object.__setattr__(sc.match_fn, "match_score", match)  # past the constructor's assert (frozen dataclass)
        object.__setattr__(sc.match_fn, "mismatch_score", mismatch)
        sc.match_scores = (match, mismatch)
        al = Aligner.with_scoring(sc)
        x, xo = _lib.concat(xs)
        y, yo = _lib.concat(ys)
        out, ops = al.align_arrays(MODES["local"], x, xo, y, yo)
        kw = dict(gap_open=-2, gap_extend=-1, match=match, mismatch=mismatch)
        oout, oops, stride = orc.align_batch(orc.make_scoring(**kw, **dict(CLIPS, xclip_prefix=0, xclip_suffix=0, yclip_prefix=0,
                                                                           yclip_suffix=0)), "local", x, xo, y, yo, threads=8)
        for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"):
            assert (out[f].astype(np.int64) == oout[f].astype(np.int64)).all(), (f, match, mismatch)
        for p in range(len(xs)):
            assert decode_ops(out[p], ops) == orc.decode_ops(oops[p * stride:p * stride + int(oout["n_ops"][p])])
