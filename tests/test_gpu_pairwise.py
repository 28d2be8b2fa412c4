"""GPU parity tests of the full-matrix aligner (kernels K1 + K2 through the C ABI) against the
reference's KATs and the CPU oracle: score, coordinates and the complete operation list must be
identical for every pair."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import blosum62_matrix, check_expect, load, scoring_kwargs
from rust_bio_amd import _lib, synth
from rust_bio_amd.pairwise import (MIN_SCORE, MODE_NAMES, Aligner, MatchParams, Scoring,
                                   decode_ops)

pytestmark = pytest.mark.gpu
KATS = load("pairwise_kats.json")
MODES = {"custom": 0, "global": 1, "semiglobal": 2, "local": 3}


def engine_scoring(kw):
    if "matrix" in kw:
        fn = kw["matrix"]
        s = Scoring(kw["gap_open"], kw["gap_extend"], fn, None)
    else:
        s = Scoring.from_scores(kw["gap_open"], kw["gap_extend"], kw["match"], kw["mismatch"])
    for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
        setattr(s, c, kw[c])
    return s


def as_dict(a):
    return {"score": a.score, "xstart": a.xstart, "xend": a.xend, "ystart": a.ystart,
            "yend": a.yend, "xlen": a.xlen, "ylen": a.ylen, "ops": a.operations}


@pytest.mark.parametrize("case", KATS["cases"], ids=[c["name"] for c in KATS["cases"]])
def test_reference_kat(case):
    kw = scoring_kwargs(case["scoring"])
    al = Aligner.with_scoring(engine_scoring(kw))
    got = al.align_batch(MODES[case["mode"]], [case["x"].encode()], [case["y"].encode()])[0]
    check_expect(as_dict(got), case["expect"], case["name"])
    assert got.mode == MODE_NAMES[MODES[case["mode"]]]
    # and the whole struct equals the oracle's
    want = orc.align(orc.make_scoring(**kw), case["mode"], case["x"].encode(), case["y"].encode())
    assert as_dict(got) == {k: want[k] for k in as_dict(got)}


def test_aligner_reuse_across_modes():
    r = KATS["aligner_reuse"]
    al = Aligner.with_scoring(engine_scoring(scoring_kwargs(r["scoring"])))
    for step in r["sequence"]:
        fn = {"semiglobal": al.semiglobal, "local": al.local, "global": al.global_}[step["mode"]]
        check_expect(as_dict(fn(r["x"].encode(), r["y"].encode())), step["expect"])


def differential(kw, mode, xs, ys, ctx_opts=None):
    """engine vs oracle on a batch; returns the number of compared pairs"""
    al = Aligner.with_scoring(engine_scoring(kw))
    if ctx_opts:
        for k, v in ctx_opts.items():
            al.ctx.set_option(k, v)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    out, ops = al.align_arrays(MODES[mode], x, xo, y, yo)
    if ctx_opts:
        for k in ctx_opts:
            al.ctx.set_option(k, 0)
    oout, oops, stride = orc.align_batch(orc.make_scoring(**kw), mode, x, xo, y, yo, threads=8)
    for f in ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen"):
        bad = np.nonzero(out[f].astype(np.int64) != oout[f].astype(np.int64))[0]
        assert len(bad) == 0, (f, mode, kw, bad[:5], xs[bad[0]], ys[bad[0]], out[f][bad[0]], oout[f][bad[0]])
    assert (out["status"] == 0).all()
    assert (out["n_ops"] == oout["n_ops"]).all()
    for p in range(len(xs)):
        want = orc.decode_ops(oops[p * stride:p * stride + int(oout["n_ops"][p])])
        got = decode_ops(out[p], ops)
        assert got == want, (mode, kw, p, xs[p], ys[p], got, want)
    return len(xs)


BASE = dict(gap_open=-5, gap_extend=-1, match=1, mismatch=-1, xclip_prefix=MIN_SCORE,
            xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)


@pytest.mark.parametrize("mode", ["local", "semiglobal", "global"])
def test_ragged_batches_including_empty(mode):
    xs, ys = synth.ragged_pairs(600, 70, seed=1)
    xs += [b"", b"ACGT", b"", b"A", b"C"]
    ys += [b"", b"", b"ACGT", b"A", b"G"]
    differential(BASE, mode, xs, ys)


def test_custom_clip_fuzz():
    # the recipe of fuzz/fuzz_targets/banded_aligner.rs:58-105 (random scoring and clip
    # penalties), seeded, against the oracle instead of the stale validator
    rng = np.random.default_rng(2)
    for trial in range(25):
        kw = dict(gap_open=-int(rng.integers(0, 8)), gap_extend=-int(rng.integers(0, 4)),
                  match=int(rng.integers(0, 6)), mismatch=-int(rng.integers(0, 6)))
        for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
            kw[c] = MIN_SCORE if rng.random() < 0.35 else -int(rng.integers(0, 12))
        xs, ys = synth.ragged_pairs(120, 48, seed=100 + trial)
        differential(kw, "custom", xs, ys)


def _tie_rich_pairs(rng, n_pairs, m, n, nalpha, stragglers):
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)[:nalpha]
    xs, ys = [], []
    for p in range(n_pairs):
        mm, nn = m, n
        if stragglers and rng.random() < 0.15:
            mm = int(rng.integers(0, m + 1))
        if stragglers and rng.random() < 0.15:
            nn = int(rng.integers(0, n + 5))
        y = alpha[rng.integers(0, nalpha, size=nn)]
        if rng.random() < 0.6 and nn > 3 and mm > 0:  # a mutated copy: the alignment has a clear diagonal, and repeats
            x = np.resize(y, mm + 6)[int(rng.integers(0, 4)):][:mm].copy()
            k = int(rng.integers(0, max(1, mm // 5)))
            x[rng.integers(0, len(x), size=k)] = alpha[rng.integers(0, nalpha, size=k)]
            x = np.resize(x, mm)
        else:
            x = alpha[rng.integers(0, nalpha, size=mm)]
        xs.append(x.astype(np.uint8).tobytes())
        ys.append(y.astype(np.uint8).tobytes())
    return xs, ys


@pytest.mark.parametrize("mode", ["local", "custom"])
def test_local_fast_flavour_of_k1p(mode):
    """K1p's LF flavour (sw_fill_pk16.inc: no x-suffix-clip fold before column n, the floor 0 by saturation) takes every
    all-zero-clip batch whose gap_open and mismatch are negative: tie-rich inputs (one- and two-letter alphabets, match
    score 0, gap_extend 0), equal lengths (the fast launch) and stragglers (the rest / second launches), against the
    oracle — and the scorings just outside its condition, which stay on the general kernel."""
    rng = np.random.default_rng(77)
    clips = dict(xclip_prefix=0, xclip_suffix=0, yclip_prefix=0, yclip_suffix=0)
    n = 0
    for trial in range(36):
        m = int(rng.choice([150, 100, 36, 61, 120, 7, 250]))
        nn = int(rng.choice([m, m + int(rng.integers(0, 20)), max(1, m - int(rng.integers(0, 20))), int(rng.integers(1, 200))]))
        kw = dict(gap_open=-int(rng.integers(1, 6)), gap_extend=-int(rng.integers(0, 3)),
                  match=int(rng.integers(0, 4)), mismatch=-int(rng.integers(1, 5)), **clips)
        if trial % 9 == 7:
            kw["gap_open"] = 0  # not LF
        if trial % 9 == 8:
            kw["mismatch"] = 0  # not LF
        xs, ys = _tie_rich_pairs(rng, int(rng.integers(1, 90)), m, nn, int(rng.integers(1, 5)), trial % 2 == 1)
        n += differential(kw, mode, xs, ys)
    assert n > 1000


def test_local_fast_flavour_equals_the_general_kernel():
    rng = np.random.default_rng(78)
    kw = dict(gap_open=-3, gap_extend=-1, match=1, mismatch=-2, xclip_prefix=0, xclip_suffix=0, yclip_prefix=0, yclip_suffix=0)
    xs, ys = _tie_rich_pairs(rng, 700, 150, 150, 4, True)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    res = []
    for off in (0, 1):
        al = Aligner.with_scoring(engine_scoring(kw))
        al.ctx.set_option("no_local_fast", off)
        out, ops = al.align_arrays(MODES["local"], x, xo, y, yo)
        al.ctx.set_option("no_local_fast", 0)
        res.append((out, [decode_ops(out[p], ops) for p in range(len(xs))]))
    for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops", "status"):
        assert (res[0][0][f] == res[1][0][f]).all(), f
    assert res[0][1] == res[1][1]


def test_rows_per_lane_configs_and_wave_packing():
    # lengths chosen to hit every (LP, R) instantiation and partial waves
    for max_len, n_pairs in [(20, 33), (60, 17), (90, 9), (120, 13), (150, 70), (190, 5),
                             (230, 6), (300, 3), (380, 5), (500, 3)]:
        xs, ys = synth.ragged_pairs(n_pairs, max_len, seed=max_len, min_len=max_len // 2)
        xs[0] = xs[0][:1] if xs[0] else b"A"
        differential(BASE, "local", xs, ys)
        differential(BASE, "global", xs, ys)


def test_multi_strip_long_sequences():
    # x longer than one strip (64 lanes x 8 rows): the row buffer hand-over between strips
    xs, ys = synth.ragged_pairs(3, 1500, seed=8, min_len=900)
    differential(BASE, "semiglobal", xs, ys)
    kw = dict(BASE, xclip_prefix=-3, xclip_suffix=-4, yclip_prefix=-2, yclip_suffix=0)
    differential(kw, "custom", xs, ys)


def test_wide_kernels_forced_and_by_magnitude():
    # the NARROW (score<<3|priority key) kernels are chosen when every score fits 28 bits; the
    # wide kernels must give the same answers, and huge magnitudes must select them by themselves
    xs, ys = synth.ragged_pairs(200, 90, seed=31)
    differential(BASE, "local", xs, ys, ctx_opts={"force_wide": 1})
    kw = dict(BASE, xclip_prefix=-3, xclip_suffix=-4, yclip_prefix=-2, yclip_suffix=0)
    differential(kw, "custom", xs, ys, ctx_opts={"force_wide": 1})
    differential(BASE, "semiglobal", xs, ys, ctx_opts={"force_wide": 1})
    big = dict(gap_open=-3_000_000, gap_extend=-700_000, match=1_000_000, mismatch=-2_000_000,
               xclip_prefix=MIN_SCORE, xclip_suffix=-5_000_000, yclip_prefix=-1_000_000, yclip_suffix=MIN_SCORE)
    differential(big, "custom", xs, ys)
    differential(big, "local", xs, ys)


def test_sub_batching_reuses_scratch():
    xs, ys = synth.ragged_pairs(300, 40, seed=12)
    differential(BASE, "local", xs, ys, ctx_opts={"chunk_pairs": 64})


def test_pipelined_host_buffer_path():
    # bg_align_batch cuts large batches into stages that flow through three staging sets (upload, kernels, download,
    # compaction overlap); tiny stages force many of them, a partial last one included
    xs, ys = synth.ragged_pairs(1000, 60, seed=14)
    xs += [b"", b"ACGT", b""]
    ys += [b"", b"", b"ACGT"]
    differential(BASE, "local", xs, ys, ctx_opts={"host_chunk_pairs": 64})
    differential(BASE, "semiglobal", xs, ys, ctx_opts={"host_chunk_pairs": 100})
    kw = dict(BASE, xclip_prefix=-3, xclip_suffix=-4, yclip_prefix=-2, yclip_suffix=0)
    differential(kw, "custom", xs, ys, ctx_opts={"host_chunk_pairs": 333})
    # and at the default stage size: same bytes as one serial batch
    x, xo, y, yo = synth.sw_pairs(280_000, 150, seed=5)
    al = Aligner.with_scoring(engine_scoring(BASE))
    out, ops = al.align_arrays(3, x, xo, y, yo)
    al.ctx.set_option("host_chunk_pairs", 1 << 30)
    out1, ops1 = al.align_arrays(3, x, xo, y, yo)
    al.ctx.set_option("host_chunk_pairs", 0)
    assert out.tobytes() == out1.tobytes() and (ops == ops1).all()


def test_pipelined_host_buffer_path_with_a_short_operations_buffer():
    """the caller's ops buffer ends inside a stage: BG_ERR_OPS_CAP, ops_used reports the need, every record is there, and the
    operations of the pairs that fit completely are the ones a large enough buffer receives"""
    import ctypes as C
    x, xo, y, yo = synth.sw_pairs(3000, 100, seed=8)
    al = Aligner.with_scoring(engine_scoring(BASE))
    al.ctx.set_option("host_chunk_pairs", 256)
    try:
        out, ops = al.align_arrays(3, x, xo, y, yo)
        need = int(out["n_ops"].sum())
        cap = need // 2 + 17
        out2 = np.zeros(3000, dtype=_lib.ALN_DTYPE)
        ops2 = np.full(cap + 64, 0xEE, dtype=np.uint8)
        used = C.c_uint64(0)
        sc = al.scoring.to_c()
        xo64, yo64 = np.ascontiguousarray(xo, dtype=np.uint64), np.ascontiguousarray(yo, dtype=np.uint64)
        rc = _lib.lib().bg_align_batch(al.ctx.h, C.byref(sc), 3, 3000, x.ctypes.data, xo64.ctypes.data, y.ctypes.data, yo64.ctypes.data,
                                       out2.ctypes.data, ops2.ctypes.data, cap, C.byref(used))
    finally:
        al.ctx.set_option("host_chunk_pairs", 0)
    assert rc == -9 and used.value == need  # BG_ERR_OPS_CAP
    assert out2.tobytes() == out.tobytes()
    fits = (out["ops_off"].astype(np.int64) + out["n_ops"].astype(np.int64)) <= cap
    last = int(np.nonzero(fits)[0].max())
    end = int(out["ops_off"][last]) + int(out["n_ops"][last])
    assert (ops2[:end] == ops[:end]).all() and (ops2[cap:] == 0xEE).all()


def test_blosum62_protein_batches():
    rng = np.random.default_rng(4)
    aa = b"ARNDCQEGHILKMFPSTWYVBZX"
    xs = [bytes(rng.choice(list(aa), size=int(rng.integers(1, 120))).astype(np.uint8)) for _ in range(150)]
    ys = [bytes(rng.choice(list(aa), size=int(rng.integers(1, 120))).astype(np.uint8)) for _ in range(150)]
    kw = dict(gap_open=-10, gap_extend=-1, matrix=blosum62_matrix(), xclip_prefix=MIN_SCORE,
              xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)
    for mode in ("local", "global", "semiglobal"):
        differential(kw, mode, xs, ys)


def test_closure_with_many_classes_uses_global_table():
    # a match function whose 256 bytes are all distinct classes (A > 64)
    fn = np.fromfunction(lambda a, b: ((a * 7 + b * 13) % 11) - 5, (256, 256), dtype=np.int64).astype(np.int32)
    rng = np.random.default_rng(6)
    xs = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 60))).astype(np.uint8)) for _ in range(60)]
    ys = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 60))).astype(np.uint8)) for _ in range(60)]
    kw = dict(gap_open=-4, gap_extend=-2, matrix=fn, xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE,
              yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)
    differential(kw, "local", xs, ys)
    differential(kw, "global", xs, ys)


@pytest.mark.parametrize("m_len", [97, 128, 129, 160, 161, 192, 193, 384, 400])
def test_tabulated_scoring_at_every_row_geometry(m_len):
    """K1's rows-per-lane choice follows the longest x; every choice must exist for the LDS table (protein), the
    table beyond 64 classes (a closure over all bytes) and wide scores — 101-160 bases once had no instantiation for
    the last two (the reference's banded-vs-full tests run 100-base pairs through a closure)."""
    rng = np.random.default_rng(m_len)
    fn = np.fromfunction(lambda a, b: np.where(a == b, 1, -1), (256, 256), dtype=np.int64).astype(np.int32)
    xs = [bytes(rng.integers(65, 69, size=m_len).astype(np.uint8)) for _ in range(6)] + [b"A" * m_len]
    ys = [bytes(rng.integers(65, 69, size=int(rng.integers(1, m_len + 30))).astype(np.uint8)) for _ in range(7)]
    clips = dict(xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)
    for mode in ("local", "global", "semiglobal"):
        differential(dict(gap_open=-5, gap_extend=-1, matrix=fn, **clips), mode, xs, ys)
        differential(dict(gap_open=-5, gap_extend=-1, matrix=fn, **clips), mode, xs, ys, ctx_opts={"force_wide": 1})
    aa = b"ARNDCQEGHILKMFPSTWYVBZX"
    px = [bytes(rng.choice(list(aa), size=m_len).astype(np.uint8)) for _ in range(4)]
    py = [bytes(rng.choice(list(aa), size=int(rng.integers(1, m_len + 30))).astype(np.uint8)) for _ in range(4)]
    for mode in ("local", "semiglobal"):
        differential(dict(gap_open=-10, gap_extend=-1, matrix=blosum62_matrix(), **clips), mode, px, py)
        differential(dict(gap_open=-10, gap_extend=-1, matrix=blosum62_matrix(), **clips), mode, px, py, ctx_opts={"force_wide": 1})


def test_positive_penalty_is_rejected_like_the_reference_asserts():
    s = Scoring.from_scores(-5, -1, 1, -1)
    s.xclip_prefix = 3
    with pytest.raises(AssertionError):
        Aligner.with_scoring(s)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    al.scoring.gap_open = 2  # bypass the Python-side assert: the C ABI must refuse too
    with pytest.raises(_lib.PenaltyError):
        al.local(b"ACGT", b"ACGT")


def test_cfg2_sample_150bp_local_bit_exact():
    # BASELINE config 2 at a size the oracle finishes in seconds: 20k pairs of 150 bp
    x, xo, y, yo = synth.sw_pairs(20_000, 150, seed=2)
    kw = BASE
    al = Aligner.with_scoring(engine_scoring(kw))
    out, ops = al.align_arrays(3, x, xo, y, yo)
    oout, oops, stride = orc.align_batch(orc.make_scoring(**kw), "local", x, xo, y, yo, threads=8)
    for f in ("score", "xstart", "xend", "ystart", "yend"):
        assert (out[f].astype(np.int64) == oout[f].astype(np.int64)).all(), f
    assert (out["n_ops"] == oout["n_ops"]).all()
    kind = (oops.reshape(len(out), stride) & 0xFF).astype(np.uint8)
    for p in range(len(out)):
        k = int(out["n_ops"][p])
        o = int(out["ops_off"][p])
        assert (ops[o:o + k] == kind[p, :k]).all(), p
    # size-independent property: re-scoring the path reproduces the score
    for p in range(0, len(out), 97):
        sc, i, j = 0, int(out["xstart"][p]), int(out["ystart"][p])
        xs_, ys_ = x[int(xo[p]):int(xo[p + 1])], y[int(yo[p]):int(yo[p + 1])]
        prev = None
        for o in ops[int(out["ops_off"][p]):int(out["ops_off"][p]) + int(out["n_ops"][p])]:
            if o in (0, 1):
                sc += 1 if xs_[i] == ys_[j] else -1
                i += 1
                j += 1
            else:
                sc += -1 if prev == o else -5
                if o == 3:
                    i += 1
                else:
                    j += 1
            prev = o
        assert sc == int(out["score"][p]) and i == int(out["xend"][p]) and j == int(out["yend"][p])


def _dev_align(al, mode, x, xo, y, yo, max_x, max_y, stream=None, dev="cuda:0"):
    """bg_align_batch_dev on host arrays: returns the device tensors (no synchronisation)."""
    import torch
    dx, dy = torch.from_numpy(np.array(x)).to(dev), torch.from_numpy(np.array(y)).to(dev)
    dxo = torch.from_numpy(np.asarray(xo).astype(np.int64)).to(dev)
    dyo = torch.from_numpy(np.asarray(yo).astype(np.int64)).to(dev)
    n = len(xo) - 1
    stride = max_x + max_y + 4
    out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    ops = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    st = stream if stream is not None else torch.cuda.current_stream()
    al.align_dev(mode, n, dx.data_ptr(), dxo.data_ptr(), dy.data_ptr(), dyo.data_ptr(), max_x, max_y, out.data_ptr(),
                 ops.data_ptr(), stride, st.cuda_stream)
    return out, ops, (dx, dy, dxo, dyo)


@pytest.mark.parametrize("pk16", [True, False])
def test_pair_beyond_the_stated_bounds_is_reported_not_computed(pk16):
    """bg_align_batch_dev trusts max_xlen / max_ylen for its scratch geometry: a pair that exceeds them must not be
    aligned into someone else's traceback tile — its record says BG_ERR_INVALID_ARG, every other pair is unaffected."""
    import torch
    from rust_bio_amd import _lib
    ctx = _lib.Context(0)
    ctx.set_option("no_pk16", 0 if pk16 else 1)
    xs, ys = [], []
    rng = np.random.default_rng(5)
    for p in range(300):
        ln = 100 if p not in (17, 130) else 180      # two pairs are longer than the bound stated below
        y = rng.integers(0, 4, size=ln)
        x = y.copy()
        x[rng.integers(0, ln, size=5)] = rng.integers(0, 4, size=5)
        xs.append(np.frombuffer(b"ACGT", dtype=np.uint8)[x].tobytes())
        ys.append(np.frombuffer(b"ACGT", dtype=np.uint8)[y].tobytes())
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)
    out, ops, keep = _dev_align(al, 3, x, xo, y, yo, 100, 100)
    torch.cuda.synchronize()
    rec = out.cpu().numpy().view(_lib.ALN_DTYPE)
    oout, _, _ = orc.align_batch(orc.make_scoring(-5, -1, 1, -1), "local", x, xo, y, yo)
    bad = np.zeros(300, dtype=bool)
    bad[[17, 130]] = True
    assert (rec["status"][bad] == -1).all() and (rec["n_ops"][bad] == 0).all() and (rec["xlen"][bad] == 180).all()
    assert (rec["status"][~bad] == 0).all()
    for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"):
        assert (rec[f][~bad].astype(np.int64) == oout[f][~bad].astype(np.int64)).all(), f


def test_two_streams_one_ctx_do_not_share_scratch_in_flight():
    """Two *_dev calls on two streams with ONE ctx, nothing synchronised in between: the second call waits on the
    device for the first one's last kernel (include/biogpu.h, "Streams and threads"), so both results are right."""
    import torch
    from rust_bio_amd import _lib
    ctx = _lib.Context(0)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    batches = [synth.sw_pairs(60_000, 150, seed=21 + k) for k in range(4)]
    torch.cuda.synchronize()
    res = []
    for k, (x, xo, y, yo) in enumerate(batches):
        st = s1 if k % 2 == 0 else s2
        with torch.cuda.stream(st):
            res.append(_dev_align(al, 3, x, xo, y, yo, 150, 150, stream=st))
    torch.cuda.synchronize()
    for (x, xo, y, yo), (out, ops, _) in zip(batches, res):
        rec = out.cpu().numpy().view(_lib.ALN_DTYPE)
        oout, _, _ = orc.align_batch(orc.make_scoring(-5, -1, 1, -1), "local", x, xo, y, yo, threads=8, want_ops=False)
        for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"):
            assert (rec[f].astype(np.int64) == oout[f].astype(np.int64)).all(), f


def test_cached_matrix_is_invalidated_across_alternating_matrices_and_banded_calls():
    """the compacted matrix stays on the device between calls (own buffer, recognised by hash + a comparison with
    the host copy).  One ctx: matrix A, a banded call under matrix B (the banded path has its own table), matrix A again,
    then B and A back to back — every call must align under ITS matrix."""
    from rust_bio_amd.banded import Aligner as BAligner
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    xs = [acgt[rng.integers(0, 4, size=int(rng.integers(20, 90)))].tobytes() for _ in range(96)]
    ys = [acgt[rng.integers(0, 4, size=int(rng.integers(20, 90)))].tobytes() for _ in range(96)]
    fa = np.fromfunction(lambda a, b: np.where(a == b, 2, -3), (256, 256), dtype=np.int64).astype(np.int32)
    fb = np.full((256, 256), -4, dtype=np.int32)  # transitions cheaper: different scores under the same gaps
    np.fill_diagonal(fb, 5)
    for u, v in ((65, 71), (67, 84)):
        fb[u, v] = fb[v, u] = -1
    clips = dict(xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)
    ctx = _lib.Context(0)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)

    def run(fn):
        kw = dict(gap_open=-5, gap_extend=-1, matrix=fn, **clips)
        al = Aligner.with_scoring(engine_scoring(kw), ctx=ctx)
        out, _ = al.align_arrays(MODES["local"], x, xo, y, yo)
        oout, _, _ = orc.align_batch(orc.make_scoring(**kw), "local", x, xo, y, yo, threads=4)
        assert (out["score"] == oout["score"]).all()
        return out["score"].copy()

    sa = run(fa)
    bal = BAligner.with_scoring(engine_scoring(dict(gap_open=-5, gap_extend=-1, matrix=fb, **clips)), 6, 8, ctx=ctx)
    bal.align_arrays(MODES["local"], x, xo, y, yo)
    assert (run(fa) == sa).all()
    sb = run(fb)
    assert (sb != sa).any()
    assert (run(fa) == sa).all()
    assert (run(fb) == sb).all()
