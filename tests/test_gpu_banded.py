"""GPU parity tests of the banded aligner (kernels K3 + K4 through bg_align_banded_batch) against
the reference's KATs and the CPU oracle: whole-Alignment equality for every pair."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import blosum62_matrix, check_expect, load, scoring_kwargs
from rust_bio_amd import _lib, synth
from rust_bio_amd.banded import Aligner
from rust_bio_amd.pairwise import MIN_SCORE, Scoring, decode_ops

pytestmark = pytest.mark.gpu
BK = load("banded_kats.json")
CMP = load("banded_compare.json")
PK = load("pairwise_kats.json")
MODES = {"custom": 0, "global": 1, "semiglobal": 2, "local": 3}


def engine_scoring(kw, some):
    if "matrix" in kw:
        s = Scoring(kw["gap_open"], kw["gap_extend"], kw["matrix"], None)
    else:
        s = Scoring.from_scores(kw["gap_open"], kw["gap_extend"], kw["match"], kw["mismatch"])
        if not some:
            # Scoring::new(closure): same numbers, match_scores = None
            s.match_scores = None
            s.match_fn = np.where(np.eye(256, dtype=bool), kw["match"], kw["mismatch"]).astype(np.int32)
    for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
        setattr(s, c, kw[c])
    return s


def as_dict(a):
    return {"score": a.score, "xstart": a.xstart, "xend": a.xend, "ystart": a.ystart, "yend": a.yend,
            "xlen": a.xlen, "ylen": a.ylen, "ops": a.operations}


def differential(kw, some, mode, k, w, xs, ys, opts=None):
    al = Aligner.with_scoring(engine_scoring(kw, some), k, w)
    for k_, v_ in (opts or {}).items():
        al.ctx.set_option(k_, v_)
    differential.last_ctx = al.ctx
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    try:
        out, ops = al.align_arrays(MODES[mode], x, xo, y, yo)
        engine_failed = False
    except _lib.BiogpuError as e:
        assert e.status in (-10, -11), e  # some pair: traceback does not terminate / panics in the reference
        out, ops = al.last_out, al.last_ops
        engine_failed = True
    for k_ in (opts or {}):
        al.ctx.set_option(k_, 0)  # (the default ctx is shared between Aligners)
    okw = dict(kw)
    okw["match_scores_some"] = 1 if some else 0
    osc = orc.make_scoring(**okw)
    n_bad = 0
    for p in range(len(xs)):
        try:
            want = orc.banded_align(osc, mode, k, w, xs[p], ys[p])
        except RuntimeError:
            # the reference itself panics / loops forever on this pair: the engine must say so
            assert out["status"][p] != 0, (mode, k, w, kw, p, xs[p], ys[p])
            n_bad += 1
            continue
        assert out["status"][p] == 0, (mode, k, w, kw, p, xs[p], ys[p], out[p])
        assert int(al.last_cells[p]) == want["band_cells"]
        got = {"score": int(out["score"][p]), "xstart": int(out["xstart"][p]), "xend": int(out["xend"][p]),
               "ystart": int(out["ystart"][p]), "yend": int(out["yend"][p]), "xlen": int(out["xlen"][p]),
               "ylen": int(out["ylen"][p]), "mode": int(out["mode"][p]), "ops": decode_ops(out[p], ops)}
        want.pop("band_cells")
        assert got == want, (mode, k, w, kw, p, xs[p], ys[p], got, want)
    assert engine_failed == (n_bad > 0)
    return n_bad


@pytest.mark.parametrize("c", BK["cases"], ids=lambda c: c["name"])
def test_reference_kat(c):
    kw = scoring_kwargs(c["scoring"])
    some = not (c["scoring"].get("closure") or "matrix" in kw)
    al = Aligner.with_scoring(engine_scoring(kw, some), c["k"], c["w"])
    got = as_dict(al.align_batch(MODES[c["mode"]], [c["x"].encode()], [c["y"].encode()])[0])
    e = dict(c["expect"])
    if e.pop("yend_is_ylen", False):
        assert got["yend"] == got["ylen"]
    if "x_aln_len" in e:
        assert got["xend"] - got["xstart"] == e.pop("x_aln_len")
        assert got["yend"] - got["ystart"] == e.pop("y_aln_len")
    check_expect(got, e, c["name"])
    differential(kw, some, c["mode"], c["k"], c["w"], [c["x"].encode()], [c["y"].encode()])


@pytest.mark.parametrize("t", CMP["tests"], ids=lambda t: t["name"])
def test_banded_equals_full_inputs(t):
    kw = dict(gap_open=-5, gap_extend=-1, match=1, mismatch=-1, xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE,
              yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)
    for mode in t["modes"]:
        differential(kw, False, mode, 10, 10, [t["x"].encode()], [t["y"].encode()])


def test_full_aligner_kats_through_banded():
    for case in PK["cases"]:
        kw = scoring_kwargs(case["scoring"])
        from_scores = case["name"] in ("test_scoring_from_scores", "test_only_clips") or case["name"].startswith("test_zero_score")
        al = Aligner.with_scoring(engine_scoring(kw, from_scores), 10, 10)
        got = as_dict(al.align_batch(MODES[case["mode"]], [case["x"].encode()], [case["y"].encode()])[0])
        check_expect(got, case["expect"], case["name"])


BASE = dict(gap_open=-5, gap_extend=-1, match=1, mismatch=-1, xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE,
            yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)


@pytest.mark.parametrize("mode", ["semiglobal", "local", "global"])
def test_random_batches(mode):
    rng = np.random.default_rng(5)
    for trial in range(6):
        xs, ys = synth.ragged_pairs(60, 420, seed=900 + trial, min_len=25)
        differential(BASE, True, mode, int(rng.integers(5, 13)), int(rng.integers(3, 14)), xs, ys)


def test_custom_clip_fuzz():
    # fuzz/fuzz_targets/banded_aligner.rs:58-105: k, w in [5,14], random scoring and clip penalties
    rng = np.random.default_rng(8)
    for trial in range(16):
        kw = dict(gap_open=-int(rng.integers(0, 8)), gap_extend=-int(rng.integers(0, 4)),
                  match=int(rng.integers(1, 6)), mismatch=-int(rng.integers(0, 6)))
        for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
            kw[c] = MIN_SCORE if rng.random() < 0.35 else -int(rng.integers(0, 12))
        xs, ys = synth.ragged_pairs(50, 160, seed=3000 + trial, min_len=12)
        differential(kw, bool(trial % 2), "custom", int(rng.integers(5, 15)), int(rng.integers(5, 15)), xs, ys)


def test_long_read_10kb_semiglobal_k16_w32():
    # BASELINE config 4 at oracle-friendly size: 10 kb pairs, k-mer 16, w = 32
    ys, xs = [], []
    for p in range(3):
        y = synth.random_dna(10_000, 40 + p)
        xm, lens = synth.mutate_fixed(y.reshape(1, -1), 50 + p, 0.06, 0.02, 0.02)
        xs.append(xm[0][:int(lens[0])].tobytes())
        ys.append(y.tobytes())
    differential(BASE, True, "semiglobal", 16, 32, xs, ys)


def test_max_cells_sentinel_and_mixed_batch():
    rng = np.random.default_rng(1)
    x = bytes(rng.choice(list(b"AC"), size=2300).astype(np.uint8))
    y = bytes(rng.choice(list(b"GT"), size=2300).astype(np.uint8))
    xs, ys = synth.ragged_pairs(5, 200, seed=77, min_len=30)
    differential(BASE, True, "semiglobal", 16, 8, xs[:2] + [x] + xs[2:], ys[:2] + [y] + ys[2:])


def test_blosum62_banded():
    rng = np.random.default_rng(4)
    aa = b"ARNDCQEGHILKMFPSTWYV"
    xs, ys = [], []
    for _ in range(40):
        y = rng.choice(list(aa), size=int(rng.integers(30, 200))).astype(np.uint8)
        x = y.copy()
        mut = rng.random(len(x)) < 0.1
        x[mut] = rng.choice(list(aa), size=int(mut.sum()))
        xs.append(bytes(x[int(rng.integers(0, 10)):]))
        ys.append(bytes(y))
    kw = dict(gap_open=-10, gap_extend=-1, matrix=blosum62_matrix(), xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE,
              yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE)
    for mode in ("local", "semiglobal"):
        differential(kw, False, mode, 4, 6, xs, ys)


# ---- the entry points that take matches / chains / prehashes from the caller (banded.rs:294-401, 938-970)
def _rand_pair2(rng, n=400, nsub=30):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    y = acgt[rng.integers(0, 4, size=n)]
    x = y.copy()
    x[rng.integers(0, n, size=nsub)] = acgt[rng.integers(0, 4, size=nsub)]
    cut = int(rng.integers(0, n // 4))
    return np.delete(x, np.arange(cut, cut + int(rng.integers(0, 8)))).tobytes(), y.tobytes()


def ops_tokens(al):
    return list(al.operations)


def test_custom_with_matches_family_vs_oracle():
    from rust_bio_amd import sparse
    from rust_bio_amd.banded import Aligner as BAligner
    rng = np.random.default_rng(41)
    sc = Scoring.from_scores(-5, -1, 1, -1).xclip(-6).yclip(-2)
    osc = orc.make_scoring(-5, -1, 1, -1, xclip_prefix=-6, xclip_suffix=-6, yclip_prefix=-2, yclip_suffix=-2)
    k, w = 7, 5
    al = BAligner.with_scoring(sc, k, w)
    for it in range(12):
        x, y = _rand_pair2(rng)
        mm = sparse.find_kmer_matches(x, y, k)
        # custom_with_matches == custom; [] == the full matrix (the fuzz target's reference run)
        a = al.custom_with_matches(x, y, mm)
        r = orc.banded_align_with(osc, "custom", k, w, x, y, mm)
        assert a.score == r["score"] and ops_tokens(a) == r["ops"]
        assert a.score == al.custom(x, y).score and ops_tokens(a) == ops_tokens(al.custom(x, y))
        e = al.custom_with_matches(x, y, [])
        rf = orc.align(osc, "custom", x, y)
        assert e.score == rf["score"] and ops_tokens(e) == rf["ops"]
        if not mm:
            continue
        # custom_with_match_path with the lcskpp chain
        path, _ = sparse.lcskpp(mm, k)
        a = al.custom_with_match_path(x, y, mm, path)
        r = orc.banded_align_with(osc, "custom", k, w, x, y, mm, path=path)
        assert a.score == r["score"] and ops_tokens(a) == r["ops"]
        # custom_with_expanded_matches, all four flag combinations
        for allowed in (None, 1):
            for union in (False, True):
                a = al.custom_with_expanded_matches(x, y, mm, allowed, union)
                r = orc.banded_align_with(osc, "custom", k, w, x, y, mm, expanded=True, allowed_mismatches=allowed,
                                          use_lcskpp_union=union)
                assert a.score == r["score"] and ops_tokens(a) == r["ops"], (it, allowed, union)


def test_with_prehash_equals_plain_calls():
    # banded.rs:53,89 (doctest) and 1464-1466: *_with_prehash(x, y, hash_kmers(y, k)) == the plain call
    from rust_bio_amd import sparse
    from rust_bio_amd.banded import Aligner as BAligner
    x = b"AGCACACGTGTGCGCTATACAGTAAGTAGTAGTACACGTGTCACAGTTGTACTAGCATGAC"
    y = b"AGCACACGTGTGCGCTATACAGTACACGTGTCACAGTTGTACTAGCATGAC"
    al = BAligner.new(-5, -1, lambda a, b: 1 if a == b else -1, 8, 6)
    h = sparse.hash_kmers(y, 8)
    s1, s2 = al.semiglobal(x, y), al.semiglobal_with_prehash(x, y, h)
    assert s1.score == s2.score and ops_tokens(s1) == ops_tokens(s2) and s2.mode == s1.mode
    c1, c2 = al.custom(x, y), al.custom_with_prehash(x, y, h)
    assert c1.score == c2.score and ops_tokens(c1) == ops_tokens(c2)


def test_device_band_builder_equals_host_builder():
    """Band::create on the device (band_device.hip) vs on host threads (band_host.cpp): same bands (cell
    counts), same alignments — including pairs the device path hands back to the host (repeats)."""
    from rust_bio_amd.banded import Aligner as BAligner
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    xs, ys = [], []
    for it in range(96):
        n = int(rng.integers(50, 3000))
        y = acgt[rng.integers(0, 4, size=n)]
        if it % 7 == 0:  # a tandem repeat: many matches per k-mer
            unit = acgt[rng.integers(0, 4, size=int(rng.integers(2, 9)))]
            y[n // 3:n // 3 + 300] = np.resize(unit, 300)[:len(y[n // 3:n // 3 + 300])]
        x = y.copy()
        nsub = max(1, n // 15)
        x[rng.integers(0, n, size=nsub)] = acgt[rng.integers(0, 4, size=nsub)]
        cut = int(rng.integers(0, max(1, n - 20)))
        x = np.delete(x, np.arange(cut, cut + int(rng.integers(0, 12))))
        if it % 5 == 0:
            x = x[int(rng.integers(0, 30)):]
        if it % 11 == 0:
            x = acgt[rng.integers(0, 4, size=int(rng.integers(5, 200)))]  # unrelated: few or no matches
        xs.append(x.tobytes())
        ys.append(y.tobytes())
    for mode, sc in [(2, Scoring.from_scores(-5, -1, 1, -1)), (3, Scoring.from_scores(-4, -2, 2, -3)),
                     (0, Scoring.from_scores(-5, -1, 1, -1).xclip(-7).yclip_prefix_(-3).yclip_suffix_(0)),
                     (1, Scoring.new(-6, -1, lambda a, b: 2 if a == b else -2))]:
        for k, w in [(8, 6), (11, 20), (5, 3), (20, 12)]:  # (k beyond 16: the LDS join compares its tail byte by byte)
            al = BAligner.with_scoring(sc, k, w)
            x, xo = _lib.concat(xs)
            y, yo = _lib.concat(ys)
            al.ctx.set_option("band_on_host", 1)
            try:
                out_h, ops_h = al.align_arrays(mode, x, xo, y, yo)
                cells_h = al.last_cells.copy()
            except Exception as e:  # noqa: BLE001 - statuses are compared below
                out_h, ops_h, cells_h = al.last_out.copy(), al.last_ops.copy(), al.last_cells.copy()
            al.ctx.set_option("band_on_host", 0)
            # both placements of the chaining kernel's tree (the global one with four pairs per wavefront — chain_rows_kernel,
            # 96 pairs of ragged match counts: rows that finish early, a last block with live rows only — and with one), both k-mer joins
            for chain_global, join_global, chain_rows in ((0, 0, 1), (1, 0, 1), (1, 0, 0), (1, 1, 1)):
                al.ctx.set_option("band_chain_global", chain_global)
                al.ctx.set_option("band_chain_rows", chain_rows)
                al.ctx.set_option("band_join_global", join_global)
                try:
                    out_d, ops_d = al.align_arrays(mode, x, xo, y, yo)
                except Exception as e:  # noqa: BLE001
                    out_d, ops_d = al.last_out, al.last_ops
                assert (al.last_cells == cells_h).all(), (mode, k, w, chain_global)
                for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops", "status"):
                    assert (out_d[f] == out_h[f]).all(), (mode, k, w, f, chain_global)
                for p in range(len(xs)):
                    if out_h["status"][p] == 0:
                        assert decode_ops(out_d[p], ops_d) == decode_ops(out_h[p], ops_h), (mode, k, w, p, chain_global)
            al.ctx.set_option("band_chain_global", -1)
            al.ctx.set_option("band_chain_rows", 1)
            al.ctx.set_option("band_join_global", 0)


def test_device_band_builder_large_batch_10kb():
    """The configuration bench.py runs (10 kb pairs, k-mer 16, w 32, sub-batches above the global-tree
    threshold): device-built bands vs host-built ones on 1536 pairs, every field and every operation."""
    from rust_bio_amd.banded import Aligner as BAligner
    P, L = 1536, 10_000
    x, off, y, _ = synth.sw_pairs(P, L, seed=91, sub=0.06, ins=0.02, dele=0.02)
    al = BAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), 16, 32)
    al.ctx.set_option("band_on_host", 1)
    out_h, ops_h = al.align_arrays(2, x, off, y, off)
    cells_h = al.last_cells.copy()
    al.ctx.set_option("band_on_host", 0)
    # k-mer join in LDS (what a 10 kb batch gets) and with its table in global memory; the chaining's event loop with four
    # pairs per wavefront (the default) and with one
    for join_global, chain_rows in ((0, 1), (1, 1), (0, 0)):
        al.ctx.set_option("band_join_global", join_global)
        al.ctx.set_option("band_chain_rows", chain_rows)
        out_d, ops_d = al.align_arrays(2, x, off, y, off)
        assert (al.last_cells == cells_h).all(), (join_global, chain_rows)
        for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops", "status", "ops_off"):
            assert (out_d[f] == out_h[f]).all(), (f, join_global, chain_rows)
        assert (ops_d[:int(out_d["n_ops"].sum())] == ops_h[:int(out_h["n_ops"].sum())]).all(), (join_global, chain_rows)
    al.ctx.set_option("band_join_global", 0)
    al.ctx.set_option("band_chain_rows", 1)


def test_fill_kernel_variants_agree():
    """K3v2 with scaled keys (default), K3v2 with explicit compare chains (force_wide) and K3 with one pair
    per wavefront (band_fill_v1) are three implementations of banded.rs:406-723: identical output."""
    from rust_bio_amd.banded import Aligner as BAligner
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    xs, ys = [], []
    for it in range(80):
        n = int(rng.integers(30, 2500))
        y = acgt[rng.integers(0, 4, size=n)]
        x = y.copy()
        nsub = max(1, n // 12)
        x[rng.integers(0, n, size=nsub)] = acgt[rng.integers(0, 4, size=nsub)]
        cut = int(rng.integers(0, max(1, n - 20)))
        x = np.delete(x, np.arange(cut, cut + int(rng.integers(0, 15))))
        if it % 4 == 0:
            x = x[int(rng.integers(0, 40)):]
        if it % 9 == 0:
            y = y[:max(10, n - int(rng.integers(0, 60)))]
        xs.append(x.tobytes())
        ys.append(y.tobytes())
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    for mode, sc in [(0, Scoring.from_scores(-5, -1, 1, -1).xclip(-9).yclip(-4)), (1, Scoring.from_scores(-3, -2, 2, -2)),
                     (2, Scoring.from_scores(-5, -1, 1, -1)), (3, Scoring.from_scores(-7, -1, 3, -4)),
                     (0, Scoring.from_scores(-5, -1, 1, -1).xclip_prefix_(0).yclip_suffix_(-1))]:
        al = BAligner.with_scoring(sc, 9, 7)
        res = []
        # band_fill_v1: -1 K3v2 whatever the batch size, 1 K3, 0 by sub-batch size (K3 for these 80 pairs)
        # band_interior_off: K3v2's general step in every strip (by default semiglobal-like scorings take a reduced step
        # in the strips that neither reach column n nor hold row m)
        for opts in ({"band_fill_v1": -1}, {"band_fill_v1": -1, "force_wide": 1}, {"band_fill_v1": 1}, {"band_fill_v1": 0},
                     {"band_fill_v1": -1, "band_interior_off": 1}, {"band_fill_v1": -1, "band_packed_off": 1},
                     {"band_fill_v1": -1, "band_p_block512": 1}):  # (round 5: K3p in 512-thread blocks compiled for 168 VGPRs)
            for k_, v_ in opts.items():
                al.ctx.set_option(k_, v_)
            try:
                out, ops = al.align_arrays(mode, x, xo, y, yo)
            except Exception:  # noqa: BLE001 - per-pair statuses are compared
                out, ops = al.last_out, al.last_ops
            res.append((out.copy(), ops.copy()))
            for k_ in opts:
                al.ctx.set_option(k_, -1 if k_ == "band_fill_v1" else 0)
        for out, ops in res[1:]:
            for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops", "status", "ops_off"):
                assert (out[f] == res[0][0][f]).all(), (mode, f)
            nb = int(out["n_ops"].sum())
            assert (ops[:nb] == res[0][1][:nb]).all(), mode


def test_device_resident_entry_equals_host_entry():
    import torch
    from rust_bio_amd.banded import Aligner as BAligner
    P, L = 600, 3000
    x, off, y, _ = synth.sw_pairs(P, L, seed=33, sub=0.06, ins=0.02, dele=0.02)
    al = BAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), 12, 20)
    out_h, ops_h = al.align_arrays(2, x, off, y, off)
    cells_h = al.last_cells.copy()  # Band::num_cells per pair, as the host-buffer call reported them
    dev = torch.device("cuda:0")
    dx, dy = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    doff = torch.from_numpy(off.astype(np.int64)).to(dev)
    stride = 2 * L + 8
    d_out = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
    d_ops = torch.zeros(P * stride, dtype=torch.uint8, device=dev)
    cells = al.align_dev(2, P, dx.data_ptr(), doff.data_ptr(), dy.data_ptr(), doff.data_ptr(), d_out.data_ptr(),
                         d_ops.data_ptr(), stride, want_cells=True)
    assert (cells == cells_h).all()
    rec = d_out.cpu().numpy().view(_lib.ALN_DTYPE)
    ops = d_ops.cpu().numpy().reshape(P, stride)
    for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops", "status"):
        assert (rec[f] == out_h[f]).all(), f
    for p in range(P):
        k_ = int(rec["n_ops"][p])
        assert int(rec["ops_off"][p]) == (p + 1) * stride - k_
        assert (ops[p, stride - k_:] == ops_h[int(out_h["ops_off"][p]):int(out_h["ops_off"][p]) + k_]).all()


@pytest.mark.parametrize("case", [
    dict(mode="semiglobal", kw=dict(gap_open=-5, gap_extend=-1, match=1, mismatch=-1), k=16, w=32),
    dict(mode="semiglobal", kw=dict(gap_open=-3, gap_extend=-2, match=2, mismatch=-4), k=12, w=9),
    dict(mode="semiglobal", kw=dict(gap_open=0, gap_extend=-1, match=1, mismatch=0), k=10, w=40),
    dict(mode="custom", kw=dict(gap_open=-4, gap_extend=-1, match=3, mismatch=-2, yclip_prefix=-3, yclip_suffix=-2), k=14, w=20),
    dict(mode="custom", kw=dict(gap_open=-6, gap_extend=0, match=1, mismatch=-3, yclip_prefix=0, yclip_suffix=MIN_SCORE), k=11, w=17),
], ids=lambda c: "%s-k%d-w%d" % (c["mode"], c["k"], c["w"]))
def test_interior_runs_long_reads_vs_oracle(case):
    """Scorings that keep x whole and clip y's prefix at a real price split every pair into K3v2 strips, an interior run
    (banded_fill2i.hip) and K3v2 strips again: 2-4 kb reads (60-120 strips of 32 rows), indels that move the band, a few
    unrelated and short pairs in the same wavefronts — whole alignments against the oracle."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(str(sorted(case['kw'].items())).encode()))
    xs, ys = [], []
    for p in range(20):
        n = int(rng.integers(1800, 4200))
        y = synth.random_dna(n, 7000 + p)
        if p % 7 == 6:
            x = synth.random_dna(int(rng.integers(300, 900)), 9000 + p)  # unrelated: the band is whatever the chain finds
        else:
            xm, lens = synth.mutate_fixed(y.reshape(1, -1), 8000 + p, 0.05, 0.03, 0.03)
            x = xm[0][:int(lens[0])]
            if p % 5 == 0:  # a long deletion and a long insertion: the band turns
                c = int(rng.integers(200, len(x) - 400))
                x = np.concatenate([x[:c], x[c + int(rng.integers(20, 90)):]])
                c = int(rng.integers(200, len(x) - 200))
                x = np.concatenate([x[:c], synth.random_dna(int(rng.integers(20, 90)), 9500 + p), x[c:]])
            if p % 4 == 1:
                x = x[int(rng.integers(1, 300)):len(x) - int(rng.integers(1, 300))]  # x inside y
        xs.append(np.asarray(x, dtype=np.uint8).tobytes())
        ys.append(np.asarray(y, dtype=np.uint8).tobytes())
    kw = dict(BASE)
    kw.update(case["kw"])
    differential(kw, True, case["mode"], case["k"], case["w"], xs, ys)
    packed = case["kw"]["gap_open"] <= -1  # (banded_api.hip: what K3p's unsigned keys need)
    redo_default = differential.last_ctx.band_redo_pairs()
    assert redo_default <= 2, redo_default  # related reads: no band cell anywhere near the floor of its strip
    # the interior runs again: on the int32 kernel alone, with every pair flagged and recomputed, and with a threshold in the
    # middle of the range (some pairs of a lane group flagged, others not)
    for opts, redo in (({"band_packed_off": 1}, "none"), ({"band_packed_thresh": 65535}, "all"), ({"band_packed_thresh": 63000}, "some")):
        differential(kw, True, case["mode"], case["k"], case["w"], xs, ys, opts)
        n = differential.last_ctx.band_redo_pairs()
        if not packed or redo == "none":
            assert n == 0, (opts, n)
        elif redo == "all":
            assert n >= 10, (opts, n)  # every pair that has an interior run
        elif case["kw"]["gap_extend"] < 0:  # (without an extension penalty no band cell sinks that far below its row's best)
            assert n >= 1, (opts, n)


@pytest.mark.parametrize("opts", [{}, {"band_tail_last": 1}, {"band_window": 1, "band_raster_late": 1},
                                  {"band_join_late": 1, "band_p_block512": 1, "band_budget_gb": 1}],
                         ids=["default", "tail-last", "round3-order", "round5-experiments"])
def test_several_sub_batches_and_the_remainder_first(opts):
    """A batch that spans several sub-batches (chunk_pairs = 16, 70 pairs: the remainder of 6 runs first, then four full
    ones) through both entry points: same alignments as the oracle whatever the order of the pipeline's stages, and the
    host-buffer flavour's compacted operations are where the records say."""
    xs, ys = synth.ragged_pairs(70, 700, seed=4711, min_len=150)
    al = Aligner.with_scoring(engine_scoring(dict(BASE, yclip_prefix=0, yclip_suffix=0), True), 9, 11)
    al.ctx.set_option("chunk_pairs", 16)
    for k_, v_ in opts.items():
        al.ctx.set_option(k_, v_)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    out, ops = al.align_arrays(MODES["custom"], x, xo, y, yo)
    osc = orc.make_scoring(**dict(BASE, yclip_prefix=0, yclip_suffix=0), match_scores_some=1)
    assert (out["status"] == 0).all()
    spans = []
    for p in range(len(xs)):
        want = orc.banded_align(osc, "custom", 9, 11, xs[p], ys[p])
        assert int(out["score"][p]) == want["score"] and decode_ops(out[p], ops) == want["ops"], p
        spans.append((int(out["ops_off"][p]), int(out["n_ops"][p])))
    spans.sort()
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))  # back to back, no overlap
    # the device-resident entry over the same sub-batches
    import torch
    dev = torch.device("cuda:0")
    dx, dy = torch.from_numpy(x.copy()).to(dev), torch.from_numpy(y.copy()).to(dev)
    dxo, dyo = torch.from_numpy(xo.astype(np.int64)).to(dev), torch.from_numpy(yo.astype(np.int64)).to(dev)
    stride = max(len(a) for a in xs) + max(len(b) for b in ys) + 8
    d_out = torch.zeros(len(xs) * 64, dtype=torch.uint8, device=dev)
    d_ops = torch.zeros(len(xs) * stride, dtype=torch.uint8, device=dev)
    al.align_dev(MODES["custom"], len(xs), dx.data_ptr(), dxo.data_ptr(), dy.data_ptr(), dyo.data_ptr(), d_out.data_ptr(),
                 d_ops.data_ptr(), stride)
    rec = d_out.view(torch.int32).view(len(xs), 16).cpu().numpy()
    for k_ in list(opts) + ["chunk_pairs"]:  # (the default ctx is shared between Aligners)
        al.ctx.set_option(k_, 0)
    assert (rec[:, 0] == out["score"]).all()


def test_many_small_sub_batches_equal_one_sub_batch():
    """The pipeline's streams (join two sub-batches ahead on its own stream and arrays, chaining / raster / row ranges, a fill's
    preparation under the previous fill, epilogue and traceback behind it) over 47 sub-batches of 32 pairs against the same
    1 500 pairs as one sub-batch: every record and every operation, twice (scratch reused), through the device-resident entry."""
    import torch
    from rust_bio_amd.banded import Aligner as BAligner
    P, L = 1500, 700
    x, off, y, _ = synth.sw_pairs(P, L, seed=123, sub=0.06, ins=0.02, dele=0.02)
    al = BAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), 10, 12)
    dev = torch.device("cuda:0")
    dx, dy = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    doff = torch.from_numpy(off.astype(np.int64)).to(dev)
    stride = 2 * L + 8
    outs = []
    for chunk in (0, 32, 32, 100, 0):
        al.ctx.set_option("chunk_pairs", chunk)
        d_out = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        d_ops = torch.zeros(P * stride, dtype=torch.uint8, device=dev)
        al.align_dev(2, P, dx.data_ptr(), doff.data_ptr(), dy.data_ptr(), doff.data_ptr(), d_out.data_ptr(), d_ops.data_ptr(), stride)
        torch.cuda.synchronize()
        outs.append((d_out.cpu().numpy().view(_lib.ALN_DTYPE).copy(), d_ops.cpu().numpy().copy()))
    al.ctx.set_option("chunk_pairs", 0)
    ref_rec, ref_ops = outs[0]
    assert (ref_rec["status"] == 0).all()
    for rec, ops in outs[1:]:
        for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops", "status", "ops_off"):
            assert (rec[f] == ref_rec[f]).all(), f
        for p in range(0, P, 7):
            k_ = int(rec["n_ops"][p])
            assert (ops[(p + 1) * stride - k_:(p + 1) * stride] == ref_ops[(p + 1) * stride - k_:(p + 1) * stride]).all(), p
    # and a handful against the oracle
    osc = orc.make_scoring(-5, -1, 1, -1)
    for p in range(0, P, 250):
        want = orc.banded_align(osc, "semiglobal", 10, 12, bytes(x[int(off[p]):int(off[p + 1])]), bytes(y[int(off[p]):int(off[p + 1])]))
        assert int(ref_rec["score"][p]) == want["score"]
