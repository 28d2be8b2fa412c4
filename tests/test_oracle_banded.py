"""Pins the banded ORACLE (oracle/banded.cpp, oracle/sparse.cpp) against the reference's
known-answer tests: sparse DP (sparse.rs:504-714), band geometry (banded.rs:1469-1618), the
banded == full differential inputs (banded.rs:1621-1753), the full-aligner KATs mirrored through
the banded aligner with k = w = 10 (banded.rs:1766-2181), the outside-band cases (2183-2415) and
the module doctest (29-90)."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as orc
from kat_util import check_expect, load, scoring_kwargs

SP = load("sparse_kats.json")
BK = load("banded_kats.json")
CMP = load("banded_compare.json")
PK = load("pairwise_kats.json")


def matches_of(c):
    return orc.find_kmer_matches(c["s1"].encode(), c["s2"].encode(), c["k"])


def test_find_kmer_matches():
    for c in SP["find_kmer_matches"]:
        assert len(matches_of(c)) == c["n_hits"]


@pytest.mark.parametrize("c", SP["lcskpp"], ids=lambda c: c["ref"])
def test_lcskpp(c):
    mm = matches_of(c)
    path, score = orc.lcskpp(mm, c["k"])
    assert score == c["score"]
    if "path" in c:
        assert path == c["path"]
    if "match_path" in c:
        assert [mm[i].tolist() for i in path] == c["match_path"]
    if c.get("diagonal"):
        assert [mm[i].tolist() for i in path] == [[i, i] for i in range(len(path))]


@pytest.mark.parametrize("c", SP["lcskpp_equals_sdpkpp_1_0_0"], ids=lambda c: c["ref"])
def test_lcskpp_equals_sdpkpp(c):
    mm = matches_of(c)
    assert orc.lcskpp(mm, c["k"]) == orc.sdpkpp(mm, c["k"], 1, 0, 0)


@pytest.mark.parametrize("c", SP["sdpkpp"], ids=lambda c: c["ref"])
def test_sdpkpp(c):
    mm = matches_of(c)
    path, score = orc.sdpkpp(mm, c["k"], c["match"], c["gap_open"], c["gap_extend"])
    assert score == c["score"]
    if "path" in c:
        assert path == c["path"]
    if c.get("diagonal"):
        assert [mm[i].tolist() for i in path] == [[i, i] for i in range(len(path))]


def band_apply(m, n, ops):
    arr = []
    for o in ops:
        if o[0] == "entry":
            arr.append([0, o[1], o[2], 0, o[3], 0])
        elif o[0] == "kmer":
            arr.append([1, o[1], o[2], o[3], o[4], 0])
    a = np.array(arr, dtype=np.uint32).reshape(-1, 6)
    st = np.zeros(n + 1, dtype=np.uint32)
    en = np.zeros(n + 1, dtype=np.uint32)
    L = orc.lib()
    L.orc_band_apply.restype = None
    L.orc_band_apply.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.orc_band_apply(m, n, a.ctypes.data, len(a), st.ctypes.data, en.ctypes.data)
    return [[int(s), int(e)] for s, e in zip(st, en)]


def test_band_add_entry_kat():
    for seq in BK["band_add_entry"]["sequences"]:
        assert band_apply(10, 10, seq["ops"]) == seq["ranges"]


def test_band_add_kmer_equals_entries():
    for c in BK["band_add_kmer"]["cases"]:
        r, col = c["start"]
        a = band_apply(c["m"], c["n"], [["kmer", r, col, c["k"], c["w"]]])
        b = band_apply(c["m"], c["n"], [["entry", r + i, col + i, c["w"]] for i in range(c["k"])])
        assert a == b


def banded_scoring(s):
    kw = scoring_kwargs(s)
    # closures (Scoring::new) carry match_scores = None, from_scores carries Some (banded.rs:1315-1318)
    kw["match_scores_some"] = 0 if (s.get("closure") or "matrix" in kw) else 1
    return orc.make_scoring(**kw)


@pytest.mark.parametrize("c", BK["cases"], ids=lambda c: c["name"])
def test_banded_kat(c):
    got = orc.banded_align(banded_scoring(c["scoring"]), c["mode"], c["k"], c["w"], c["x"].encode(), c["y"].encode())
    e = dict(c["expect"])
    if e.pop("yend_is_ylen", False):
        assert got["yend"] == got["ylen"]
    if "x_aln_len" in e:
        assert got["xend"] - got["xstart"] == e.pop("x_aln_len")
        assert got["yend"] - got["ystart"] == e.pop("y_aln_len")
    check_expect(got, e, c["name"])


@pytest.mark.parametrize("t", CMP["tests"], ids=lambda t: t["name"])
def test_banded_equals_full(t):
    # banded.rs:1422-1467: closure scoring (match_scores None), k = w = 10
    sc = orc.make_scoring(CMP["gap_open"], CMP["gap_extend"], CMP["match"], CMP["mismatch"], match_scores_some=0)
    for mode in t["modes"]:
        b = orc.banded_align(sc, mode, CMP["k"], CMP["w"], t["x"].encode(), t["y"].encode())
        f = orc.align(sc, mode, t["x"].encode(), t["y"].encode())
        b.pop("band_cells")
        assert b == f, (t["name"], mode)


@pytest.mark.parametrize("case", PK["cases"], ids=[c["name"] for c in PK["cases"]])
def test_full_aligner_kats_through_banded(case):
    # banded.rs:1766-2181 repeat the full aligner's KATs with k = 10, w = 10; the closures there
    # are Scoring::new (match_scores None) except test_scoring_from_scores / only_clips / zero_score
    s = dict(case["scoring"])
    from_scores = case["name"] in ("test_scoring_from_scores", "test_only_clips") or case["name"].startswith("test_zero_score")
    kw = scoring_kwargs(s)
    kw["match_scores_some"] = 1 if from_scores else 0
    got = orc.banded_align(orc.make_scoring(**kw), case["mode"], 10, 10, case["x"].encode(), case["y"].encode())
    check_expect(got, case["expect"], case["name"])


def test_max_cells_sentinel():
    # banded.rs:407-420: more than 5 M band cells -> empty alignment, score MIN_SCORE; no k-mer
    # match -> full matrix (banded.rs:1309-1313)
    rng = np.random.default_rng(1)
    x = bytes(rng.choice(list(b"AC"), size=2300).astype(np.uint8))
    y = bytes(rng.choice(list(b"GT"), size=2300).astype(np.uint8))
    sc = orc.make_scoring(-5, -1, 1, -1)
    got = orc.banded_align(sc, "semiglobal", 16, 8, x, y)
    assert got["band_cells"] == 2301 * 2301 > 5_000_000
    assert got["score"] == orc.MIN_SCORE and got["ops"] == [] and got["xlen"] == 0 and got["ylen"] == 0
    assert got["mode"] == 2  # the wrapper still stamps its mode


def test_band_monotone_on_random_inputs():
    # starts and ends of non-empty columns never decrease
    from rust_bio_amd import synth
    rng = np.random.default_rng(3)
    sc_modes = [(orc.MIN_SCORE, orc.MIN_SCORE, 0, 0), (0, 0, 0, 0), (orc.MIN_SCORE,) * 4, (-3, -4, 0, -2)]
    for trial in range(150):
        xs, ys = synth.ragged_pairs(1, 400, seed=1000 + trial, min_len=30)
        xp, xs_, yp, ys_ = sc_modes[trial % 4]
        sc = orc.make_scoring(-5, -1, 1, -1, xclip_prefix=xp, xclip_suffix=xs_, yclip_prefix=yp, yclip_suffix=ys_)
        k, w = int(rng.integers(4, 12)), int(rng.integers(2, 14))
        st, en, cells = orc.band_create(sc, k, w, xs[0], ys[0])
        ne = [(int(s), int(e)) for s, e in zip(st, en) if e > s]
        assert all(a[0] <= b[0] and a[1] <= b[1] for a, b in zip(ne, ne[1:])), (trial, k, w)


# ---- sparse.rs helpers behind custom_with_expanded_matches (banded.rs:338-389)
def test_expand_kmer_matches_kat():
    g = load("sparse_kats.json")["expand_kmer_matches"]
    for c in g["cases"]:
        x, y = c["x"].encode(), c["y"].encode()
        assert orc.find_kmer_matches(x, y, g["k"]).tolist() == c["matches"]
        got = orc.expand_kmer_matches(x, y, g["k"], c["matches"], g["allowed_mismatches"])
        assert got == [tuple(m) for m in c["expanded"]], c


def test_union_path_contains_sdpkpp_path():
    import numpy as np
    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for _ in range(30):
        y = acgt[rng.integers(0, 4, size=300)]
        x = y.copy()
        x[rng.integers(0, 300, size=20)] = acgt[rng.integers(0, 4, size=20)]
        mm = orc.find_kmer_matches(x.tobytes(), y.tobytes(), 6)
        if not len(mm):
            continue
        sdp = orc.sdpkpp(mm, 6, 1, -5, -1)[0]
        lcs = orc.lcskpp(mm, 6)[0]
        uni = orc.sdpkpp_union_lcskpp_path(mm, 6, 1, -5, -1)
        # sparse.rs:297-329: lcskpp prefix + whole sdpkpp path + lcskpp suffix
        i = uni.index(sdp[0])
        assert uni[i:i + len(sdp)] == sdp
        assert uni[:i] == lcs[:i] and set(uni[i + len(sdp):]) <= set(lcs)


def test_custom_with_matches_equals_custom():
    # banded.rs:313-321 with matches = find_kmer_matches(x, y, k) is `custom` (282-285)
    import numpy as np
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    sc = orc.make_scoring(-5, -1, 1, -1, xclip_prefix=-3, yclip_suffix=-2)
    for _ in range(20):
        y = acgt[rng.integers(0, 4, size=200)].tobytes()
        xa = np.frombuffer(y, dtype=np.uint8).copy()
        xa[rng.integers(0, 200, size=12)] = acgt[rng.integers(0, 4, size=12)]
        x = xa.tobytes()
        mm = orc.find_kmer_matches(x, y, 8)
        a = orc.banded_align(sc, "custom", 8, 6, x, y)
        b = orc.banded_align_with(sc, "custom", 8, 6, x, y, mm)
        assert (a["score"], a["ops"]) == (b["score"], b["ops"])
        # an empty match list is the full matrix (banded.rs:1309-1313; the fuzz target's reference run)
        full = orc.align(sc, "custom", x, y)
        c = orc.banded_align_with(sc, "custom", 8, 6, x, y, [])
        assert (c["score"], c["ops"]) == (full["score"], full["ops"])
