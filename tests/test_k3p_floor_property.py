"""The claim K3p (rust-bio_amd/csrc/banded_fill2p.hip) rests on, checked on the CPU with the kernel's arithmetic restated in
plain Python: an interior strip of the banded fill computed in UNSIGNED SATURATING 16-bit keys relative to the maximum of the
row above (0 = "minus infinity", values below the floor stick to it) gives, wherever the kernel does not flag the pair,

  * the same S of every band cell,
  * the same move of every band cell (bits 1-3 of the traceback byte),
  * the same "I opened" / "D opened" bits wherever the cell they open from is inside the band and above the threshold in
    this strip's frame — every cell of the strip itself is, a cell of the row above may have sunk below the floor when the
    base rose; elsewhere I / D are minus infinity in truth, or so far below the cell's S that neither S nor the I / D of the
    next cell can come from them: no traceback reaches those bits,
  * the same last row handed to the next strip,

as the int32 recurrence of K3i (keys score << 4 | priority << 1 | opened; banded.rs:556-680 without the clip machinery an
interior strip does not have) — and it flags exactly when a band cell's S is at or below fb + |gap open| * 16 + 32,
fb = 32 * match * 16 + 16 being the bound on everything that derives from the floor inside a strip.
Random strips: 32 rows, diagonal bands of random width and slope, boundary rows with values that reach below the floor,
scorings with match 0..3, y-prefix-clip candidates above and below the floor.  No GPU, no library: this pins the ARGUMENT;
tests/test_gpu_banded.py and tests/fuzz_banded.py (k3p mode) pin the kernel."""
import random

NEG = -(1 << 40)
C_MATCH, C_SUBST, C_INS, C_DEL, C_YP = 6, 5, 3, 2, 0
KI, KD = C_INS << 1, C_DEL << 1
ROWS = 32


def make_strip(rng):
    match = rng.choice([0, 1, 1, 2, 3])
    mismatch = -rng.randint(1, 5)
    go = -rng.randint(1, 8)
    ge = -rng.randint(0, 3) if rng.random() < 0.7 else -rng.randint(40, 300)  # (steep extensions: cells that do reach the floor)
    width = rng.randint(3, 40) if ge > -40 else rng.randint(25, 40)
    ncols = ROWS + width + rng.randint(0, 6)
    # band of row r (0-based inside the strip): columns cf[r] .. cl[r], non-decreasing
    cf, cl = [], []
    a = rng.randint(1, 4)
    for r in range(ROWS):
        if rng.random() < 0.8:
            a += 1
        lo = min(a, ncols)
        hi = min(ncols, lo + width + rng.randint(-2, 2))
        cf.append(lo)
        cl.append(max(lo, hi))
    for r in range(1, ROWS):  # monotone like Band::create's
        cf[r] = max(cf[r], cf[r - 1])
        cl[r] = max(cl[r], cl[r - 1])
    a0 = max(1, cf[0] - rng.randint(0, 2))
    above = (a0, max(a0, cl[0] - rng.randint(0, 2)))
    base_score = rng.randint(-3000, 30000)
    # how far below the row maximum the row above reaches (with steep extensions: far, so that no neighbour rescues a cell)
    depth = rng.choice([30, 300, 3000, 6000]) if ge > -40 else 6000
    S_above, I_above = {}, {}
    best_col = rng.randint(above[0], max(above[0], above[1]))
    for j in range(above[0], above[1] + 1):
        S_above[j] = base_score - (0 if j == best_col else rng.randint(0, depth) if ge > -40 else rng.randint(4500, 9000))
        I_above[j] = NEG if rng.random() < 0.2 else S_above[j] + go + ge * rng.randint(0, 4) - rng.randint(0, depth // 4)
    x = [rng.randint(0, 3) for _ in range(ROWS)]
    y = [rng.randint(0, 3) for _ in range(ncols + 2)]
    for r in range(ROWS):  # a diagonal of matches somewhere inside the band
        j = min(ncols, cf[r] + 1 + (r % 3 == 0))
        if rng.random() < 0.7:
            y[j] = x[r]
    ycl0 = min(S_above.values()) - rng.choice([0, 50, 5000]) if rng.random() < 0.5 or ge <= -40 else base_score - rng.randint(1, 200)
    ycl = [min(base_score - 1, ycl0) + ge * r for r in range(ROWS)]  # below the maximum of the row above, falling
    return dict(match=match, mismatch=mismatch, go=go, ge=ge, ncols=ncols, cf=cf, cl=cl, above=above, S_above=S_above,
                I_above=I_above, x=x, y=y, ycl=ycl, base=max(S_above.values()))


def exact(st):
    """K3i's step in unbounded integers; returns {(r, j): (S, byte)} over the band cells and the last row's (S, I) per column"""
    mk, mmk = (st["match"] << 4) | (C_MATCH << 1), (st["mismatch"] << 4) + (C_SUBST << 1)
    ge16, go_ti, go_td = st["ge"] << 4, (st["go"] << 4) + (KI | 1), (st["go"] << 4) + (KD | 1)
    NEGS = NEG << 4
    S_prev = {j: NEGS for j in range(0, st["ncols"] + 2)}
    I_prev = {j: NEGS | KI for j in range(0, st["ncols"] + 2)}
    for j, v in st["S_above"].items():
        S_prev[j] = v << 4
        I_prev[j] = (NEGS | KI) if st["I_above"][j] == NEG else ((st["I_above"][j] << 4) | KI)
    out = {}
    for r in range(ROWS):
        S_cur, I_cur = {j: NEGS for j in S_prev}, {j: NEGS | KI for j in S_prev}
        left_S, Dl = NEGS, NEGS | KD
        yk = st["ycl"][r] << 4
        for j in range(1, st["ncols"] + 1):
            inb = st["cf"][r] <= j <= st["cl"][r]
            m_key = S_prev[j - 1] + (mk if st["x"][r] == st["y"][j] else mmk)
            Iv = max(I_prev[j] + ge16, S_prev[j] + go_ti)
            Dv = max(Dl + ge16, left_S + go_td)
            kb = max(m_key, Iv, Dv, yk)
            if inb:
                S_cur[j] = kb & ~15
                I_cur[j] = Iv & ~1
                Dl = Dv & ~1
                out[(r, j)] = (S_cur[j] >> 4, (Iv & 1) | (kb & 0xE) | ((Dv & 1) << 4))
            else:
                Dl = NEGS | KD
            left_S = S_cur[j]
        S_prev, I_prev = S_cur, I_cur
    return out, {j: (S_prev[j], I_prev[j] & ~15) for j in range(1, st["ncols"] + 1)}


def subs(a, b):
    return a - b if a > b else 0


def packed(st):
    """the same strip in K3p's unsigned 16-bit keys; returns the cells, the last row in absolute keys, and the flag"""
    match_k = (st["match"] << 4) | (C_MATCH << 1)
    misc = ((-st["mismatch"]) << 4) - (C_SUBST << 1)
    DELTA, GE = match_k + misc, (-st["ge"]) << 4
    GOI, GOD = ((-st["go"]) << 4) - (KI | 1), ((-st["go"]) << 4) - (KD | 1)
    target = (0xFFF0 - DELTA - (st["match"] << 9) - 32) & ~15
    thresh = (st["match"] << 9) + 16 + ((-st["go"]) << 4) + 32
    shift = (st["base"] << 4) - target

    def rel(v):
        return min(max(v - shift, 0), 0xFFFF)
    S_prev = {j: 0 for j in range(0, st["ncols"] + 2)}
    I_prev = {j: KI for j in range(0, st["ncols"] + 2)}
    for j, v in st["S_above"].items():
        S_prev[j] = rel(v << 4)
        I_prev[j] = (0 if st["I_above"][j] == NEG else rel(st["I_above"][j] << 4)) | KI
    bound_rel = dict(S_prev)
    out, lo = {}, 0xFFFF
    for r in range(ROWS):
        S_cur, I_cur = {j: 0 for j in S_prev}, {j: 0 for j in S_prev}
        left_S, Dl = 0, 0
        yk = rel(st["ycl"][r] << 4)
        for j in range(1, st["ncols"] + 1):
            inb = st["cf"][r] <= j <= st["cl"][r]
            e = 1 if st["x"][r] == st["y"][j] else 0
            mad = (e * DELTA + S_prev[j - 1]) & 0xFFFF
            assert e * DELTA + S_prev[j - 1] <= 0xFFFF, "the base keeps the multiply-add inside 16 bits"
            m_key = subs(mad, misc)
            Iv = max(subs(I_prev[j], GE), subs(S_prev[j], GOI))
            Dv = max(subs(Dl, GE), subs(left_S, GOD))
            kb = max(m_key, Iv, Dv, yk)
            best = kb & ~15
            if inb:
                S_cur[j], I_cur[j], Dl = best, Iv & ~1, Dv & ~1
                lo = min(lo, best)
                out[(r, j)] = (best, (Iv & 1) | (kb & 0xE) | ((Dv & 1) << 4))
            else:
                Dl = 0
            left_S = S_cur[j]
        S_prev, I_prev = S_cur, I_cur
    last = {j: (S_prev[j] + shift, (I_prev[j] & ~15) + shift) for j in range(1, st["ncols"] + 1)}
    return out, last, lo <= thresh, shift, bound_rel, thresh


def test_unflagged_strips_are_exact_and_flags_fire_when_a_cell_sinks_to_the_floor():
    rng = random.Random(20260924)
    n_ok = n_flag = n_cells = 0
    for _ in range(1500):
        st = make_strip(rng)
        ex, ex_last = exact(st)
        pk, pk_last, flagged, shift, bound_rel, thresh = packed(st)
        assert set(ex) == set(pk)
        if flagged:
            n_flag += 1
            continue
        n_ok += 1
        for (r, j), (s_true, byte_true) in ex.items():
            s_rel, byte = pk[(r, j)]
            n_cells += 1
            assert s_rel + shift == s_true << 4, (r, j, st["match"], st["mismatch"], st["go"], st["ge"])
            assert (byte & 0xE) == (byte_true & 0xE), (r, j)
            above_ok = (st["cf"][r - 1] <= j <= st["cl"][r - 1]) if r else (st["above"][0] <= j <= st["above"][1] and bound_rel[j] > thresh)
            if above_ok:
                assert (byte & 1) == (byte_true & 1), ("I opened", r, j)
            if st["cf"][r] <= j - 1 <= st["cl"][r]:
                assert (byte & 16) == (byte_true & 16), ("D opened", r, j)
        for j in range(st["cf"][ROWS - 1], st["cl"][ROWS - 1] + 1):  # the row the next strip (or K3v2's phase 2) reads
            assert pk_last[j][0] == ex_last[j][0], ("last row S", j)
    # the sample holds both kinds, and enough cells to mean something
    assert n_ok >= 300 and n_flag >= 150 and n_cells >= 200_000, (n_ok, n_flag, n_cells)


def test_strips_chained_through_the_hand_over_row():
    """Several strips of one pair, each started from what the strip above handed on: the exact model from its own exact row, the
    16-bit model from ITS row — S plus shift, I plus shift with "minus infinity" arriving as the floor of the strip that wrote it —
    re-based on the exact maximum of that row.  A floor value resurrected by a falling base must not change a cell the kernel does
    not flag (both candidates of the next cell's I move by the same amount: the strip above checked its S against the threshold
    in the frame the I was written in)."""
    rng = random.Random(99)
    n_pairs_ok = n_strips_ok = 0
    for _ in range(250):
        st = make_strip(rng)
        if st["ge"] <= -40:
            continue
        ex_st, pk_st = dict(st), dict(st)
        ok = True
        for k in range(4):
            ex, ex_last = exact(ex_st)
            pk, pk_last, flagged, shift, bound_rel, thresh = packed(pk_st)
            if flagged:
                ok = False
                break
            for key, (s_true, byte_true) in ex.items():
                s_rel, byte = pk[key]
                assert s_rel + shift == s_true << 4 and (byte & 0xE) == (byte_true & 0xE), (k, key)
            n_strips_ok += 1
            # the next strip: the band goes on to the right, new symbols, the clip candidate keeps falling
            last_lo, last_hi = st["cf"][ROWS - 1], st["cl"][ROWS - 1]
            width = last_hi - last_lo
            cf, cl, a = [], [], last_lo
            for r in range(ROWS):
                if rng.random() < 0.8:
                    a += 1
                cf.append(a)
                cl.append(a + max(3, width + rng.randint(-1, 1)))
            for r in range(1, ROWS):
                cf[r], cl[r] = max(cf[r], cf[r - 1]), max(cl[r], cl[r - 1])
            ncols = cl[-1] + 2
            x = [rng.randint(0, 3) for _ in range(ROWS)]
            y = [rng.randint(0, 3) for _ in range(ncols + 2)]
            for r in range(ROWS):
                if rng.random() < rng.choice([0.2, 0.9]):  # stretches that lose score: the base falls
                    y[min(ncols, cf[r] + 1)] = x[r]
            ycl = [st["ycl"][-1] + st["ge"] * (ROWS * k + r + 1) for r in range(ROWS)]
            common = dict(st, cf=cf, cl=cl, ncols=ncols, x=x, y=y, ycl=ycl, above=(last_lo, last_hi))
            ex_st = dict(common, S_above={j: ex_last[j][0] >> 4 for j in range(last_lo, last_hi + 1)},
                         I_above={j: (NEG if ex_last[j][1] < -(1 << 38) else ex_last[j][1] >> 4) for j in range(last_lo, last_hi + 1)})
            ex_st["base"] = max(ex_st["S_above"].values())
            # the 16-bit model's own hand-over: absolute keys, a clean I, nothing marks a floor value as one
            pk_st = dict(common, S_above={j: pk_last[j][0] >> 4 for j in range(last_lo, last_hi + 1)},
                         I_above={j: pk_last[j][1] >> 4 for j in range(last_lo, last_hi + 1)})
            pk_st["base"] = max(pk_st["S_above"].values())
            assert pk_st["base"] == ex_st["base"]
            st = common
        n_pairs_ok += ok
    assert n_pairs_ok >= 60 and n_strips_ok >= 400, (n_pairs_ok, n_strips_ok)
