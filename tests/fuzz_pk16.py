"""Long differential fuzz of K1p (run by hand on a GPU box: python tests/fuzz_pk16.py SEED SECONDS): batches of
equal-length reads (the fast launch), with stragglers and unequal couples mixed in (the rest launch), over random
modes, clip patterns and MatchParams scorings inside and just outside the 12-bit bound, against the CPU oracle.
Round 2, final K1p (unpredicated steady-state steps) and K2 (common move sequence): seeds 7 and 11, 60 s each:
907 813 + 862 813 pairs, 0 mismatches.  Round 3 (packed-stream loads in the same kernel body): seeds 5 and 21,
1 234 724 pairs, 0 mismatches.
Round 3, final K1p (the LF flavour of the local kernel: no fold during the fill, the floor by saturation; multiply-add
cell packing in every flavour): seeds 31, 51, 61, 73: 1 023 922 + 1 181 880 + 603 645 + 1 292 857 pairs (+ the runs
before the last refactoring), 0 mismatches."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_py as orc
from rust_bio_amd import _lib
from rust_bio_amd.pairwise import MIN_SCORE, Aligner, Scoring, decode_ops

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0 = time.time()
n_pairs = n_fail = rounds = 0
CL = ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix")
while time.time() - t0 < budget and n_fail == 0:
    rounds += 1
    m = int(rng.choice([int(rng.integers(1, 385)), 150, 100, 96, 250, 36, 76]))
    n = int(rng.choice([m, m + int(rng.integers(0, 40)), max(1, m - int(rng.integers(0, 30))), int(rng.integers(1, 400))]))
    lim = max(1, 2040 // (max(m, n) + 2))  # magnitudes up to the bound, sometimes past it
    top = lim + (2 if rng.random() < 0.1 else 0)
    go, ge = -int(rng.integers(0, top + 1)), -int(rng.integers(0, min(top, 4) + 1))
    ma, mi = int(rng.integers(0, top + 1)), -int(rng.integers(0, top + 1))
    mode = int(rng.choice([0, 1, 2, 3, 3]))
    clips = {}
    if mode == 0:
        for c in CL:
            r = rng.random()
            clips[c] = MIN_SCORE if r < 0.35 else (0 if r < 0.6 else -int(rng.integers(1, top + 1)))
    nalpha = int(rng.integers(1, 6))
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)[:nalpha]
    P = int(rng.integers(1, 140))
    stragglers = rng.random() < 0.5
    xs, ys = [], []
    for p in range(P):
        mm, nn = m, n
        if stragglers and rng.random() < 0.08:
            mm = int(rng.integers(0, m + 1))
        if stragglers and rng.random() < 0.08:
            nn = int(rng.integers(0, n + 8))
        y = alpha[rng.integers(0, nalpha, size=nn)]
        if rng.random() < 0.7 and nn > 3 and mm > 0:
            x = np.resize(y, mm + 10)[int(rng.integers(0, 5)):][:mm + 4].copy()
            k = int(rng.integers(0, max(1, mm // 6)))
            x[rng.integers(0, len(x), size=k)] = alpha[rng.integers(0, nalpha, size=k)]
            if rng.random() < 0.6 and len(x) > 2:
                c = int(rng.integers(0, len(x) - 1))
                x = np.delete(x, np.arange(c, min(len(x), c + int(rng.integers(1, 7)))))
            x = np.resize(x, mm)
        else:
            x = alpha[rng.integers(0, nalpha, size=mm)]
        xs.append(x.astype(np.uint8).tobytes())
        ys.append(y.astype(np.uint8).tobytes())
    sc = Scoring.from_scores(go, ge, ma, mi)
    for c, v in clips.items():
        setattr(sc, c, v)
    kw = {c: clips.get(c, MIN_SCORE) for c in CL}
    osc = orc.make_scoring(go, ge, ma, mi, **kw)
    al = Aligner.with_scoring(sc)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    out, ops = al.align_arrays(mode, x, xo, y, yo)
    oout, oops, stride = orc.align_batch(osc, mode, x, xo, y, yo, threads=8)
    for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"):
        bad = np.nonzero(out[f].astype(np.int64) != oout[f].astype(np.int64))[0]
        if len(bad):
            n_fail += len(bad)
            p = int(bad[0])
            print("MISMATCH", f, "mode", mode, (go, ge, ma, mi), clips, "m,n", m, n, "pair", p, "of", P, xs[p], ys[p], out[f][p], oout[f][p])
            break
    else:
        for p in range(P):
            want = orc.decode_ops(oops[p * stride:p * stride + int(oout["n_ops"][p])])
            if decode_ops(out[p], ops) != want:
                n_fail += 1
                print("OPS MISMATCH mode", mode, (go, ge, ma, mi), clips, "m,n", m, n, "pair", p, xs[p], ys[p])
                break
    n_pairs += P
print("rounds", rounds, "pairs", n_pairs, "failures", n_fail)
