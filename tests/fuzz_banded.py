"""Long differential fuzz (run by hand on a GPU box: python tests/fuzz_banded.py SEED SECONDS): banded engine
(device band builder + K3v2/K3 + K4) vs the CPU oracle over random k, w, modes, clips and sequences.
Round 1: 126 473 pairs in 1817 configurations, 0 mismatches; round 2 (K3v2 chunks, 8-row traceback lines, LDS k-mer
join, compacted raster): 181 875 pairs in 2624 configurations (seed 20260924, 150 s), 0 mismatches; round 3 (Sn / Ly per
block of 16 steps, device compaction, K3 for small sub-batches): seeds 77, 5 and 20260925, 226 909 pairs in 3272 configurations, 0 mismatches.
Round 3, final host path (operations downloaded per sub-batch): seed 71, 120 s: 144 386 pairs, 0 mismatches."""
import sys, time
import numpy as np
import os
os.environ.setdefault("BG_BAND_FILL_V1", "-1")  # small fuzz batches still run K3v2 (see tests/conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_py as orc
from rust_bio_amd import _lib
from rust_bio_amd.banded import Aligner
from rust_bio_amd.pairwise import Scoring, decode_ops, MIN_SCORE

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
# third argument "k3p": only scorings that split pairs into K3v2 strips and an interior run (x kept whole, y's prefix clipped at
# a real price), longer reads, and K3p's redo threshold drawn per round (0 = the real one, 65535 = everything recomputed by
# the int32 kernels, in between = some pairs of a lane group flagged and others not)
K3P = len(sys.argv) > 3 and sys.argv[3] == "k3p"
n_redo = 0
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
t0 = time.time(); n_pairs = 0; n_fail = 0; rounds = 0
while time.time() - t0 < budget:
    rounds += 1
    k = int(rng.integers(3, 14)); w = int(rng.integers(1, 25))
    go = -int(rng.integers(0, 9)); ge = -int(rng.integers(0, 4)); ma = int(rng.integers(0, 5)); mi = -int(rng.integers(0, 6))
    mode = int(rng.integers(0, 4))
    clips = {}
    if mode == 0:
        for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
            r = rng.random()
            clips[c] = MIN_SCORE if r < 0.4 else (0 if r < 0.6 else -int(rng.integers(1, 30)))
    if K3P:
        go = -int(rng.integers(1, 9)); mi = -int(rng.integers(1, 6))
        mode = 2 if rng.random() < 0.5 else 0
        if mode == 0:
            clips = dict(xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE, yclip_prefix=-int(rng.integers(0, 12)),
                         yclip_suffix=MIN_SCORE if rng.random() < 0.3 else -int(rng.integers(0, 12)))
    closure = rng.random() < 0.3 and not K3P
    sc = Scoring.from_scores(go, ge, ma, mi) if not closure else Scoring.new(go, ge, (lambda a, b, ma=ma, mi=mi: ma if a == b else mi))
    for c, v in clips.items():
        sc = getattr(sc, c + "_")(v) if v != MIN_SCORE else sc
    kw = dict(xclip_prefix=clips.get("xclip_prefix", MIN_SCORE), xclip_suffix=clips.get("xclip_suffix", MIN_SCORE),
              yclip_prefix=clips.get("yclip_prefix", MIN_SCORE), yclip_suffix=clips.get("yclip_suffix", MIN_SCORE))
    if closure:
        mat = np.full((256, 256), mi, dtype=np.int32); np.fill_diagonal(mat, ma)
        osc = orc.make_scoring(go, ge, matrix=mat, **kw)
    else:
        osc = orc.make_scoring(go, ge, ma, mi, **kw)
    xs, ys = [], []
    P = int(rng.integers(20, 120))
    for _ in range(P):
        n = int(rng.integers(1, 2600 if K3P else 900))
        alpha = acgt[:int(rng.integers(2, 5))]
        y = alpha[rng.integers(0, len(alpha), size=n)]
        x = y.copy()
        if rng.random() < 0.8 and n > 4:
            ns = int(rng.integers(0, max(1, n // 6)))
            x[rng.integers(0, n, size=ns)] = alpha[rng.integers(0, len(alpha), size=ns)]
            for _ in range(int(rng.integers(0, 4))):
                c = int(rng.integers(0, max(1, len(x) - 1)))
                if rng.random() < 0.5:
                    x = np.delete(x, np.arange(c, min(len(x), c + int(rng.integers(1, 12)))))
                else:
                    x = np.insert(x, c, alpha[rng.integers(0, len(alpha), size=int(rng.integers(1, 12)))])
            if rng.random() < 0.3:
                x = x[int(rng.integers(0, max(1, len(x) // 3))):]
            if rng.random() < 0.3:
                x = x[:max(0, len(x) - int(rng.integers(0, max(1, len(x) // 3))))]
        else:
            x = alpha[rng.integers(0, len(alpha), size=int(rng.integers(0, 300)))]
        xs.append(x.tobytes()); ys.append(y.tobytes())
    al = Aligner.with_scoring(sc, k, w)
    x, xo = _lib.concat(xs); y, yo = _lib.concat(ys)
    thr = 0
    if K3P:
        thr = [0, 0, 65535, int(rng.integers(58000, 65300))][int(rng.integers(0, 4))]
        al.ctx.set_option("band_packed_thresh", thr)
        al.ctx.set_option("band_packed_off", 1 if os.environ.get("FUZZ_PACKED_OFF") else 0)
    try:
        out, ops = al.align_arrays(mode, x, xo, y, yo)
    except Exception:
        out, ops = al.last_out, al.last_ops
    if K3P:
        n_redo += al.ctx.band_redo_pairs()
        al.ctx.set_option("band_packed_thresh", 0)
    for p in range(P):
        n_pairs += 1
        try:
            ref = orc.banded_align(osc, ["custom", "global", "semiglobal", "local"][mode], k, w, xs[p], ys[p])
        except Exception:
            if out["status"][p] == 0:
                n_fail += 1; print("engine ok where oracle failed", k, w, mode, xs[p], ys[p])
            continue
        if out["status"][p] != 0:
            n_fail += 1
            if n_fail < 6:
                print("status", out["status"][p], k, w, mode, go, ge, ma, mi, clips, "thr", thr, len(xs[p]), len(ys[p]))
            continue
        got = (int(out["score"][p]), int(out["xstart"][p]), int(out["xend"][p]), int(out["ystart"][p]), int(out["yend"][p]), decode_ops(out[p], ops))
        want = (ref["score"], ref["xstart"], ref["xend"], ref["ystart"], ref["yend"], ref["ops"])
        if got != want:
            n_fail += 1
            if n_fail < 6:
                print("MISMATCH", k, w, mode, go, ge, ma, mi, clips, closure, thr, xs[p], ys[p], got[:5], want[:5])
print(f"rounds {rounds} pairs {n_pairs} failures {n_fail}" + (f" pairs recomputed behind K3p {n_redo}" if K3P else ""))
sys.exit(1 if n_fail else 0)
