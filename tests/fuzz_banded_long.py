"""Differential fuzz of the banded engine's INTERIOR path (K3p packed 16-bit strips + detect-and-recompute through the int32
kernels) on LONG reads — run by hand on a GPU box:  python tests/fuzz_banded_long.py SEED SECONDS [MAXLEN=12000]
tests/fuzz_banded.py's k3p mode stops at 2 600 bp (its oracle is called pair by pair from Python); here reads go to 12 kb —
beyond configs[3]'s 10 kb: ~375 strips per pair, the re-basing chained through every one of them — and the oracle runs in
batches on all host threads.  Scorings: the ones that split a pair into K3v2 end strips and an interior run (semiglobal, and
custom with x kept whole); error rates from 0 to 25 %, long indels (the band follows them), partial overlaps; the redo
threshold drawn per round (0 = the real one, 65535 = every pair recomputed, in between = some)."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("BG_BAND_FILL_V1", "-1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_py as orc  # noqa: E402
from rust_bio_amd import _lib  # noqa: E402
from rust_bio_amd.banded import Aligner  # noqa: E402
from rust_bio_amd.pairwise import MIN_SCORE, Scoring  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
MAXLEN = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
threads = len(os.sched_getaffinity(0))
try:  # cgroup CPU quota: threads beyond it are only throttled (run A: 256 "cores" on a 16-CPU quota made the oracle 2.5 x slower)
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if quota != "max":
        threads = min(threads, max(1, -(-int(quota) // int(period))))
except (OSError, ValueError):
    pass
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
t0 = time.time()
n_pairs = n_fail = rounds = n_redo = 0
sum_len = 0
while time.time() - t0 < budget:
    rounds += 1
    k = int(rng.integers(8, 20))
    w = int(rng.integers(4, 48))
    go = -int(rng.integers(1, 9)); ge = -int(rng.integers(0, 4)); ma = int(rng.integers(1, 5)); mi = -int(rng.integers(1, 6))
    mode = 2 if rng.random() < 0.6 else 0
    clips = {}
    if mode == 0:
        clips = dict(xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE, yclip_prefix=-int(rng.integers(0, 12)),
                     yclip_suffix=MIN_SCORE if rng.random() < 0.3 else -int(rng.integers(0, 12)))
    sc = Scoring.from_scores(go, ge, ma, mi)
    for c, v in clips.items():
        sc = getattr(sc, c + "_")(v) if v != MIN_SCORE else sc
    kw = {c: clips.get(c, MIN_SCORE) for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix")}
    osc = orc.make_scoring(go, ge, ma, mi, **kw)
    P = int(rng.integers(32, 192))
    err = float(rng.choice([0.0, 0.02, 0.06, 0.10, 0.15, 0.25]))
    xs, ys = [], []
    for _ in range(P):
        n = int(rng.integers(2000, MAXLEN + 1))
        y = acgt[rng.integers(0, 4, size=n)]
        x = y.copy()
        ns = int(n * err * 0.6)
        if ns:
            x[rng.integers(0, n, size=ns)] = acgt[rng.integers(0, 4, size=ns)]
        for _ in range(int(n * err * 0.04) + int(rng.integers(0, 4))):
            c = int(rng.integers(0, max(1, len(x) - 1)))
            ln = int(rng.integers(1, 40)) if rng.random() < 0.9 else int(rng.integers(40, 400))
            if rng.random() < 0.5:
                x = np.delete(x, np.arange(c, min(len(x), c + ln)))
            else:
                x = np.insert(x, c, acgt[rng.integers(0, 4, size=ln)])
        r = rng.random()
        if r < 0.15:
            x = x[int(rng.integers(0, max(1, len(x) // 3))):]
        elif r < 0.3:
            x = x[:max(1, len(x) - int(rng.integers(0, max(1, len(x) // 3))))]
        if len(x) == 0:
            x = y[:1].copy()
        xs.append(x.tobytes()); ys.append(y.tobytes())
        sum_len += len(x) + len(y)
    al = Aligner.with_scoring(sc, k, w)
    x, xo = _lib.concat(xs); y, yo = _lib.concat(ys)
    thr = [0, 0, 0, 65535, int(rng.integers(58000, 65300))][int(rng.integers(0, 5))]
    al.ctx.set_option("band_packed_thresh", thr)
    try:
        out, ops = al.align_arrays(mode, x, xo, y, yo)
    except Exception:
        out, ops = al.last_out, al.last_ops
    n_redo += al.ctx.band_redo_pairs()
    al.ctx.set_option("band_packed_thresh", 0)
    mname = ["custom", "global", "semiglobal", "local"][mode]
    oout, oops, ostride, ocells = orc.banded_align_batch(osc, mname, k, w, x, xo, y, yo, threads=threads)
    kind = (oops.reshape(P, ostride) & 0xFF).astype(np.uint8)
    for p in range(P):
        n_pairs += 1
        ok = out["status"][p] == 0 and all(int(out[f][p]) == int(oout[f][p]) for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"))
        ok = ok and int(al.last_cells[p]) == int(ocells[p])
        if ok:
            kq, oq = int(out["n_ops"][p]), int(out["ops_off"][p])
            ok = bool((ops[oq:oq + kq] == kind[p, :kq]).all())
        if not ok:
            n_fail += 1
            if n_fail < 6:
                print("MISMATCH", dict(k=k, w=w, mode=mode, go=go, ge=ge, ma=ma, mi=mi, clips=clips, thr=thr, err=err, pair=p, xlen=len(xs[p]),
                                       ylen=len(ys[p]), status=int(out["status"][p]), score=(int(out["score"][p]), int(oout["score"][p]))))
                np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_long_bad_{n_fail}.npy"),
                        np.array([xs[p], ys[p], repr((k, w, mode, go, ge, ma, mi, clips, thr))], dtype=object), allow_pickle=True)
print(f"fuzz_banded_long: rounds {rounds} pairs {n_pairs} mean_len {sum_len / max(1, 2 * n_pairs):.0f} maxlen {MAXLEN} failures {n_fail} "
      f"pairs recomputed behind K3p {n_redo}")
sys.exit(1 if n_fail else 0)
