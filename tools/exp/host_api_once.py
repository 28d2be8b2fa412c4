"""One warm bg_align_batch call on 1M x 150 bp pairs (for rocprofv3 --kernel-trace --memory-copy-trace timelines)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
torch.cuda.init()
from rust_bio_amd import _lib, synth
from rust_bio_amd.pairwise import Aligner, Scoring
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 150
x, xo, y, yo = synth.sw_pairs(n, L, seed=2)
al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
for k, v in (a.split("=") for a in sys.argv[2:]):
    al.ctx.set_option(k, int(v))
out, ops = al.align_arrays(3, x, xo, y, yo)
al.align_arrays(3, x, xo, y, yo, out=out, ops=ops)
for _ in range(3):
    t0 = time.perf_counter()
    al.align_arrays(3, x, xo, y, yo, out=out, ops=ops)
    dt = time.perf_counter() - t0
    print("%.2f ms  %.0f GCUPS" % (dt * 1e3, n * L * L / dt / 1e9), file=sys.stderr, flush=True)
