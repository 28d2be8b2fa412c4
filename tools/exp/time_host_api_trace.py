import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
torch.cuda.init()
from rust_bio_amd import _lib, synth
from rust_bio_amd.pairwise import Aligner, Scoring
n, L = 1_000_000, 150
x, xo, y, yo = synth.sw_pairs(n, L, seed=2)
al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
out, ops = al.align_arrays(3, x, xo, y, yo)
for chunk in (131072, 147456, 196608, 262144):
    al.ctx.set_option("host_chunk_pairs", chunk)
    al.align_arrays(3, x, xo, y, yo, out=out, ops=ops)
    print("---- chunk", chunk, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(2):
        al.align_arrays(3, x, xo, y, yo, out=out, ops=ops)
    dt = (time.perf_counter() - t0) / 2
    print("chunk %8d: %.2f ms  %.0f GCUPS" % (chunk, dt * 1e3, n * L * L / dt / 1e9), file=sys.stderr, flush=True)
