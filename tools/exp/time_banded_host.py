"""bg_align_banded_batch through pageable host buffers at BASELINE configs[3]'s size (BG_TRACE=1 prints the host laps)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from rust_bio_amd import _lib, synth_gpu
from rust_bio_amd.pairwise import Scoring
from rust_bio_amd.banded import Aligner as BandedAligner
Pb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
Lb, kb, wb = 10_000, 16, 32
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
bx, bxo, by, byo = synth_gpu.sw_pairs_big(Pb, Lb, seed=4, device=dev, sub=0.06, ins=0.02, dele=0.02, chunk=64)
hx, hy = bx.cpu().numpy(), by.cpu().numpy()
hoff = np.arange(Pb + 1, dtype=np.uint64) * np.uint64(Lb)
del bx, by
bal = BandedAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), kb, wb, ctx=ctx)
o, q = bal.align_arrays(2, hx, hoff, hy, hoff)
for rep in range(2):
    print("---- call", rep, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    bal.align_arrays(2, hx, hoff, hy, hoff, out=o, ops=q)
    dt = time.perf_counter() - t0
    print("%.1f ms  %.0f pairs/s" % (dt * 1e3, Pb / dt), flush=True)
