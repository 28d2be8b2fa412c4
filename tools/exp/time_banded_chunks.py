"""Device-resident banded call of <pairs> 10 kb pairs under several chunk_pairs settings: python tools/exp/time_banded_chunks.py <pairs> <chunk> [<chunk> ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import torch
torch.cuda.init()
from rust_bio_amd import _lib
from rust_bio_amd import synth_gpu
from rust_bio_amd.pairwise import Scoring
from rust_bio_amd.banded import Aligner as BandedAligner
Pb = int(sys.argv[1])
chunks = [int(v) for v in sys.argv[2:]] or [0]
Lb, kb, wb = 10_000, 16, 32
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
bx, bxo, by, byo = synth_gpu.sw_pairs_big(Pb, Lb, seed=4, device=dev, sub=0.06, ins=0.02, dele=0.02, chunk=64)
bal = BandedAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), kb, wb, ctx=ctx)
d_boff = torch.arange(Pb + 1, dtype=torch.int64, device=dev) * Lb
bstride = 2 * Lb + 8
d_bout = torch.empty(Pb * 64, dtype=torch.uint8, device=dev)
d_bops = torch.empty(Pb * bstride, dtype=torch.uint8, device=dev)
def step():
    return bal.align_dev(2, Pb, bx.data_ptr(), d_boff.data_ptr(), by.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(), d_bops.data_ptr(), bstride, want_cells=True)
ref = None
for c in chunks:
    ctx.set_option("chunk_pairs", c)
    step(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); cells = step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    cs = float(np.asarray(cells).sum())
    rec = d_bout.clone()
    same = True if ref is None else bool((rec == ref).all().item())
    if ref is None: ref = rec
    print("chunk_pairs %6d  wall %.1f ms (%.0f pairs/s, %.1f G band cells/s)  records equal to the first run: %s" % (c, best * 1e3, Pb / best, cs / best / 1e9, same), flush=True)
