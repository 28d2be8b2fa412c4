"""fill / traceback / wall time of the banded pipeline at small batch sizes, K3v2 (eight pairs per wavefront) against K3
(one pair per wavefront; ctx option band_fill_v1): python tools/exp/time_banded_small.py [sizes...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from rust_bio_amd import _lib, synth_gpu
from rust_bio_amd.pairwise import Scoring
from rust_bio_amd.banded import Aligner as BandedAligner
sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 1696, 4096, 8192, 12500]
Lb, kb, wb = 10_000, 16, 32
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
Pmax = max(sizes)
bx, bxo, by, byo = synth_gpu.sw_pairs_big(Pmax, Lb, seed=4, device=dev, sub=0.06, ins=0.02, dele=0.02, chunk=64)
bal = BandedAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), kb, wb, ctx=ctx)
d_boff = torch.arange(Pmax + 1, dtype=torch.int64, device=dev) * Lb
bstride = 2 * Lb + 8
d_bout = torch.empty(Pmax * 64, dtype=torch.uint8, device=dev)
d_bops = torch.empty(Pmax * bstride, dtype=torch.uint8, device=dev)
for Pb in sizes:
    for opt in (sys.argv[0] and [("v2", 0), ("v1", 1)]):
        ctx.set_option("band_fill_v1", opt[1])
        def step():
            bal.align_dev(2, Pb, bx.data_ptr(), d_boff.data_ptr(), by.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(), d_bops.data_ptr(), bstride)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
        ctx.enable_timing(True); step(); torch.cuda.synchronize(); t = ctx.timing(); ctx.enable_timing(False)
        print("%6d pairs %s  wall %7.2f ms (%8.0f pairs/s)  fill %7.2f ms  traceback %6.2f ms" % (Pb, opt[0], wall * 1e3, Pb / wall, t["fill_ms"], t["traceback_ms"]), flush=True)
