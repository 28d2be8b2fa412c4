"""time the banded kernels for one library variant: python scratch/exp/time_banded.py <lib.so> [pairs]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import torch
torch.cuda.init()
from rust_bio_amd import _lib
_lib.SO_PATH = os.path.abspath(sys.argv[1])
from rust_bio_amd import synth_gpu
from rust_bio_amd.pairwise import Scoring
from rust_bio_amd.banded import Aligner as BandedAligner
Pb = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
OPTS = [a.split("=") for a in sys.argv[3:]]  # ctx options, name=value
Lb, kb, wb = 10_000, 16, 32
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
for k_, v_ in OPTS:
    ctx.set_option(k_, int(v_))
bx, bxo, by, byo = synth_gpu.sw_pairs_big(Pb, Lb, seed=4, device=dev, sub=0.06, ins=0.02, dele=0.02, chunk=64)
bal = BandedAligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), kb, wb, ctx=ctx)
d_boff = torch.arange(Pb + 1, dtype=torch.int64, device=dev) * Lb
bstride = 2 * Lb + 8
d_bout = torch.empty(Pb * 64, dtype=torch.uint8, device=dev)
d_bops = torch.empty(Pb * bstride, dtype=torch.uint8, device=dev)
def step():
    bal.align_dev(2, Pb, bx.data_ptr(), d_boff.data_ptr(), by.data_ptr(), d_boff.data_ptr(), d_bout.data_ptr(), d_bops.data_ptr(), bstride)
step(); torch.cuda.synchronize()
import time
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
ctx.enable_timing(True)
step(); torch.cuda.synchronize()
t = ctx.timing()
print("%-28s wall %.1f ms  (%.0f pairs/s)  fill %.2f ms  traceback %.2f ms  %s" % (os.path.basename(sys.argv[1]), wall * 1e3, Pb / wall, t["fill_ms"], t["traceback_ms"], {k: round(v, 2) for k, v in t.items() if k not in ("fill_ms", "traceback_ms")}))
