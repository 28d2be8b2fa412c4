"""A/B of a banded ctx option on the device-resident path: python tools/exp/time_banded_ab.py <pairs> <option> [<option> ...]
An option may be written name=value:restore (e.g. band_chain_rows=0:1 for an option that defaults to 1).
Every option is timed at 0 and at 1 (wall of a call, event-timed fill and traceback) and the records of the two runs are compared."""
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
Pb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
opts = sys.argv[2:] or ["band_interior_off"]
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
def measure(tag):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); cells = step(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    ctx.enable_timing(True)
    step(); torch.cuda.synchronize()
    t = ctx.timing()
    ctx.enable_timing(False)
    c = float(np.asarray(cells).sum())
    print("%-28s wall %.1f ms (%.0f pairs/s, %.1f G band cells/s)  fill %.2f ms  traceback %.2f ms  dp-only %.1f G" % (
        tag, wall * 1e3, Pb / wall, c / wall / 1e9, t["fill_ms"], t["traceback_ms"], c / (t["fill_ms"] + t["traceback_ms"]) / 1e6), flush=True)
    return d_bout.clone(), d_bops.clone()
base = measure("default")
print("   redo pairs:", ctx.band_redo_pairs(), flush=True)
for o in opts:
    name, val, back = o, 1, 0
    if "=" in o:
        name, rest = o.split("=")
        val, back = (int(v) for v in rest.split(":"))
    ctx.set_option(name, val)
    got = measure("%s=%d" % (name, val))
    print("   redo pairs:", ctx.band_redo_pairs(), flush=True)
    ctx.set_option(name, back)
    rec_a, rec_b = base[0].view(torch.int32).view(Pb, 16), got[0].view(torch.int32).view(Pb, 16)
    print("   records equal:", bool((rec_a == rec_b).all().item()), " ops equal:", bool((base[1] == got[1]).all().item()), flush=True)
