"""bg_fm_backward_search_batch through pageable host buffers (BG_TRACE_HOST=1 prints where the host side spends its time)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from rust_bio_amd import _lib, synth_gpu
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.suffix_array import bwt_dev, suffix_array_dev
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
n_g, n_q, P = 100_000_000, 10_000_000, 100
g = synth_gpu.genome(n_g, seed=3, device=dev)
sa = suffix_array_dev(g, ctx=ctx)
b = bwt_dev(g, sa, ctx=ctx)
fm = FMIndex.from_device(b, 128, b"ACGTNacgtn", ctx=ctx)
pat, off = synth_gpu.fm_patterns(g, n_q, P, seed=4)
hp = pat.cpu().numpy()
ho = np.arange(n_q + 1, dtype=np.uint64) * np.uint64(P)
fm.backward_search_arrays(hp[:P * 1000], ho[:1001])
out = None
for rep in range(5):
    t0 = time.perf_counter()
    out = fm.backward_search_arrays(hp, ho, out=out if rep >= 2 else None)  # the first two calls page fresh result arrays in
    dt = time.perf_counter() - t0
    print("%.2f ms  %.1f M queries/s" % (dt * 1e3, n_q / dt / 1e6), flush=True)
