"""K1p vs K1 on ragged batches: python scratch/exp/time_ragged.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
torch.cuda.init()
from rust_bio_amd import _lib
from rust_bio_amd.pairwise import Aligner, Scoring
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
n = 1_000_000
rng = np.random.default_rng(1)
for name, lens_x, lens_y in [("uniform 150", np.full(n, 150), np.full(n, 150)),
                              ("x in [100,150], y 150", rng.integers(100, 151, n), np.full(n, 150)),
                              ("x,y in [100,150]", rng.integers(100, 151, n), rng.integers(100, 151, n)),
                              ("1% stragglers", np.where(rng.random(n) < 0.01, 140, 150), np.full(n, 150)),
                              ("x in [150,192]", rng.integers(150, 193, n), np.full(n, 192))]:
    xo = np.zeros(n + 1, dtype=np.int64); xo[1:] = np.cumsum(lens_x)
    yo = np.zeros(n + 1, dtype=np.int64); yo[1:] = np.cumsum(lens_y)
    x = torch.randint(0, 4, (int(xo[-1]),), dtype=torch.uint8, device=dev)
    y = torch.randint(0, 4, (int(yo[-1]),), dtype=torch.uint8, device=dev)
    dxo, dyo = torch.from_numpy(xo).to(dev), torch.from_numpy(yo).to(dev)
    mx, my = int(lens_x.max()), int(lens_y.max())
    stride = mx + my + 4
    d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_ops = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)
    res = []
    for mode in (3, 2):
        for opt in (0, 2, 1):
            ctx.set_option("no_pk16", 1 if opt == 1 else 0)
            ctx.set_option("no_couples", 1 if opt == 2 else 0)
            def step():
                al.align_dev(mode, n, x.data_ptr(), dxo.data_ptr(), y.data_ptr(), dyo.data_ptr(), mx, my, d_out.data_ptr(), d_ops.data_ptr(), stride,
                             torch.cuda.current_stream().cuda_stream)
            step(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); step(); step(); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 2)
    ctx.set_option("no_pk16", 0)
    ctx.set_option("no_couples", 0)
    print("%-24s local: K1p+couples %.2f  K1p %.2f  K1 %.2f ms | semiglobal: %.2f  %.2f  %.2f ms" % (name, res[0], res[1], res[2], res[3], res[4], res[5]))
