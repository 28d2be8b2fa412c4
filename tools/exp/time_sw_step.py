"""Wall time of the headline step (bg_align_batch_dev, 1 M x 150 bp Aligner::local, device-resident) under ctx options, with the
records and operations of every variant compared with the first one's:
    python tools/exp/time_sw_step.py [pairs] [option=value[,option=value...]] ...      e.g.  no_tb_overlap=1  chunk_pairs=131072"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from rust_bio_amd import _lib, synth_gpu  # noqa: E402
from rust_bio_amd.pairwise import Aligner, Scoring  # noqa: E402

args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 1_000_000
L = 150
dev = torch.device("cuda:0")
x, xo, y, yo = synth_gpu.sw_pairs_big(n, L, seed=2, device=dev)
stride = 2 * L + 4
stream = torch.cuda.current_stream().cuda_stream
ref = None
for variant in [""] + args:
    ctx = _lib.Context(0)
    for kv in filter(None, variant.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    d_ops = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)

    def step():
        al.align_dev(3, n, x.data_ptr(), xo.data_ptr(), y.data_ptr(), yo.data_ptr(), L, L, d_out.data_ptr(), d_ops.data_ptr(), stride, stream)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    rec = d_out.view(torch.int32).view(n, 16)
    same = None
    if ref is None:
        ref = (rec[:, :8].clone(), d_ops.clone())
    else:  # records (score, coordinates, n_ops) and every operation byte of the right-aligned slots
        k = rec[:, 7].to(torch.int64)
        idx = torch.arange(stride, device=dev).view(1, -1)
        live = idx >= (stride - k).view(-1, 1)
        same = bool(torch.equal(ref[0], rec[:, :8]) and torch.equal(ref[1].view(n, stride)[live], d_ops.view(n, stride)[live]))
    print("%-44s %.3f ms per step  %.1f GCUPS%s" % (variant or "(default)", dt * 1e3, n * L * L / dt / 1e9, "" if same is None else "  same results: %s" % same), flush=True)
    del al, ctx
