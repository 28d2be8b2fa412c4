"""time K1p fill for one library variant: python scratch/exp/time_sw.py <lib.so> [pairs]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import importlib
import torch
torch.cuda.init()
pkg = importlib.import_module("rust-bio_amd".replace("-", "_")) if False else None
from rust_bio_amd import _lib
_lib.SO_PATH = os.path.abspath(sys.argv[1])
from rust_bio_amd import synth_gpu
from rust_bio_amd.pairwise import Aligner, Scoring
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
L = 150
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
x, xo, y, yo = synth_gpu.sw_pairs_big(n, L, seed=2, device=dev)
d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
stride = 2 * L + 4
d_ops = torch.empty(n * stride, dtype=torch.uint8, device=dev)
al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), ctx=ctx)
stream = torch.cuda.current_stream().cuda_stream
def step():
    al.align_dev(3, n, x.data_ptr(), xo.data_ptr(), y.data_ptr(), yo.data_ptr(), L, L, d_out.data_ptr(), d_ops.data_ptr(), stride, stream)
for k, v in (a.split("=") for a in sys.argv[3:]):
    ctx.set_option(k, int(v))
step(); torch.cuda.synchronize()
ctx.enable_timing(True)
for _ in range(3): step()
torch.cuda.synchronize()
t = ctx.timing()
print("%-40s fill %.3f ms  traceback %.3f ms" % (os.path.basename(sys.argv[1]), t["fill_ms"] / t["fill_launches"], t["traceback_ms"] / t["traceback_launches"]))
