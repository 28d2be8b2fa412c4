"""K7 (FMDIndex::all_smems) on T$R$ of a synthetic genome: reads/s of bg_fmd_smems_batch64_dev.
    [BG_SO=<variant .so>] python tools/exp/time_fmd.py [genome_bp=50000000] [reads=500000]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rust_bio_amd import _lib  # noqa: E402

if os.environ.get("BG_SO"):
    _lib.SO_PATH = os.path.abspath(os.environ["BG_SO"])
from rust_bio_amd import synth_gpu  # noqa: E402
from rust_bio_amd.fmindex import FMIndex  # noqa: E402
from rust_bio_amd.suffix_array import bwt_dev, suffix_array_dev  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
L, MINLEN, CAP = 150, 20, 24
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
N = 2 * G + 2
comp = torch.zeros(256, dtype=torch.uint8, device=dev)
for a_, b_ in zip(b"ACGTN", b"TGCAN"):
    comp[a_] = b_
g = torch.empty(N, dtype=torch.uint8, device=dev)
g[:G] = synth_gpu.genome(G, seed=51, device=dev)[:G]
g[G] = ord("$")
g[G + 1:N - 1] = comp[g[:G].flip(0).to(torch.int64)]
g[N - 1] = ord("$")
d_sa = suffix_array_dev(g, ctx=ctx)
d_b = bwt_dev(g, d_sa, ctx=ctx)
del d_sa
fm = FMIndex.from_device(d_b, 128, b"ACGTNacgtn", ctx=ctx)
z = synth_gpu.splitmix64(4321, NR, dev)
start = torch.remainder(z & ((1 << 62) - 1), G - L)
ar = torch.arange(L, dtype=torch.int64, device=dev)
reads = g[start.view(-1, 1) + ar.view(1, -1)].clone()
u = synth_gpu.splitmix64(99, NR * L, dev).view(NR, L)
acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
reads = torch.where((u & 0xFFFF) < int(0.03 * 65536), acgt[(u >> 20) & 3], reads)
rc = torch.arange(NR, device=dev) % 2 == 1
reads[rc] = comp[reads[rc].flip(1).to(torch.int64)]
reads = reads.contiguous()
r_off = torch.arange(NR + 1, dtype=torch.int64, device=dev) * L
d_cnt = torch.zeros(NR, dtype=torch.int32, device=dev)
d_out = torch.zeros((NR, CAP, 6), dtype=torch.int64, device=dev)


def step():
    _lib.check(_lib.lib().bg_fmd_smems_batch64_dev(fm.h, 1, NR, reads.data_ptr(), r_off.data_ptr(), None, MINLEN, L, CAP, d_cnt.data_ptr(), d_out.data_ptr(), None))


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("%-24s %d reads  %.1f ms  %.2f M reads/s  smems %d  checksum %d" % (os.path.basename(os.environ.get("BG_SO", "libbiogpu.so")), NR, dt * 1e3, NR / dt / 1e6,
                                                                       int(d_cnt.sum()), int(d_out.sum() % 1000003)))
