"""A/B of the FM search's two-queries-per-quad kernel (fm option "ilp": queries per quad): python tools/exp/time_fm_ilp.py [genome_bp] [queries] [pattern_len]
Builds the index on the device, runs the byte-pattern search (and the packed one) with ilp = 1 .. 4, compares every result array."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import torch
torch.cuda.init()
from rust_bio_amd import _lib, synth_gpu, pack2
if os.environ.get("BG_SO"):  # a variant build of the library (A/B of compile-time choices)
    _lib.SO_PATH = os.path.abspath(os.environ["BG_SO"])
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.suffix_array import bwt_dev, suffix_array_dev
n_g = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
g_dev = synth_gpu.genome(n_g, seed=33, device=dev)
d_sa = suffix_array_dev(g_dev, ctx=ctx)
d_b = bwt_dev(g_dev, d_sa, ctx=ctx)
fm = FMIndex.from_device(d_b, 128, b"ACGTNacgtn", ctx=ctx)
del d_sa, d_b
pat, off = synth_gpu.fm_patterns(g_dev, n_q, P, seed=34)
bufs = [torch.empty(n_q, dtype=torch.uint8, device=dev), torch.empty(n_q, dtype=torch.int64, device=dev),
        torch.empty(n_q, dtype=torch.int64, device=dev), torch.empty(n_q, dtype=torch.int32, device=dev)]
ppk, bad = pack2.pack_dev(pat, codes=fm.pattern_codes(), ctx=ctx, stream=0)
soff = torch.arange(n_q + 1, dtype=torch.int64, device=dev) * P
def run(packed):
    for b in bufs:
        b.fill_(77)
    args = (n_q, (ppk if packed else pat).data_ptr(), (soff if packed else off).data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), 0)
    f = fm.backward_search_packed_dev if packed else fm.backward_search_dev
    f(*args); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); f(*args); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), [b.clone() for b in bufs]
for packed in (False, True):
    fm.set_option("ilp", 1)
    t0, r0 = run(packed)
    line = "%s genome %d queries %d x %d: quad %.3f ms (%.1f M q/s)" % ("packed" if packed else "bytes ", n_g, n_q, P, t0 * 1e3, n_q / t0 / 1e6)
    for ilp in (2,):
        fm.set_option("ilp", ilp)
        t1, r1 = run(packed)
        same = all(bool((a == b).all().item()) for a, b in zip(r0, r1))
        line += "  ilp%d %.3f ms (%.1f M q/s, equal %s)" % (ilp, t1 * 1e3, n_q / t1 / 1e6, same)
    print(line, flush=True)
