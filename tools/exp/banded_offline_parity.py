"""One offline pass of BASELINE configs[3] against the oracle, ALL pairs: the 100 000 x 10 kb pairs of
bench.py's banded leg (same generator, same seed) through bg_align_banded_batch_dev — band builder, K3v2 end strips, K3p
interior runs, K4 — compared record by record and operation by operation with oracle/banded.cpp on the box's host threads.
    python tools/exp/banded_offline_parity.py [pairs=100000] [chunk=4096] > gpurun_out/r05_banded_offline_parity.json
(~4 minutes of the 16-thread oracle; bench.py's own run checks 5 120 of them.)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import oracle_py as orc  # noqa: E402
from csrc_hash import csrc_sha  # noqa: E402
from rust_bio_amd import synth_gpu  # noqa: E402
from rust_bio_amd.banded import Aligner  # noqa: E402
from rust_bio_amd.pairwise import Scoring  # noqa: E402

Pb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Lb, kb, wb = 10_000, 16, 32
dev = torch.device("cuda:0")
threads = len(os.sched_getaffinity(0))
try:  # cgroup CPU quota: threads beyond it are only throttled (run A: 256 "cores" on a 16-CPU quota made the oracle 2.5 x slower)
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if quota != "max":
        threads = min(threads, max(1, -(-int(quota) // int(period))))
except (OSError, ValueError):
    pass
bx, _, by, _ = synth_gpu.sw_pairs_big(Pb, Lb, seed=4, device=dev, sub=0.06, ins=0.02, dele=0.02, chunk=64)
hx, hy = bx.cpu().numpy(), by.cpu().numpy()
hoff = np.arange(Pb + 1, dtype=np.uint64) * np.uint64(Lb)
bal = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1), kb, wb)
t0 = time.time()
bout, bops = bal.align_arrays(2, hx, hoff, hy, hoff)
t_gpu = time.time() - t0
redo = bal.ctx.band_redo_pairs()
cells = bal.last_cells.copy()
osc = orc.make_scoring(-5, -1, 1, -1)
bad_rec = bad_ops = bad_cells = 0
first_bad = None
t0 = time.time()
for lo in range(0, Pb, chunk):
    hi = min(Pb, lo + chunk)
    n = hi - lo
    oout, oops, ostride, ocells = orc.banded_align_batch(osc, "semiglobal", kb, wb, hx[lo * Lb:hi * Lb], hoff[:n + 1], hy[lo * Lb:hi * Lb],
                                                         hoff[:n + 1], threads=threads)
    rec_ok = np.ones(n, dtype=bool)
    for f in ("score", "xstart", "xend", "ystart", "yend", "n_ops"):
        rec_ok &= bout[f][lo:hi].astype(np.int64) == oout[f].astype(np.int64)
    rec_ok &= bout["status"][lo:hi] == 0
    bad_rec += int((~rec_ok).sum())
    bad_cells += int((cells[lo:hi] != ocells).sum())
    kind = (oops.reshape(n, ostride) & 0xFF).astype(np.uint8)
    for p in range(n):
        kq, oq = int(bout["n_ops"][lo + p]), int(bout["ops_off"][lo + p])
        if not rec_ok[p] or not (bops[oq:oq + kq] == kind[p, :kq]).all():
            bad_ops += 1
            if first_bad is None:
                first_bad = lo + p
    del oout, oops, kind
t_cpu = time.time() - t0
print(json.dumps({"workload": f"{Pb} x 10 kb pairs (bench.py's banded leg: synth_gpu.sw_pairs_big seed 4, 6% sub, 2% ins, 2% del), "
                              f"banded::Aligner::semiglobal, k-mer {kb}, w {wb}",
                  "banded_pairs_checked_offline": Pb, "records_differ": bad_rec, "operation_lists_differ": bad_ops,
                  "band_cell_counts_differ": bad_cells, "first_bad_pair": first_bad, "bit_exact": bad_rec == 0 and bad_ops == 0 and bad_cells == 0,
                  "pairs_recomputed_by_int32_kernels": int(redo), "engine_s": round(t_gpu, 2), "oracle_s": round(t_cpu, 1),
                  "oracle_threads": threads, "band_cells": int(cells.sum()), "csrc_sha": csrc_sha(ROOT)}))
