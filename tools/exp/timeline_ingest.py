"""Phase timeline of fq_fused_kernel from a -DFQ_TIMELINE variant build (tools/exp/ko_build.sh fastq_ingest.hip tl -DFQ_TIMELINE):
wall-clock stamps (100 MHz) of thread 0 of every block at the phase boundaries; prints the median / p90 duration of each phase
and the blocks' lifetimes.     BG_SO=tools/exp/_ko/libbiogpu_tl.so python tools/exp/timeline_ingest.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rust_bio_amd import _lib  # noqa: E402

_lib.SO_PATH = os.path.abspath(os.environ["BG_SO"])
from rust_bio_amd import fastq, synth  # noqa: E402

ctx = _lib.Context(0)
text = synth.fastq_text(1_000_000, 150, seed=6)
d_text = torch.from_numpy(text).to("cuda:0")
bufs = fastq.alloc_dev(len(text), d_text.device)
for _ in range(4):
    fastq.parse_dev(d_text, ctx=ctx, bufs=bufs)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.SO_PATH)
n_tiles = (len(text) + 16383) // 16384
tl = np.zeros(16 * 32768, dtype=np.uint64)
rc = lib.bg_debug_fq_timeline(tl.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(tl)))
assert rc == 0, rc
tl = tl.reshape(-1, 16)[:n_tiles, :10].astype(np.int64)
t00 = tl[:, 0].min()
us = (tl - t00) / 100.0
names = ["load+count", "newline list", "look-back 1", "records", "look-back 2", "piece tables", "copy issue", "copy barrier", "record stores"]
print("tiles", n_tiles, "kernel span %.1f us" % (us[:, 9].max()))
for k, nm in enumerate(names):
    d = us[:, k + 1] - us[:, k]
    print("%-14s median %7.2f us   p90 %7.2f   mean %7.2f" % (nm, np.median(d), np.percentile(d, 90), d.mean()))
life = us[:, 9] - us[:, 0]
print("%-14s median %7.2f us   p90 %7.2f   mean %7.2f" % ("block", np.median(life), np.percentile(life, 90), life.mean()))
# how far behind its predecessor's end does a block end / start
print("start(i) - start(i-1024): median %.2f us" % np.median(us[1024:, 0] - us[:-1024, 0]))
conc = [(np.sum((us[:, 0] <= t) & (us[:, 9] > t))) for t in np.linspace(us[:, 9].max() * 0.2, us[:, 9].max() * 0.8, 7)]
print("blocks resident at 7 instants:", conc)
