"""FASTQ ingest (bg_fastq_parse_dev) on 1 M four-line records of 150 bp, text in HBM: call wall time, GB/s of text.
    [BG_SO=<variant .so>] python tools/exp/time_ingest.py [records] [ctx option=value ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rust_bio_amd import _lib  # noqa: E402

if os.environ.get("BG_SO"):
    _lib.SO_PATH = os.path.abspath(os.environ["BG_SO"])
from rust_bio_amd import fastq, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
ctx = _lib.Context(0)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
text = synth.fastq_text(n, 150, seed=6)
d_text = torch.from_numpy(text).to("cuda:0")
bufs = fastq.alloc_dev(len(text), d_text.device)
for _ in range(3):
    k, st, *_ = fastq.parse_dev(d_text, ctx=ctx, bufs=bufs)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 20
for _ in range(R):
    fastq.parse_dev(d_text, ctx=ctx, bufs=bufs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print("%-24s records %d status %s  %.3f ms per call  %.1f GB/s of text" % (os.path.basename(os.environ.get("BG_SO", "libbiogpu.so")), k, st, dt * 1e3, len(text) / dt / 1e9))
