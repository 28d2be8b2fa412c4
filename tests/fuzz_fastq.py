"""Differential fuzz of the FASTQ ingest (run by hand on a GPU box: python tests/fuzz_fastq.py SEED SECONDS): random texts through
bg_fastq_parse_dev twice — the one-pass kernel (fq_fused_kernel, with the general kernels F1 .. F6 behind it where it raises its
flag) and the general kernels alone (ctx option fq_no_fused) — against the CPU oracle (oracle/fastq.cpp: fastq.rs:266-410): status,
error position, every record's id / description / sequence / quality / Record::check, offsets.
Texts: 1 .. 6000 four-line records whose read lengths come from one of several regimes (a few bases, short reads, 150 bp,
1 - 3 kb, 5 - 40 kb: tiles with hundreds of records, tiles inside one line, lines that begin more than 31 KB in front of the tile
their record ends in), LF or CRLF, headers with and without descriptions / leading and trailing blanks / empty ids, qualities that
begin with '@' or '+', sequences with lower case, '-', '.', '*' and invalid bytes, unequal lengths, a prefix of random length in
front (every tile phase) — and, in a quarter of the rounds, one defect: a byte >= 0x80 somewhere, a wrapped (multi-line) record,
a missing '@', a missing '+', an empty line, a truncated tail, no newline at the end."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle_py as orc  # noqa: E402
from rust_bio_amd import _lib, fastq  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
SEQ_OK = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)
SEQ_ODD = np.frombuffer(b"ACGTN-.*RYKM#5 ", dtype=np.uint8)
t0 = time.time()
rounds = n_rec = n_bytes = n_fail = n_fallback = 0
no_fused = _lib.Context(0)
no_fused.set_option("fq_no_fused", 1)


def rec(k, ln, nl, odd):
    alpha = SEQ_ODD if odd and rng.random() < 0.3 else SEQ_OK
    seq = alpha[rng.integers(0, len(alpha), size=ln)].tobytes()
    qln = ln if rng.random() > (0.05 if odd else 0.0) else max(1, ln + int(rng.integers(-2, 3)))
    qual = rng.integers(33, 75, size=qln).astype(np.uint8)
    if qln and rng.random() < 0.1:
        qual[0] = ord("@") if rng.random() < 0.5 else ord("+")
    hid = b"" if odd and rng.random() < 0.05 else b"r%d" % k if rng.random() < 0.7 else bytes(rng.integers(97, 123, size=int(rng.integers(1, 40))).astype(np.uint8))
    hdr = b"@" + hid
    u = rng.random()
    if u < 0.4:
        hdr += b" " + bytes(rng.integers(97, 123, size=int(rng.integers(0, 30))).astype(np.uint8))
    elif u < 0.5:
        hdr += b"  two  blanks "
    elif u < 0.55 and odd:
        hdr = b"@ " + hid
    if odd and rng.random() < 0.1:
        hdr += b" \t"
    plus = b"+" if rng.random() < 0.8 else b"+" + hid
    tail = b" " if odd and rng.random() < 0.05 else b""
    return hdr + nl + seq + tail + nl + plus + nl + qual.tobytes() + nl


while time.time() - t0 < budget and n_fail == 0:
    rounds += 1
    regime = int(rng.integers(0, 6))
    lo, hi, cnt = [(1, 12, 6000), (20, 60, 4000), (150, 151, 3000), (100, 400, 2000), (1000, 3000, 300), (5000, 40000, 40)][regime]
    n = int(rng.integers(1, cnt + 1))
    nl = b"\r\n" if rng.random() < 0.2 else b"\n"
    odd = rng.random() < 0.4
    parts = [rec(k, int(rng.integers(lo, hi)), nl, odd) for k in range(n)]
    if regime == 3 and rng.random() < 0.5:  # a few long records among short ones
        for _ in range(3):
            parts[int(rng.integers(0, n))] = rec(10**6, int(rng.integers(9000, 34000)), nl, odd)
    pre = int(rng.integers(0, 400))
    text = (b"@p\n" + b"A" * pre + b"\n+\n" + b"I" * pre + b"\n" if rng.random() < 0.7 else b"") + b"".join(parts)
    defect = rng.random()
    if defect < 0.25 and len(text) > 20:
        kind = int(rng.integers(0, 7))
        at = int(rng.integers(0, len(text)))
        if kind == 0:
            text = text[:at] + bytes([int(rng.integers(128, 256))]) + text[at + 1:]
        elif kind == 1:
            j = int(rng.integers(0, n))
            parts[j] = b"@w\nACGT\nACGT\n+\nIIII\nIIII\n"
            text = b"".join(parts)
        elif kind == 2:
            nlp = text.find(b"\n@", at)
            if nlp >= 0:
                text = text[:nlp + 1] + b"x" + text[nlp + 2:]
        elif kind == 3:
            nlp = text.find(b"\n+", at)
            if nlp >= 0:
                text = text[:nlp + 1] + b"-" + text[nlp + 2:]
        elif kind == 4:
            nlp = text.find(b"\n", at)
            if nlp >= 0:
                text = text[:nlp + 1] + nl + text[nlp + 1:]
        elif kind == 5:
            text = text[:at]
        else:
            text = text.rstrip(b"\r\n")
    if not text:
        continue
    want, wst, wpos = orc.fastq_parse(text)
    p = fastq.parse_arrays(text)
    q = fastq.parse_arrays(text, ctx=no_fused)
    ok = (p.status, len(p)) == (wst, len(want)) and (q.status, len(q)) == (wst, len(want))
    if ok and wst != "ok":
        ok = p.err_pos == wpos and q.err_pos == wpos
    if ok:
        ok = bool((p.recs == q.recs).all() and (p.seq_off == q.seq_off).all() and (p.qual_off == q.qual_off).all()
                  and (p.seq[:int(p.seq_off[len(p)])] == q.seq[:int(q.seq_off[len(q)])]).all()
                  and (p.qual[:int(p.qual_off[len(p)])] == q.qual[:int(q.qual_off[len(q)])]).all())
    if ok:
        for k, w in enumerate(want):
            r = p.record(k)
            if (r._id, r._desc, r._seq, r._qual, fastq.CHECK[r._check]) != (w["id"], w["desc"], w["seq"], w["qual"], w["check"]):
                ok = False
                print("MISMATCH record", k, "round", rounds, "regime", regime, (r._id, r._desc[:20], len(r._seq), fastq.CHECK[r._check]),
                      "want", (w["id"], w["desc"][:20], len(w["seq"]), w["check"]), flush=True)
                break
    if not ok:
        n_fail += 1
        fn = "/tmp/fuzz_fastq_fail_%d.fq" % rounds
        open(fn, "wb").write(text)
        print("MISMATCH round", rounds, "regime", regime, "bytes", len(text), "status", p.status, q.status, "want", wst, "records", len(p), len(q), len(want),
              "->", fn, flush=True)
    n_rec += len(want)
    n_bytes += len(text)
print("rounds", rounds, "records", n_rec, "bytes", n_bytes, "failures", n_fail, flush=True)
sys.exit(1 if n_fail else 0)
