"""`bio::io::fastq` reading side (io/fastq.rs:153-527) on a FASTQ text held in memory: the records are parsed
on the device (csrc/fastq_ingest.hip) — line index, four-line hypothesis, sequential walk only where it fails —
and the sequences land concatenated with offsets, i.e. in the layout the aligner's batch calls take."""
import ctypes as C

import numpy as np

from . import _lib

STATUS = ["ok", "MissingAt", "IncompleteRecord", "Io"]
CHECK = ["ok", "EmptyId", "NonAsciiSequence", "InvalidSequence", "NonAsciiQualities", "UnequalLength"]


class ReadError(Exception):
    """fastq::ReadError (fastq.rs:113-126)"""

    def __init__(self, kind, pos):
        super().__init__(f"{kind} at byte {pos}")
        self.kind, self.pos = kind, pos


class CheckError(Exception):
    """fastq::CheckError (fastq.rs:129-150)"""

    def __init__(self, kind):
        super().__init__(kind)
        self.kind = kind


class Record:
    """fastq::Record (fastq.rs:309-452)"""

    def __init__(self, id_=b"", desc=None, seq=b"", qual=b"", check_code=0):
        self._id, self._desc, self._seq, self._qual, self._check = id_, desc, seq, qual, check_code

    def id(self):
        return self._id.decode()

    def desc(self):
        return None if self._desc is None else self._desc.decode()

    def seq(self):
        return self._seq

    def qual(self):
        return self._qual

    def is_empty(self):  # fastq.rs:363-365
        return not self._id and self._desc is None and not self._seq and not self._qual

    def check(self):  # fastq.rs:388-410, evaluated on the device with the parse
        if self._check:
            raise CheckError(CHECK[self._check])

    def __eq__(self, o):
        return (self._id, self._desc, self._seq, self._qual) == (o._id, o._desc, o._seq, o._qual)

    def __repr__(self):
        return f"Record(id={self._id!r}, desc={self._desc!r}, seq={self._seq!r}, qual={self._qual!r})"


class Parsed:
    """columns of one parse: recs (bg_fastq_record_t), seq/qual (concatenated), seq_off/qual_off (n+1)"""

    def __init__(self, text, recs, seq, seq_off, qual, qual_off, status, err_pos):
        self.text, self.recs, self.seq, self.seq_off, self.qual, self.qual_off = text, recs, seq, seq_off, qual, qual_off
        self.status, self.err_pos = STATUS[status], err_pos

    def __len__(self):
        return len(self.recs)

    def record(self, k):
        r, t = self.recs[k], self.text
        return Record(t[int(r["id_off"]):int(r["id_off"]) + int(r["id_len"])].tobytes(),
                      t[int(r["desc_off"]):int(r["desc_off"]) + int(r["desc_len"])].tobytes() if r["has_desc"] else None,
                      self.seq[int(self.seq_off[k]):int(self.seq_off[k + 1])].tobytes(),
                      self.qual[int(self.qual_off[k]):int(self.qual_off[k + 1])].tobytes(), int(r["check"]))


def parse_arrays(text, ctx=None):
    """All records up to the end of the text or the first ReadError (Parsed.status / err_pos)."""
    ctx = ctx or _lib.default_context()
    t = np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
    cap = len(t) // 4 + 2
    recs = np.zeros(cap, dtype=_lib.FQREC_DTYPE)
    seq = np.zeros(max(1, len(t)), dtype=np.uint8)
    qual = np.zeros(max(1, len(t)), dtype=np.uint8)
    so = np.zeros(cap + 1, dtype=np.uint64)
    qo = np.zeros(cap + 1, dtype=np.uint64)
    n, st, ep = C.c_uint64(0), C.c_int32(0), C.c_uint64(0)
    _lib.check(_lib.lib().bg_fastq_parse(ctx.h, t.ctypes.data, len(t), recs.ctypes.data, cap, seq.ctypes.data, so.ctypes.data,
                                         qual.ctypes.data, qo.ctypes.data, C.byref(n), C.byref(st), C.byref(ep)), "bg_fastq_parse")
    k = int(n.value)
    return Parsed(t, recs[:k], seq[:int(so[k])], so[:k + 1], qual[:int(qo[k])], qo[:k + 1], st.value, int(ep.value))


def parse_dev(d_text, ctx=None, stream=0, bufs=None):
    """d_text: uint8 torch tensor on the device.  Returns (n_records, status name, err_pos, d_recs, d_seq, d_seq_off,
    d_qual, d_qual_off) with everything but the first three left in HBM (d_seq_off is an int64 tensor usable as x_off).
    `bufs`: the five tensors of `alloc_dev(len)` to reuse across calls (a caller's own buffers)."""
    import torch
    ctx = ctx or _lib.default_context()
    ln = int(d_text.numel())
    cap = ln // 4 + 2
    d_recs, d_seq, d_qual, d_so, d_qo = bufs if bufs is not None else alloc_dev(ln, d_text.device)
    n, st, ep = C.c_uint64(0), C.c_int32(0), C.c_uint64(0)
    _lib.check(_lib.lib().bg_fastq_parse_dev(ctx.h, d_text.data_ptr(), ln, d_recs.data_ptr(), cap, d_seq.data_ptr(), d_so.data_ptr(),
                                             d_qual.data_ptr(), d_qo.data_ptr(), C.byref(n), C.byref(st), C.byref(ep), stream),
               "bg_fastq_parse_dev")
    k = int(n.value)
    return k, STATUS[st.value], int(ep.value), d_recs[:k * 56], d_seq, d_so[:k + 1], d_qual, d_qo[:k + 1]


def alloc_dev(ln, device):
    """output buffers of parse_dev for a text of `ln` bytes: records, sequences, qualities, their offsets"""
    import torch
    cap = ln // 4 + 2
    return (torch.empty(cap * 56, dtype=torch.uint8, device=device), torch.empty(max(1, ln), dtype=torch.uint8, device=device),
            torch.empty(max(1, ln), dtype=torch.uint8, device=device), torch.empty(cap + 1, dtype=torch.int64, device=device),
            torch.empty(cap + 1, dtype=torch.int64, device=device))


class Reader:
    """fastq::Reader over an in-memory text (`Reader::new(&[u8])`, fastq.rs:170-176)."""

    def __init__(self, text, ctx=None):
        self._parsed = parse_arrays(text, ctx)

    def records(self):
        """`Reader::records()` (fastq.rs:218-220): yields Records; raises ReadError where the iterator yields Err."""
        p = self._parsed
        for k in range(len(p)):
            yield p.record(k)
        if p.status != "ok":
            raise ReadError(p.status, p.err_pos)
