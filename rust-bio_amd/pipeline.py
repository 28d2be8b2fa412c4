"""Seed-and-extend read mapping (BASELINE configs[4]) — host mirror of `bg_seed_extend_batch[_dev]`.

The reference has no such function; callers compose it from `FMIndex::backward_search`, `Interval::occ` and
`Aligner::semiglobal` (/root/reference/src/lib.rs:129-165, benches/fmindex.rs:20-38).  The composition — which
seeds vote, hit -> proposed read start, per-read dedup, window gather, best-hit reduction, the winners'
operations — runs in HIP kernels behind the C ABI (rust-bio_amd/csrc/seed_extend.hip); its definition is in
include/biogpu.h (the tests hold a CPU statement of it).  This module only marshals arguments."""
import ctypes as C

import numpy as np

from . import _lib
from .pairwise import MIN_SCORE  # noqa: F401  (score of an unmapped read)


class SeedParams:
    def __init__(self, seed_len=20, stride=10, max_occ=16, pad=25):
        self.seed_len, self.stride, self.max_occ, self.pad = seed_len, stride, max_occ, pad

    def to_c(self):
        return _lib.SeedParamsC(self.seed_len, self.stride, self.max_occ, self.pad)


def attach_text(fm, text=None, d_text=None):
    """bg_fm_set_text (host bytes, copied) or bg_fm_set_text_dev (a uint8 cuda tensor, borrowed): all n bytes the
    index was built from, final sentinel included."""
    if d_text is not None:
        fm._text_keepalive = d_text
        _lib.check(_lib.lib().bg_fm_set_text_dev(fm.h, d_text.data_ptr(), d_text.numel()), "bg_fm_set_text_dev")
    else:
        t = _lib.as_u8(text)
        _lib.check(_lib.lib().bg_fm_set_text(fm.h, t.ctypes.data, len(t)), "bg_fm_set_text")


def seed_extend_arrays(fm, scoring, reads, read_off, params=None, want_ops=True, allow_out_of_alphabet=False):
    """Host-buffer batch: returns (hits: SEED_HIT_DTYPE[n], ops: uint8[], winners' operations back to back).
    A seed that reaches a byte outside the index's alphabet raises AlphabetError (the reference's backward_search
    panics there) unless allow_out_of_alphabet: such seeds simply do not vote."""
    params = params or SeedParams()
    rd = _lib.as_u8(reads)
    off = np.ascontiguousarray(read_off, dtype=np.uint64)
    n = len(off) - 1
    hits = np.zeros(n, dtype=_lib.SEED_HIT_DTYPE)
    cap = int(2 * off[-1] + (2 * params.pad + 4) * n) + 8 if want_ops else 0
    ops = np.zeros(max(cap, 1), dtype=np.uint8) if want_ops else None
    used = C.c_uint64(0)
    sc, pc = scoring.to_c(), params.to_c()
    rc = _lib.lib().bg_seed_extend_batch(fm.h, C.byref(sc), C.byref(pc), n, rd.ctypes.data, off.ctypes.data,
                                         hits.ctypes.data, ops.ctypes.data if want_ops else None, cap, C.byref(used))
    if not (rc == -7 and allow_out_of_alphabet):
        _lib.check(rc, "bg_seed_extend_batch")
    return hits, (ops[:used.value] if want_ops else None)


def seed_extend_dev(fm, scoring, n_reads, d_reads, d_read_off, max_read_len, d_hits, d_ops=0, ops_stride=0, params=None,
                    stream=0, totals=None):
    """Device-resident batch (pointers are ints); `totals`, if given, is a uint64[2] numpy array that receives
    (suffix-array rows resolved, candidates aligned)."""
    params = params or SeedParams()
    sc, pc = scoring.to_c(), params.to_c()
    _lib.check(_lib.lib().bg_seed_extend_batch_dev(fm.h, C.byref(sc), C.byref(pc), n_reads, d_reads, d_read_off, max_read_len,
                                                   d_hits, d_ops, ops_stride, totals.ctypes.data if totals is not None else None,
                                                   stream), "bg_seed_extend_batch_dev")
