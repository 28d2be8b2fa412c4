"""pairwise::banded::Aligner — host mirror of /root/reference/src/alignment/pairwise/banded.rs:122-1004
on top of the C ABI (bg_align_banded_batch, bg_band_create_batch)."""
import ctypes as C

import numpy as np

from . import _lib
from .pairwise import (MODE_CUSTOM, MODE_GLOBAL, MODE_LOCAL, MODE_SEMIGLOBAL, Scoring, to_alignment)

MAX_CELLS = 5_000_000  # banded.rs:104


class Aligner:
    """banded::Aligner<F> with k-mer length `k` and window `w` (banded.rs:139-267)."""

    def __init__(self, scoring, k, w, ctx=None):
        for nm in ("gap_open", "gap_extend", "xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
            assert getattr(scoring, nm) <= 0, f"{nm} can't be positive"  # banded.rs:215-232
        self.scoring, self.k, self.w, self._ctx = scoring, int(k), int(w), ctx

    @property
    def ctx(self):
        if self._ctx is None or self._ctx is False:
            self._ctx = _lib.default_context()
        return self._ctx

    @staticmethod
    def new(gap_open, gap_extend, match_fn, k, w, ctx=None):  # banded.rs:150
        return Aligner(Scoring.new(gap_open, gap_extend, match_fn), k, w, ctx)

    @staticmethod
    def with_capacity(m, n, gap_open, gap_extend, match_fn, k, w, ctx=None):  # banded.rs:174
        return Aligner(Scoring.new(gap_open, gap_extend, match_fn), k, w, ctx)

    @staticmethod
    def with_scoring(scoring, k, w, ctx=None):  # banded.rs:259
        return Aligner(scoring, k, w, ctx)

    @staticmethod
    def with_capacity_and_scoring(m, n, scoring, k, w, ctx=None):  # banded.rs:208
        return Aligner(scoring, k, w, ctx)

    def get_mut_scoring(self):  # banded.rs:272
        return self.scoring

    def band_create_arrays(self, mode, x, x_off, y, y_off):
        """Band::create per pair (host): returns (band_off, start, end, cells)."""
        xb, yb = _lib.as_u8(x), _lib.as_u8(y)
        xo = np.ascontiguousarray(x_off, dtype=np.uint64)
        yo = np.ascontiguousarray(y_off, dtype=np.uint64)
        n = len(xo) - 1
        ylen = np.diff(yo).astype(np.uint64) + np.uint64(1)
        boff = np.zeros(n + 1, dtype=np.uint64)
        boff[1:] = np.cumsum(ylen)
        start = np.zeros(int(boff[-1]), dtype=np.uint32)
        end = np.zeros(int(boff[-1]), dtype=np.uint32)
        cells = np.zeros(n, dtype=np.uint64)
        sc = self.scoring.to_c()
        _lib.check(_lib.lib().bg_band_create_batch(C.byref(sc), mode, self.k, self.w, n, xb.ctypes.data,
                                                   xo.ctypes.data, yb.ctypes.data, yo.ctypes.data,
                                                   boff.ctypes.data, start.ctypes.data, end.ctypes.data,
                                                   cells.ctypes.data), "bg_band_create_batch")
        return boff, start, end, cells

    def align_arrays(self, mode, x, x_off, y, y_off, want_ops=True):
        xb, yb = _lib.as_u8(x), _lib.as_u8(y)
        xo = np.ascontiguousarray(x_off, dtype=np.uint64)
        yo = np.ascontiguousarray(y_off, dtype=np.uint64)
        n = len(xo) - 1
        out = np.zeros(n, dtype=_lib.ALN_DTYPE)
        cap = int(xo[-1] + yo[-1]) + 4 * n + 8 if want_ops else 0
        ops = np.zeros(max(cap, 1), dtype=np.uint8) if want_ops else None
        used = C.c_uint64(0)
        cells = np.zeros(n, dtype=np.uint64)
        sc = self.scoring.to_c()
        rc = _lib.lib().bg_align_banded_batch(self.ctx.h, C.byref(sc), mode, self.k, self.w, n, xb.ctypes.data,
                                              xo.ctypes.data, yb.ctypes.data, yo.ctypes.data, out.ctypes.data,
                                              ops.ctypes.data if want_ops else None, cap, C.byref(used),
                                              cells.ctypes.data)
        self.last_cells = cells
        self.last_out, self.last_ops = out, ops
        _lib.check(rc, "bg_align_banded_batch")
        return out, ops

    def align_batch(self, mode, xs, ys):
        x, xo = _lib.concat(xs)
        y, yo = _lib.concat(ys)
        out, ops = self.align_arrays(mode, x, xo, y, yo)
        return [to_alignment(out[p], ops) for p in range(len(xs))]

    def custom(self, x, y): return self.align_batch(MODE_CUSTOM, [x], [y])[0]          # banded.rs:282
    def global_(self, x, y): return self.align_batch(MODE_GLOBAL, [x], [y])[0]         # banded.rs:872
    def semiglobal(self, x, y): return self.align_batch(MODE_SEMIGLOBAL, [x], [y])[0]  # banded.rs:901
    def local(self, x, y): return self.align_batch(MODE_LOCAL, [x], [y])[0]            # banded.rs:972
