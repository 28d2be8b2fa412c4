"""pairwise::banded::Aligner — host mirror of /root/reference/src/alignment/pairwise/banded.rs:122-1004
on top of the C ABI (bg_align_banded_batch, bg_band_create_batch, bg_band_from_matches_batch,
bg_align_banded_bands_batch)."""
import ctypes as C

import numpy as np

from . import _lib, sparse
from .pairwise import (MODE_CUSTOM, MODE_GLOBAL, MODE_LOCAL, MODE_SEMIGLOBAL, Scoring, check_defined, to_alignment)

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

    def align_arrays(self, mode, x, x_off, y, y_off, want_ops=True, out=None, ops=None):
        """Host-buffer batch through bg_align_banded_batch.  Returns (records, ops_buf); `out` / `ops` may be the arrays of an
        earlier call of the same shape (reused, like a caller's own Vecs — fresh ones cost a page fault per 4 KB written)."""
        xb, yb = _lib.as_u8(x), _lib.as_u8(y)
        xo = np.ascontiguousarray(x_off, dtype=np.uint64)
        yo = np.ascontiguousarray(y_off, dtype=np.uint64)
        n = len(xo) - 1
        if out is None or len(out) != n:
            out = np.zeros(n, dtype=_lib.ALN_DTYPE)
        cap = int(xo[-1] + yo[-1]) + 4 * n + 8 if want_ops else 0
        if want_ops and (ops is None or len(ops) != max(cap, 1)):
            ops = np.zeros(max(cap, 1), dtype=np.uint8)
        if not want_ops:
            ops = None
        used = C.c_uint64(0)
        cells = np.zeros(n, dtype=np.uint64)
        sc = self.scoring.to_c()
        check_defined(self.scoring, xb, xo, yb, yo)  # a closure that panics on a byte pair of this pair (pairwise.py)
        rc = _lib.lib().bg_align_banded_batch(self.ctx.h, C.byref(sc), mode, self.k, self.w, n, xb.ctypes.data,
                                              xo.ctypes.data, yb.ctypes.data, yo.ctypes.data, out.ctypes.data,
                                              ops.ctypes.data if want_ops else None, cap, C.byref(used),
                                              cells.ctypes.data)
        self.last_cells = cells
        self.last_out, self.last_ops = out, ops
        _lib.check(rc, "bg_align_banded_batch")
        return out, ops

    def align_dev(self, mode, n_pairs, d_x, d_x_off, d_y, d_y_off, d_out, d_ops, ops_stride, stream=0, want_cells=False):
        """Device-resident batch (pointers are ints): results stay in HBM; returns Band::num_cells per pair
        when asked.  Synchronous."""
        cells = np.zeros(n_pairs, dtype=np.uint64) if want_cells else None
        sc = self.scoring.to_c()
        _lib.check(_lib.lib().bg_align_banded_batch_dev(self.ctx.h, C.byref(sc), mode, self.k, self.w, n_pairs, d_x, d_x_off,
                                                        d_y, d_y_off, d_out, d_ops, ops_stride,
                                                        cells.ctypes.data if want_cells else None, stream),
                   "bg_align_banded_batch_dev")
        return cells

    def align_batch(self, mode, xs, ys):
        x, xo = _lib.concat(xs)
        y, yo = _lib.concat(ys)
        out, ops = self.align_arrays(mode, x, xo, y, yo)
        return [to_alignment(out[p], ops) for p in range(len(xs))]

    # ---- entry points that take the k-mer matches / the chain from the caller (banded.rs:294-401, 938-970):
    # the band is built on the host from what they pass, then compute_alignment runs on the device
    def bands_from_matches(self, mode, x_off, y_off, matches, match_off, path=None, path_off=None):
        """Band::create_with_matches / create_from_match_path for a batch -> (band_off, start, end, cells)."""
        xo = np.ascontiguousarray(x_off, dtype=np.uint64)
        yo = np.ascontiguousarray(y_off, dtype=np.uint64)
        n = len(xo) - 1
        mm = np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 2))
        mo = np.ascontiguousarray(match_off, dtype=np.uint64)
        boff = np.zeros(n + 1, dtype=np.uint64)
        boff[1:] = np.cumsum(np.diff(yo).astype(np.uint64) + np.uint64(1))
        start = np.zeros(int(boff[-1]), dtype=np.uint32)
        end = np.zeros(int(boff[-1]), dtype=np.uint32)
        cells = np.zeros(n, dtype=np.uint64)
        pp = po = None
        if path is not None:
            pp = np.ascontiguousarray(path, dtype=np.uint32)
            po = np.ascontiguousarray(path_off, dtype=np.uint64)
        sc = self.scoring.to_c()
        rc = _lib.lib().bg_band_from_matches_batch(C.byref(sc), mode, self.k, self.w, n, xo.ctypes.data, yo.ctypes.data,
                                                   mm.ctypes.data, mo.ctypes.data,
                                                   pp.ctypes.data if pp is not None else None,
                                                   po.ctypes.data if po is not None else None, boff.ctypes.data,
                                                   start.ctypes.data, end.ctypes.data, cells.ctypes.data)
        assert rc != -1, "incoming matches must be sorted / path index out of bounds"  # the reference's panics
        _lib.check(rc, "bg_band_from_matches_batch")
        return boff, start, end, cells

    def align_bands_arrays(self, mode, x, x_off, y, y_off, band_off, band_start, band_end, want_ops=True):
        """compute_alignment (banded.rs:406-869) over explicit bands."""
        xb, yb = _lib.as_u8(x), _lib.as_u8(y)
        xo = np.ascontiguousarray(x_off, dtype=np.uint64)
        yo = np.ascontiguousarray(y_off, dtype=np.uint64)
        bo = np.ascontiguousarray(band_off, dtype=np.uint64)
        bs = np.ascontiguousarray(band_start, dtype=np.uint32)
        be = np.ascontiguousarray(band_end, dtype=np.uint32)
        n = len(xo) - 1
        out = np.zeros(n, dtype=_lib.ALN_DTYPE)
        cap = int(xo[-1] + yo[-1]) + 4 * n + 8 if want_ops else 0
        ops = np.zeros(max(cap, 1), dtype=np.uint8) if want_ops else None
        used = C.c_uint64(0)
        cells = np.zeros(n, dtype=np.uint64)
        sc = self.scoring.to_c()
        rc = _lib.lib().bg_align_banded_bands_batch(self.ctx.h, C.byref(sc), mode, n, xb.ctypes.data, xo.ctypes.data,
                                                    yb.ctypes.data, yo.ctypes.data, bo.ctypes.data, bs.ctypes.data,
                                                    be.ctypes.data, out.ctypes.data,
                                                    ops.ctypes.data if want_ops else None, cap, C.byref(used),
                                                    cells.ctypes.data)
        self.last_cells = cells
        _lib.check(rc, "bg_align_banded_bands_batch")
        return out, ops

    def _with_matches(self, mode, x, y, matches, path=None):
        xb, yb = bytes(x), bytes(y)
        xo, yo = np.array([0, len(xb)], dtype=np.uint64), np.array([0, len(yb)], dtype=np.uint64)
        mm = np.asarray(matches, dtype=np.uint32).reshape(-1, 2)
        bo, bs, be, _ = self.bands_from_matches(mode, xo, yo, mm, [0, len(mm)],
                                                None if path is None else list(path),
                                                None if path is None else [0, len(path)])
        out, ops = self.align_bands_arrays(mode, xb, xo, yb, yo, bo, bs, be)
        return to_alignment(out[0], ops)

    def custom_with_matches(self, x, y, matches):  # banded.rs:313-321
        return self._with_matches(MODE_CUSTOM, x, y, matches)

    def custom_with_match_path(self, x, y, matches, path):  # banded.rs:391-401
        return self._with_matches(MODE_CUSTOM, x, y, matches, path)

    def custom_with_expanded_matches(self, x, y, matches, allowed_mismatches, use_lcskpp_union):  # banded.rs:338-389
        expanded = (sparse.expand_kmer_matches(x, y, self.k, matches, allowed_mismatches)
                    if allowed_mismatches is not None else list(matches))
        if use_lcskpp_union:
            ms = self.scoring.match_scores[0] if self.scoring.match_scores is not None else 2  # banded.rs:105
            path = sparse.sdpkpp_union_lcskpp_path(expanded, self.k, ms, self.scoring.gap_open, self.scoring.gap_extend)
            return self._with_matches(MODE_CUSTOM, x, y, expanded, path)
        return self._with_matches(MODE_CUSTOM, x, y, expanded)

    def custom_with_prehash(self, x, y, y_kmer_hash):  # banded.rs:294-302
        return self._with_matches(MODE_CUSTOM, x, y, sparse.find_kmer_matches_seq2_hashed(x, y_kmer_hash, self.k))

    def semiglobal_with_prehash(self, x, y, y_kmer_hash):  # banded.rs:938-970
        return self._with_matches(MODE_SEMIGLOBAL, x, y, sparse.find_kmer_matches_seq2_hashed(x, y_kmer_hash, self.k))

    def custom(self, x, y): return self.align_batch(MODE_CUSTOM, [x], [y])[0]          # banded.rs:282
    def global_(self, x, y): return self.align_batch(MODE_GLOBAL, [x], [y])[0]         # banded.rs:872
    def semiglobal(self, x, y): return self.align_batch(MODE_SEMIGLOBAL, [x], [y])[0]  # banded.rs:901
    def local(self, x, y): return self.align_batch(MODE_LOCAL, [x], [y])[0]            # banded.rs:972
