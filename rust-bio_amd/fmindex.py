"""FMIndex / backward_search — reference: src/data_structures/fmindex.rs:69-248."""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _lib

COMPLETE, PARTIAL, ABSENT, PANIC = 0, 1, 2, 3


@dataclass(frozen=True)
class Interval:  # fmindex.rs:69-80
    lower: int
    upper: int

    def occ(self, sa):
        """Interval::occ (fmindex.rs:75-79); `sa` is a device-attached Raw/SampledSuffixArray or a host array."""
        if hasattr(sa, "fm"):
            return [int(v) for v in sa.fm.interval_occ_arrays([self.lower], [self.upper])[1]]
        return [int(sa[p]) for p in range(self.lower, self.upper)]


@dataclass(frozen=True)
class BackwardSearchResult:  # fmindex.rs:92-96
    kind: str  # "Complete" | "Partial" | "Absent"
    interval: Interval = None
    matched_len: int = 0

    @staticmethod
    def from_raw(tag, lo, hi, ml):
        if tag == COMPLETE:
            return BackwardSearchResult("Complete", Interval(int(lo), int(hi)), int(ml))
        if tag == PARTIAL:
            return BackwardSearchResult("Partial", Interval(int(lo), int(hi)), int(ml))
        if tag == ABSENT:
            return BackwardSearchResult("Absent")
        raise _lib.AlphabetError(-7, "backward_search")


class FMIndex:
    """FMIndex::new(bwt, less, occ) (fmindex.rs:245-247): uploads the index to the device."""

    def __init__(self, bwt_arr, less_arr, occ, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self._bwt = _lib.as_u8(bwt_arr)
        self._less = np.ascontiguousarray(less_arr, dtype=np.uint64)
        alpha = _lib.as_u8(occ.alphabet)
        self.h = C.c_void_p()
        _lib.check(_lib.lib().bg_fm_build(self.ctx.h, self._bwt.ctypes.data, len(self._bwt),
                                          self._less.ctypes.data, len(self._less), occ.k,
                                          alpha.ctypes.data, len(alpha), C.byref(self.h)),
                   "FMIndex::new")

    @classmethod
    def from_device(cls, d_bwt, k, alphabet, ctx=None, stream=0):
        """bg_fm_build_dev: the index from a BWT that lives in HBM (a uint8 cuda tensor, e.g. bwt_dev's); `less` is
        computed on the way and kept in `self._less`."""
        self = cls.__new__(cls)
        self.ctx = ctx or _lib.default_context()
        if k < 1:
            raise ValueError("k must be >= 1")
        alpha = _lib.as_u8(bytes(alphabet))
        self._less = np.zeros(int(alpha.max()) + 2, dtype=np.uint64)
        self._bwt = None
        self._d_bwt = d_bwt
        self.h = C.c_void_p()
        _lib.check(_lib.lib().bg_fm_build_dev(self.ctx.h, d_bwt.data_ptr(), d_bwt.numel(), k, alpha.ctypes.data, len(alpha),
                                              self._less.ctypes.data, C.byref(self.h), stream), "FMIndex::new (device)")
        return self

    def __len__(self):
        """the text length n (bwt.len() of the reference's FMIndex)"""
        n = C.c_uint64(0)
        _lib.check(_lib.lib().bg_fm_len(self.h, C.byref(n)), "bg_fm_len")
        return int(n.value)

    def bwt(self):
        """the BWT bytes; a handle that was loaded (or built from a device BWT) reads them back out of its rank
        blocks (bg_fm_bwt) — what FMDIndex::from needs of a deserialized FMIndex (fmindex.rs:311-329)"""
        if self._bwt is None:
            b = np.empty(len(self), dtype=np.uint8)
            _lib.check(_lib.lib().bg_fm_bwt(self.h, b.ctypes.data), "bg_fm_bwt")
            self._bwt = b
        return self._bwt

    def less(self):
        """the `less` array the index answers with (bwt.rs:186-199)"""
        if self._less is None:
            ln = C.c_uint32(0)
            _lib.check(_lib.lib().bg_fm_less(self.h, None, C.byref(ln)), "bg_fm_less")
            ls = np.zeros(ln.value, dtype=np.uint64)
            _lib.check(_lib.lib().bg_fm_less(self.h, ls.ctypes.data, C.byref(ln)), "bg_fm_less")
            self._less = ls
        return self._less

    def set_option(self, key, value):
        _lib.check(_lib.lib().bg_fm_set_option(self.h, key.encode(), int(value)), "bg_fm_set_option")

    def device_bytes(self):
        return int(_lib.lib().bg_fm_device_bytes(self.h))

    def step2_bytes(self):
        """bytes of the 2-step rank blocks the searches use (0: single steps)"""
        return int(_lib.lib().bg_fm_step2_bytes(self.h))

    def backward_search_arrays(self, pat, pat_off, out=None):
        """Batch over concatenated patterns; returns (tag u8, lower u64, upper u64, matched u32).
        Raises AlphabetError if any query reached a byte outside the alphabet.  `out`: the four arrays of an earlier
        call of the same size (reused, like a caller's own Vecs — fresh arrays are paged in while they are filled)."""
        p = _lib.as_u8(pat)
        off = np.ascontiguousarray(pat_off, dtype=np.uint64)
        n = len(off) - 1
        if out is not None and len(out[0]) == n:
            tag, lo, hi, ml = out
        else:
            tag = np.zeros(n, dtype=np.uint8)
            lo = np.zeros(n, dtype=np.uint64)
            hi = np.zeros(n, dtype=np.uint64)
            ml = np.zeros(n, dtype=np.uint32)
        rc = _lib.lib().bg_fm_backward_search_batch(self.h, n, p.ctypes.data, off.ctypes.data,
                                                    tag.ctypes.data, lo.ctypes.data,
                                                    hi.ctypes.data, ml.ctypes.data)
        self.last_raw = (tag, lo, hi, ml)
        _lib.check(rc, "backward_search")
        return tag, lo, hi, ml

    def backward_search_batch(self, patterns):
        buf, off = _lib.concat(patterns)
        tag, lo, hi, ml = self.backward_search_arrays(buf, off)
        return [BackwardSearchResult.from_raw(*r) for r in zip(tag, lo, hi, ml)]

    def backward_search(self, pattern):
        """FMIndexable::backward_search (fmindex.rs:144-208) for one pattern (batch of 1)."""
        return self.backward_search_batch([bytes(pattern)])[0]

    def backward_search_dev(self, n_q, d_pat, d_off, d_tag, d_lo, d_hi, d_ml, stream=0):
        """Device-resident batch: arguments are device pointers (ints), asynchronous."""
        _lib.check(_lib.lib().bg_fm_backward_search_batch_dev(self.h, n_q, d_pat, d_off, d_tag,
                                                              d_lo, d_hi, d_ml, stream),
                   "backward_search_dev")

    def pattern_codes(self):
        """the byte value of each 2-bit code of this index (what its patterns are packed with: pack2.pack_dev)"""
        cb = (C.c_uint8 * 4)()
        _lib.check(_lib.lib().bg_fm_pattern_codes(self.h, cb), "bg_fm_pattern_codes")
        return bytes(cb)

    def backward_search_packed_dev(self, n_q, d_packed, d_sym_off, d_tag, d_lo, d_hi, d_ml, stream=0):
        """backward_search_dev on a 2-bit pattern stream (pack2.py); d_sym_off: symbol offsets (n_q + 1)"""
        _lib.check(_lib.lib().bg_fm_backward_search_packed_dev(self.h, n_q, d_packed, d_sym_off, d_tag, d_lo, d_hi, d_ml, stream),
                   "backward_search_packed_dev")

    def backward_search_count_lines_dev(self, n_q, d_pat, d_off, d_tag, d_lo, d_hi, d_ml, stream=0):
        """the same search with its 64-byte block loads counted (synchronous) -> lines"""
        lines = C.c_uint64(0)
        _lib.check(_lib.lib().bg_fm_backward_search_count_lines_dev(self.h, n_q, d_pat, d_off, d_tag, d_lo, d_hi, d_ml,
                                                                    C.byref(lines), stream), "backward_search_count_lines_dev")
        return int(lines.value)

    def interval_occ_arrays(self, lower, upper):
        """Interval::occ for a batch: returns (out_off, positions); needs an attached suffix array."""
        lo = np.ascontiguousarray(lower, dtype=np.uint64)
        hi = np.ascontiguousarray(upper, dtype=np.uint64)
        n = len(lo)
        off = np.zeros(n + 1, dtype=np.uint64)
        total = int(np.maximum(hi.astype(np.int64) - lo.astype(np.int64), 0).sum())
        pos = np.zeros(max(total, 1), dtype=np.uint64)
        _lib.check(_lib.lib().bg_interval_occ_batch(self.h, n, lo.ctypes.data, hi.ctypes.data, off.ctypes.data,
                                                    pos.ctypes.data, total), "Interval::occ")
        return off, pos[:total]

    def interval_occ_dev(self, n_iv, d_lower, d_out_off, total, d_pos, stream=0):
        _lib.check(_lib.lib().bg_interval_occ_batch_dev(self.h, n_iv, d_lower, d_out_off, total, d_pos, stream),
                   "Interval::occ (dev)")

    def save(self, path):
        """Serialize (the reference derives it for FMIndex / Occ / SampledSuffixArray: fmindex.rs:214, bwt.rs:76,
        suffix_array.rs:124): bg_fm_save — BWT, less, alphabet, k, the attached suffix array, the owned text."""
        _lib.check(_lib.lib().bg_fm_save(self.h, os.fsencode(path)), "FMIndex::serialize")

    @classmethod
    def load(cls, path, ctx=None):
        """Deserialize: bg_fm_load — the handle answers like the saved one (search, Interval::occ, seed-and-extend)."""
        self = cls.__new__(cls)
        self.ctx = ctx or _lib.default_context()
        self._bwt = None
        self._less = None
        self._d_bwt = None
        self.h = C.c_void_p()
        _lib.check(_lib.lib().bg_fm_load(self.ctx.h, os.fsencode(path), C.byref(self.h)), "FMIndex::deserialize")
        self.less()  # (bwt() stays lazy: n bytes)
        return self

    def close(self):
        if self.h:
            _lib.lib().bg_fm_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass(frozen=True)
class BiInterval:  # fmindex.rs:250-283
    lower: int
    lower_rev: int
    size: int
    match_size: int

    def forward(self):
        return Interval(self.lower, self.lower + self.size)

    def revcomp(self):
        return Interval(self.lower_rev, self.lower_rev + self.size)


class FMDIndex:
    """FMDIndex::from(fmindex) (fmindex.rs:311-329): bi-directional search over an FM index of T$R$...
    `smems` / `all_smems` (363-501) run on the device; results are (BiInterval, position, length)."""

    def __init__(self, fmindex, records32=False, check=True):
        if check:  # (check=False: a caller that knows — the engine refuses a BWT outside the alphabet anyway, BG_ERR_UNSUPPORTED)
            b = fmindex.bwt()
            ok = np.isin(b, np.frombuffer(b"ACGTNacgtn$", dtype=np.uint8)).all()
            assert ok, "Expecting BWT over the DNA alphabet (including N) with the sentinel $."
        self.fm = fmindex
        self.records32 = records32

    def smems_arrays(self, pat, pat_off, i_pos, l, all_=False, cap=None):
        p = _lib.as_u8(pat)
        off = np.ascontiguousarray(pat_off, dtype=np.uint64)
        n = len(off) - 1
        if cap is None:
            cap = int(np.diff(off).max()) + 1 if n else 1
        ip = np.ascontiguousarray(i_pos, dtype=np.uint32) if i_pos is not None else None
        cnt = np.zeros(n, dtype=np.uint32)
        # (BiInterval is usize in the reference, fmindex.rs:254-259: the uint64-record entry point serves both index layouts;
        #  records32=True keeps the uint32 flavour of rounds 2-5 reachable for the tests)
        if self.records32:
            out = np.zeros((n, cap, 6), dtype=np.uint32)
            fn = _lib.lib().bg_fmd_smems_batch
        else:
            out = np.zeros((n, cap, 6), dtype=np.uint64)
            fn = _lib.lib().bg_fmd_smems_batch64
        rc = fn(self.fm.h, 1 if all_ else 0, n, p.ctypes.data, off.ctypes.data,
                ip.ctypes.data if ip is not None else None, l, cap, cnt.ctypes.data, out.ctypes.data)
        _lib.check(rc, "FMDIndex::smems")
        return cnt, out

    @staticmethod
    def _decode(cnt, out, q):
        return [(BiInterval(int(r[0]), int(r[1]), int(r[2]), int(r[3])), int(r[4]), int(r[5])) for r in out[q, :cnt[q]]]

    def smems_batch(self, patterns, positions, l):
        buf, off = _lib.concat(patterns)
        cnt, out = self.smems_arrays(buf, off, positions, l)
        return [self._decode(cnt, out, q) for q in range(len(patterns))]

    def all_smems_batch(self, patterns, l):
        buf, off = _lib.concat(patterns)
        cnt, out = self.smems_arrays(buf, off, None, l, all_=True)
        return [self._decode(cnt, out, q) for q in range(len(patterns))]

    def _interval(self, op, iv=None, a=0):
        dt = np.uint32 if self.records32 else np.uint64
        if iv is None:
            ivn = np.zeros((1, 4), dtype=dt)
        elif isinstance(iv, BiInterval):
            ivn = np.array([[iv.lower, iv.lower_rev, iv.size, iv.match_size]], dtype=dt)
        else:
            ivn = np.array([list(iv)], dtype=dt)
        out = np.zeros((1, 4), dtype=dt)
        ops = np.array([op], dtype=np.uint8)
        sym = np.array([a], dtype=np.uint8)
        fn = _lib.lib().bg_fmd_interval_batch if self.records32 else _lib.lib().bg_fmd_interval_batch64
        _lib.check(fn(self.fm.h, 1, ops.ctypes.data, ivn.ctypes.data, sym.ctypes.data, out.ctypes.data), "FMDIndex interval")
        return BiInterval(*(int(v) for v in out[0]))

    def init_interval(self): return self._interval(0)                    # fmindex.rs:517
    def init_interval_with(self, a): return self._interval(1, None, a)   # fmindex.rs:504
    def backward_ext(self, iv, a): return self._interval(2, iv, a)       # fmindex.rs:527
    def forward_ext(self, iv, a): return self._interval(3, iv, a)        # fmindex.rs:560

    def smems(self, pattern, i, l):  # fmindex.rs:363
        return self.smems_batch([bytes(pattern)], [i], l)[0]

    def all_smems(self, pattern, l):  # fmindex.rs:479
        return self.all_smems_batch([bytes(pattern)], l)[0]
