"""suffix_array(text) — host builder behind bg_suffix_array
(reference: src/data_structures/suffix_array.rs:264-284)."""
import numpy as np

from . import _lib


def suffix_array(text):
    t = _lib.as_u8(text)
    sa = np.zeros(len(t), dtype=np.uint64)
    _lib.check(_lib.lib().bg_suffix_array(t.ctypes.data, len(t), sa.ctypes.data), "suffix_array")
    return sa


def sentinel(text):
    """suffix_array.rs `sentinel()`: the last byte of the text."""
    t = _lib.as_u8(text)
    return int(t[-1])


class SampledSuffixArray:
    """`RawSuffixArray::sample` (suffix_array.rs:86-120) + `SampledSuffixArray::get` (157-184).

    The samples and the extra (sentinel) rows are built on the host exactly like the reference;
    `get` runs on the device once the array is attached to an FMIndex over the same text."""

    def __init__(self, sa, text, bwt_arr, sampling_rate, fmindex=None):
        sa = np.ascontiguousarray(sa, dtype=np.uint64)
        b = _lib.as_u8(bwt_arr)
        self.s = int(sampling_rate)
        self.sentinel = sentinel(text)
        self.n = len(sa)
        self.sample = np.ascontiguousarray(sa[::self.s])
        rows = np.nonzero((b == self.sentinel) & (np.arange(self.n) % self.s != 0))[0].astype(np.uint64)
        self.extra_rows = rows
        self.extra_pos = np.ascontiguousarray(sa[rows.astype(np.intp)])
        self.fm = None
        if fmindex is not None:
            self.attach(fmindex)

    def attach(self, fmindex):
        _lib.check(_lib.lib().bg_fm_set_sampled_suffix_array(
            fmindex.h, self.sample.ctypes.data, len(self.sample), self.s, self.sentinel,
            self.extra_rows.ctypes.data, self.extra_pos.ctypes.data, len(self.extra_rows)), "SampledSuffixArray")
        self.fm = fmindex
        return self

    def sampling_rate(self):
        return self.s

    def __len__(self):
        return self.n

    def get_batch(self, index):
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        out = np.zeros(len(idx), dtype=np.uint64)
        _lib.check(_lib.lib().bg_sa_get_batch(self.fm.h, len(idx), idx.ctypes.data, out.ctypes.data), "SuffixArray::get")
        return out

    def get(self, index):
        v = int(self.get_batch([index])[0])
        return None if v == NONE else v


class RawSuffixArray:
    """`RawSuffixArray` (suffix_array.rs:25,134-141) resident on the device."""

    def __init__(self, sa, fmindex):
        self.sa = np.ascontiguousarray(sa, dtype=np.uint64)
        self.n = len(self.sa)
        _lib.check(_lib.lib().bg_fm_set_suffix_array(fmindex.h, self.sa.ctypes.data, self.n), "RawSuffixArray")
        self.fm = fmindex

    def __len__(self):
        return self.n

    def sample(self, text, bwt_arr, sampling_rate):
        return SampledSuffixArray(self.sa, text, bwt_arr, sampling_rate)

    get_batch = SampledSuffixArray.get_batch
    get = SampledSuffixArray.get


NONE = 0xFFFFFFFFFFFFFFFF


def suffix_array_dev(d_text, ctx=None, stream=0, wide=None):
    """`suffix_array` for a text that lives in HBM (a uint8 cuda tensor ending in a unique, smallest sentinel):
    returns the suffix array as a cuda tensor of n uint32 values (dtype torch.int32 holds the bits; view it as
    torch.uint32 or widen with `.to(torch.int64) & 0xFFFFFFFF`).  bg_suffix_array_dev.
    wide (default: n >= 2^32 - 1): 64-bit positions — a torch.int64 tensor, bg_suffix_array_dev64 (suffix_array.rs:264: usize)."""
    import torch
    ctx = ctx or _lib.default_context()
    n = d_text.numel()
    if wide is None:
        wide = n >= 0xFFFFFFFF
    if wide:
        d_sa = torch.empty(n, dtype=torch.int64, device=d_text.device)
        _lib.check(_lib.lib().bg_suffix_array_dev64(ctx.h, d_text.data_ptr(), n, d_sa.data_ptr(), stream), "suffix_array (device, 64-bit)")
        return d_sa
    d_sa = torch.empty(n, dtype=torch.int32, device=d_text.device)
    _lib.check(_lib.lib().bg_suffix_array_dev(ctx.h, d_text.data_ptr(), n, d_sa.data_ptr(), stream), "suffix_array (device)")
    return d_sa


def bwt_dev(d_text, d_sa, ctx=None, stream=0):
    """`bwt(text, pos)` (bwt.rs:39-49) on the device: a uint8 cuda tensor."""
    import torch
    ctx = ctx or _lib.default_context()
    n = d_text.numel()
    d_bwt = torch.empty(n, dtype=torch.uint8, device=d_text.device)
    fn = _lib.lib().bg_bwt_dev64 if d_sa.dtype == torch.int64 else _lib.lib().bg_bwt_dev  # (the suffix array's width says which)
    _lib.check(fn(ctx.h, d_text.data_ptr(), d_sa.data_ptr(), n, d_bwt.data_ptr(), stream), "bwt (device)")
    return d_bwt


def sample_dev(d_sa, d_bwt, sentinel_byte, sampling_rate, ctx=None, stream=0):
    """`RawSuffixArray::sample` (suffix_array.rs:86-120) from device arrays: a SampledSuffixArray ready to attach."""
    import ctypes as C
    ctx = ctx or _lib.default_context()
    n = d_sa.numel()
    s = SampledSuffixArray.__new__(SampledSuffixArray)
    s.s, s.sentinel, s.n, s.fm = int(sampling_rate), int(sentinel_byte), n, None
    s.sample = np.zeros((n + s.s - 1) // s.s, dtype=np.uint64)
    cap = 1 << 16
    rows, pos = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint64)
    ne = C.c_uint64(0)
    import torch
    fn = _lib.lib().bg_sa_sample_dev64 if d_sa.dtype == torch.int64 else _lib.lib().bg_sa_sample_dev
    _lib.check(fn(ctx.h, d_sa.data_ptr(), d_bwt.data_ptr(), n, s.s, s.sentinel, s.sample.ctypes.data,
                                           rows.ctypes.data, pos.ctypes.data, cap, C.byref(ne), stream), "RawSuffixArray::sample (device)")
    s.extra_rows, s.extra_pos = rows[:ne.value].copy(), pos[:ne.value].copy()
    return s
