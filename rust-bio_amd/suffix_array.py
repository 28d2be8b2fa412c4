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
