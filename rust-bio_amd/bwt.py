"""bwt(), less(), Occ — reference: src/data_structures/bwt.rs:39-49, 186-199, 76-183."""
import ctypes as C

import numpy as np

from . import _lib


def bwt(text, pos):
    t = _lib.as_u8(text)
    sa = np.ascontiguousarray(pos, dtype=np.uint64)
    assert len(t) == len(sa)  # bwt.rs:40
    out = np.zeros(len(t), dtype=np.uint8)
    _lib.check(_lib.lib().bg_bwt(t.ctypes.data, sa.ctypes.data, len(t), out.ctypes.data), "bwt")
    return out


def less(bwt_arr, alphabet):
    b = _lib.as_u8(bwt_arr)
    a = _lib.as_u8(bytes(alphabet))
    ln = C.c_uint32(0)
    _lib.check(_lib.lib().bg_less(None, 0, a.ctypes.data, len(a), None, C.byref(ln)), "less")
    out = np.zeros(ln.value, dtype=np.uint64)
    _lib.check(_lib.lib().bg_less(b.ctypes.data, len(b), a.ctypes.data, len(a), out.ctypes.data,
                                  C.byref(ln)), "less")
    return out


class Occ:
    """Occ::new(&bwt, k, &alphabet) (bwt.rs:94-125).  The sampled table itself is built in the
    engine's own block layout when the FMIndex is constructed (bg_fm_build); this object only
    carries the parameters, as `Occ::get`'s result does not depend on k."""

    def __init__(self, bwt_arr, k, alphabet):
        if k < 1:
            raise ValueError("k must be >= 1")
        self.k = int(k)
        self.alphabet = bytes(alphabet)
        b = _lib.as_u8(bwt_arr)
        m = max(self.alphabet) + 1
        if len(b) and int(b.max()) >= m:  # bwt.rs:114: curr_occ[c as usize] out of bounds
            raise _lib.AlphabetError(-7, "Occ::new")
