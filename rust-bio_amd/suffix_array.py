"""suffix_array(text) — host builder behind bg_suffix_array
(reference: src/data_structures/suffix_array.rs:264-284)."""
import numpy as np

from . import _lib


def suffix_array(text):
    t = _lib.as_u8(text)
    sa = np.zeros(len(t), dtype=np.uint64)
    _lib.check(_lib.lib().bg_suffix_array(t.ctypes.data, len(t), sa.ctypes.data), "suffix_array")
    return sa
