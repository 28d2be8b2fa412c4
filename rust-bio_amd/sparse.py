"""alignment::sparse — host mirror of /root/reference/src/alignment/sparse.rs (the parts the banded
aligner's entry points take or produce: k-mer matches, chains, expansion)."""
import ctypes as C

import numpy as np

from . import _lib

_PANIC = 0xFFFFFFFFFFFFFFFF


def _xy(matches):
    return np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 2))


def _pairs(arr):
    return [(int(a), int(b)) for a, b in arr]


class KmerHash:
    """hash_kmers(seq, k) (sparse.rs:350-359): the q-gram index of a sequence.  Only its identity
    (sequence + k) matters to the callers here, the matches are always recomputed by the engine."""

    def __init__(self, seq, k):
        self.seq, self.k = bytes(seq), int(k)


def hash_kmers(seq, k):
    return KmerHash(seq, k)


def find_kmer_matches(seq1, seq2, k):  # sparse.rs:337-348
    a, b = _lib.as_u8(seq1), _lib.as_u8(seq2)
    n = _lib.lib().bg_sparse_find_kmer_matches(a.ctypes.data, len(a), b.ctypes.data, len(b), k, None, 0)
    out = np.zeros((max(n, 1), 2), dtype=np.uint32)
    _lib.lib().bg_sparse_find_kmer_matches(a.ctypes.data, len(a), b.ctypes.data, len(b), k, out.ctypes.data, n)
    return _pairs(out[:n])


def find_kmer_matches_seq2_hashed(seq1, seq2_hash, k):  # sparse.rs:383-402
    assert seq2_hash.k == k
    return find_kmer_matches(seq1, seq2_hash.seq, k)


def _path_call(fn, mm, *args):
    path = np.zeros(2 * len(mm) + 2, dtype=np.uint32)
    n = fn(mm.ctypes.data, len(mm), *args, path.ctypes.data, len(path))
    assert n != _PANIC, "incoming matches must be sorted"  # sparse.rs:79-84, 212-217
    return [int(v) for v in path[:n]]


def sdpkpp_path(matches, k, match_score, gap_open, gap_extend):  # sparse.rs:188-295 (.path)
    assert gap_open <= 0 and gap_extend <= 0, "gap parameters cannot be positive"
    return _path_call(_lib.lib().bg_sparse_sdpkpp, _xy(matches), k, match_score, gap_open, gap_extend)


def lcskpp(matches, k):  # sparse.rs:67-143 -> (path, score)
    mm = _xy(matches)
    path = np.zeros(len(mm) + 1, dtype=np.uint32)
    score = C.c_uint32(0)
    n = _lib.lib().bg_sparse_lcskpp(mm.ctypes.data, len(mm), k, path.ctypes.data, len(path), C.byref(score))
    assert n != _PANIC, "incoming matches must be sorted."
    return [int(v) for v in path[:n]], int(score.value)


def sdpkpp_union_lcskpp_path(matches, k, match_score, gap_open, gap_extend):  # sparse.rs:297-329
    return _path_call(_lib.lib().bg_sparse_sdpkpp_union_lcskpp_path, _xy(matches), k, match_score, gap_open, gap_extend)


def expand_kmer_matches(seq1, seq2, k, sorted_matches, allowed_mismatches):  # sparse.rs:404-500
    a, b = _lib.as_u8(seq1), _lib.as_u8(seq2)
    mm = _xy(sorted_matches)
    cap = len(mm) + len(a) + len(b) + 8
    while True:
        out = np.zeros((cap, 2), dtype=np.uint32)
        n = _lib.lib().bg_sparse_expand_kmer_matches(a.ctypes.data, len(a), b.ctypes.data, len(b), k, mm.ctypes.data, len(mm),
                                                     allowed_mismatches, out.ctypes.data, cap)
        assert n != _PANIC, "incoming matches must be sorted"
        if n <= cap:
            return _pairs(out[:n])
        cap = n
