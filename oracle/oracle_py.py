"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under rust-bio_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

MIN_SCORE = -858993459
MODES = {"custom": 0, "global": 1, "semiglobal": 2, "local": 3}
OP_NAMES = ["M", "S", "D", "I", "X", "Y"]
TAGS = ["complete", "partial", "absent", "panic"]


class Scoring(C.Structure):
    _fields_ = [("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("xclip_prefix", C.c_int32), ("xclip_suffix", C.c_int32),
                ("yclip_prefix", C.c_int32), ("yclip_suffix", C.c_int32),
                ("match_score", C.c_int32), ("mismatch_score", C.c_int32),
                ("match_scores_some", C.c_int32),
                ("matrix", C.POINTER(C.c_int32))]


class AlignmentRec(C.Structure):
    _fields_ = [("score", C.c_int32), ("ystart", C.c_uint64), ("xstart", C.c_uint64),
                ("yend", C.c_uint64), ("xend", C.c_uint64), ("ylen", C.c_uint64),
                ("xlen", C.c_uint64), ("n_ops", C.c_uint64), ("mode", C.c_int32)]


ALN_DTYPE = np.dtype([("score", "<i4"), ("_p0", "<i4"), ("ystart", "<u8"), ("xstart", "<u8"),
                      ("yend", "<u8"), ("xend", "<u8"), ("ylen", "<u8"), ("xlen", "<u8"),
                      ("n_ops", "<u8"), ("mode", "<i4"), ("_p1", "<i4")])
assert ALN_DTYPE.itemsize == C.sizeof(AlignmentRec)


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.orc_align.restype = C.c_int
        L.orc_align.argtypes = [C.POINTER(Scoring), C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                C.c_uint64, C.POINTER(AlignmentRec), C.c_void_p, C.c_uint64]
        L.orc_align_batch.restype = C.c_int
        L.orc_align_batch.argtypes = [C.POINTER(Scoring), C.c_int, C.c_uint64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_int]
        L.orc_test_lf_hook.restype = None
        L.orc_test_lf_hook.argtypes = [C.c_int]
        L.orc_test_lf_lx_reads.restype = C.c_uint64
        L.orc_test_lf_lx_reads.argtypes = []
        L.orc_suffix_array.restype = C.c_int
        L.orc_suffix_array.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_bwt.restype = None
        L.orc_bwt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_less.restype = C.c_uint64
        L.orc_less.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_occ_new.restype = C.c_void_p
        L.orc_occ_new.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]
        L.orc_occ_free.restype = None
        L.orc_occ_free.argtypes = [C.c_void_p]
        L.orc_occ_row.restype = C.POINTER(C.c_uint64)
        L.orc_occ_row.argtypes = [C.c_void_p, C.c_uint32, u64p]
        L.orc_occ_get.restype = C.c_int
        L.orc_occ_get.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint8, u64p]
        L.orc_backward_search.restype = C.c_int
        L.orc_backward_search.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                          C.c_void_p, C.c_void_p, C.c_uint64, u64p, u64p, u64p]
        L.orc_backward_search_batch.restype = None
        L.orc_backward_search_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int]
        L.orc_intervals_by_scan.restype = None
        L.orc_intervals_by_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        for name, res, args in [
            ("orc_banded_align", C.c_int,
             [C.POINTER(Scoring), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
              C.c_void_p, C.c_uint64, C.POINTER(AlignmentRec), C.c_void_p, C.c_uint64, u64p]),
            ("orc_banded_align_batch", C.c_int,
             [C.POINTER(Scoring), C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p,
              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
              C.c_int]),
            ("orc_band_create", C.c_uint64,
             [C.POINTER(Scoring), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
              C.c_uint64, C.c_void_p, C.c_void_p]),
            ("orc_find_kmer_matches", C.c_uint64,
             [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64]),
            ("orc_sdpkpp", C.c_uint64,
             [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p,
              C.c_uint64, u32p]),
            ("orc_lcskpp", C.c_uint64,
             [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, u32p]),
            ("orc_sdpkpp_union_lcskpp_path", C.c_uint64,
             [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64]),
            ("orc_expand_kmer_matches", C.c_uint64,
             [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32,
              C.c_void_p, C.c_uint64]),
            ("orc_banded_align_with", C.c_int,
             [C.POINTER(Scoring), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
              C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(AlignmentRec),
              C.c_void_p, C.c_uint64, u64p, C.c_void_p, C.c_void_p]),
            ("orc_fmd_check", C.c_int, [C.c_void_p, C.c_uint64]),
            ("orc_fmd_ext_calls", C.c_uint64, [C.c_int]),
            ("orc_fmd_smems", C.c_int64,
             [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
              C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]),
            ("orc_fmd_interval", C.c_int,
             [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_uint8, C.c_void_p]),
            ("orc_sa_sample", C.c_void_p, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
            ("orc_sa_sample_free", None, [C.c_void_p]),
            ("orc_sa_sample_counts", C.c_uint64, [C.c_void_p, u64p]),
            ("orc_sa_sample_export", None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
            ("orc_sampled_sa_get", C.c_int,
             [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]),
            ("orc_fastq_parse", C.c_int,
             [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, u64p, C.POINTER(C.c_int32), u64p]),
            ("orc_seed_extend_batch", C.c_int,
             [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
              C.POINTER(Scoring), C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
              C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
            ("orc_seed_extend_batch_sa32", C.c_int,
             [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
              C.POINTER(Scoring), C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
              C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
            ("orc_pretty", C.c_int64, [C.POINTER(AlignmentRec), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                       C.c_uint64, C.c_void_p, C.c_uint64]),
            ("orc_cigar", C.c_int64, [C.POINTER(AlignmentRec), C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]),
        ]:
            if hasattr(L, name):
                f = getattr(L, name)
                f.restype = res
                f.argtypes = args
        _lib = L
    return _lib


def _buf(b):
    """bytes / bytearray / ndarray -> contiguous uint8 ndarray"""
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def make_scoring(gap_open, gap_extend, match=0, mismatch=0, xclip_prefix=MIN_SCORE,
                 xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE,
                 matrix=None, match_scores_some=None):
    """Returns (Scoring, keepalive). matrix: int32[256,256] tabulated match_fn or None."""
    sc = Scoring()
    sc.gap_open, sc.gap_extend = gap_open, gap_extend
    sc.xclip_prefix, sc.xclip_suffix = xclip_prefix, xclip_suffix
    sc.yclip_prefix, sc.yclip_suffix = yclip_prefix, yclip_suffix
    sc.match_score, sc.mismatch_score = match, mismatch
    keep = None
    if matrix is not None:
        keep = np.ascontiguousarray(matrix, dtype=np.int32).reshape(256 * 256)
        sc.matrix = keep.ctypes.data_as(C.POINTER(C.c_int32))
        sc.match_scores_some = 0 if match_scores_some is None else int(match_scores_some)
    else:
        sc.matrix = None
        sc.match_scores_some = 1 if match_scores_some is None else int(match_scores_some)
    return sc, keep


def decode_ops(ops_u64):
    out = []
    for v in ops_u64:
        kind, ln = int(v) & 0xFF, int(v) >> 8
        out.append(OP_NAMES[kind] + (str(ln) if kind >= 4 else ""))
    return out


def _rec_to_dict(rec, ops):
    return {"score": int(rec.score), "xstart": int(rec.xstart), "xend": int(rec.xend),
            "ystart": int(rec.ystart), "yend": int(rec.yend), "xlen": int(rec.xlen),
            "ylen": int(rec.ylen), "mode": int(rec.mode), "ops": ops}


def align(scoring, mode, x, y):
    """One pair through the oracle Aligner; returns dict with ops as token list."""
    sc = scoring[0] if isinstance(scoring, tuple) else scoring
    xb, yb = _buf(x), _buf(y)
    cap = len(xb) + len(yb) + 8
    ops = np.zeros(cap, dtype=np.uint64)
    rec = AlignmentRec()
    rc = lib().orc_align(C.byref(sc), MODES[mode] if isinstance(mode, str) else mode,
                         xb.ctypes.data, len(xb), yb.ctypes.data, len(yb), C.byref(rec),
                         ops.ctypes.data, cap)
    if rc:
        raise RuntimeError(f"oracle align failed rc={rc}")
    return _rec_to_dict(rec, decode_ops(ops[:rec.n_ops]))


def align_batch(scoring, mode, x, x_off, y, y_off, threads=1, want_ops=True):
    sc = scoring[0] if isinstance(scoring, tuple) else scoring
    xb, yb = _buf(x), _buf(y)
    x_off = np.ascontiguousarray(x_off, dtype=np.uint64)
    y_off = np.ascontiguousarray(y_off, dtype=np.uint64)
    n = len(x_off) - 1
    out = np.zeros(n, dtype=ALN_DTYPE)
    stride = 0
    ops = None
    if want_ops and n:
        stride = int((np.diff(x_off) + np.diff(y_off)).max()) + 8
        ops = np.zeros(n * stride, dtype=np.uint64)
    rc = lib().orc_align_batch(C.byref(sc), MODES[mode] if isinstance(mode, str) else mode, n,
                               xb.ctypes.data, x_off.ctypes.data, yb.ctypes.data,
                               y_off.ctypes.data, out.ctypes.data,
                               ops.ctypes.data if ops is not None else None, stride, threads)
    if rc:
        raise RuntimeError(f"oracle align_batch failed rc={rc}")
    return out, ops, stride


def lf_hook(on):
    """test hook of pairwise.cpp: Aligner::custom without the x-suffix-clip fold of the columns before n (what the engine's
    LF kernel leaves out); a traceback that asks for such an Lx[j] makes the call fail and is counted"""
    lib().orc_test_lf_hook(1 if on else 0)


def lf_lx_reads():
    return int(lib().orc_test_lf_lx_reads())


def banded_align(scoring, mode, k, w, x, y):
    sc = scoring[0] if isinstance(scoring, tuple) else scoring
    xb, yb = _buf(x), _buf(y)
    cap = len(xb) + len(yb) + 8
    ops = np.zeros(cap, dtype=np.uint64)
    rec = AlignmentRec()
    cells = C.c_uint64(0)
    rc = lib().orc_banded_align(C.byref(sc), MODES[mode] if isinstance(mode, str) else mode, k, w,
                                xb.ctypes.data, len(xb), yb.ctypes.data, len(yb), C.byref(rec),
                                ops.ctypes.data, cap, C.byref(cells))
    if rc:
        raise RuntimeError(f"oracle banded_align failed rc={rc}")
    d = _rec_to_dict(rec, decode_ops(ops[:rec.n_ops]))
    d["band_cells"] = int(cells.value)
    return d


def banded_align_batch(scoring, mode, k, w, x, x_off, y, y_off, threads=1, want_ops=True):
    sc = scoring[0] if isinstance(scoring, tuple) else scoring
    xb, yb = _buf(x), _buf(y)
    x_off = np.ascontiguousarray(x_off, dtype=np.uint64)
    y_off = np.ascontiguousarray(y_off, dtype=np.uint64)
    n = len(x_off) - 1
    out = np.zeros(n, dtype=ALN_DTYPE)
    cells = np.zeros(n, dtype=np.uint64)
    stride = 0
    ops = None
    if want_ops and n:
        stride = int((np.diff(x_off) + np.diff(y_off)).max()) + 8
        ops = np.zeros(n * stride, dtype=np.uint64)
    rc = lib().orc_banded_align_batch(C.byref(sc), MODES[mode] if isinstance(mode, str) else mode,
                                      k, w, n, xb.ctypes.data, x_off.ctypes.data, yb.ctypes.data,
                                      y_off.ctypes.data, out.ctypes.data,
                                      ops.ctypes.data if ops is not None else None, stride,
                                      cells.ctypes.data, threads)
    if rc:
        raise RuntimeError(f"oracle banded_align_batch failed rc={rc}")
    return out, ops, stride, cells


def band_create(scoring, k, w, x, y):
    sc = scoring[0] if isinstance(scoring, tuple) else scoring
    xb, yb = _buf(x), _buf(y)
    n = len(yb)
    start = np.zeros(n + 1, dtype=np.uint32)
    end = np.zeros(n + 1, dtype=np.uint32)
    cells = lib().orc_band_create(C.byref(sc), k, w, xb.ctypes.data, len(xb), yb.ctypes.data, n,
                                  start.ctypes.data, end.ctypes.data)
    return start, end, int(cells)


def find_kmer_matches(x, y, k):
    xb, yb = _buf(x), _buf(y)
    cnt = lib().orc_find_kmer_matches(xb.ctypes.data, len(xb), yb.ctypes.data, len(yb), k, None, 0)
    out = np.zeros((max(cnt, 1), 2), dtype=np.uint32)
    lib().orc_find_kmer_matches(xb.ctypes.data, len(xb), yb.ctypes.data, len(yb), k,
                                out.ctypes.data, cnt)
    return out[:cnt]


def sdpkpp(matches, k, match_score, gap_open, gap_extend):
    mm = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
    path = np.zeros(max(len(mm), 1), dtype=np.uint32)
    score = C.c_uint32(0)
    cnt = lib().orc_sdpkpp(mm.ctypes.data, len(mm), k, match_score, gap_open, gap_extend,
                           path.ctypes.data, len(path), C.byref(score))
    return path[:cnt].tolist(), int(score.value)


def lcskpp(matches, k):
    mm = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
    path = np.zeros(max(len(mm), 1), dtype=np.uint32)
    score = C.c_uint32(0)
    cnt = lib().orc_lcskpp(mm.ctypes.data, len(mm), k, path.ctypes.data, len(path),
                           C.byref(score))
    return path[:cnt].tolist(), int(score.value)


# ---------------------------------------------------------------- FM index
def suffix_array(text):
    t = _buf(text)
    sa = np.zeros(len(t), dtype=np.uint64)
    if lib().orc_suffix_array(t.ctypes.data, len(t), sa.ctypes.data):
        raise ValueError("Expecting extra sentinel symbol being lexicographically smallest at "
                         "the end of the text.")
    return sa


def bwt(text, sa):
    t = _buf(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint64)
    out = np.zeros(len(t), dtype=np.uint8)
    lib().orc_bwt(t.ctypes.data, sa.ctypes.data, len(t), out.ctypes.data)
    return out


def less(bwt_arr, alphabet):
    b, a = _buf(bwt_arr), _buf(alphabet)
    m = lib().orc_less(b.ctypes.data, len(b), a.ctypes.data, len(a), None)
    out = np.zeros(m, dtype=np.uint64)
    if lib().orc_less(b.ctypes.data, len(b), a.ctypes.data, len(a), out.ctypes.data) == 0:
        raise IndexError("bwt symbol beyond max_symbol+1")
    return out


class Occ:
    def __init__(self, bwt_arr, k, alphabet):
        self.bwt = _buf(bwt_arr)
        a = _buf(alphabet)
        self.k = k
        self.h = lib().orc_occ_new(self.bwt.ctypes.data, len(self.bwt), k, a.ctypes.data, len(a))
        if not self.h:
            raise IndexError("bwt symbol beyond alphabet max_symbol")

    def row(self, a):
        ln = C.c_uint64(0)
        p = lib().orc_occ_row(self.h, a, C.byref(ln))
        return [int(p[i]) for i in range(ln.value)]

    def get(self, r, a):
        out = C.c_uint64(0)
        if lib().orc_occ_get(self.h, self.bwt.ctypes.data, len(self.bwt), r, a, C.byref(out)):
            raise IndexError("index out of bounds (non-alphabet symbol)")
        return int(out.value)

    def __del__(self):
        try:
            lib().orc_occ_free(self.h)
        except Exception:
            pass


def backward_search(bwt_arr, less_arr, occ, pattern):
    b, p = _buf(bwt_arr), _buf(pattern)
    ls = np.ascontiguousarray(less_arr, dtype=np.uint64)
    lo, hi, ml = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    tag = lib().orc_backward_search(b.ctypes.data, len(b), ls.ctypes.data, len(ls), occ.h,
                                    p.ctypes.data, len(p), C.byref(lo), C.byref(hi), C.byref(ml))
    return TAGS[tag], int(lo.value), int(hi.value), int(ml.value)


def backward_search_batch(bwt_arr, less_arr, occ, pat, pat_off, threads=1):
    b, p = _buf(bwt_arr), _buf(pat)
    ls = np.ascontiguousarray(less_arr, dtype=np.uint64)
    off = np.ascontiguousarray(pat_off, dtype=np.uint64)
    n = len(off) - 1
    tag = np.zeros(n, dtype=np.uint8)
    lo = np.zeros(n, dtype=np.uint64)
    hi = np.zeros(n, dtype=np.uint64)
    ml = np.zeros(n, dtype=np.uint64)
    lib().orc_backward_search_batch(b.ctypes.data, len(b), ls.ctypes.data, len(ls), occ.h, n,
                                    p.ctypes.data, off.ctypes.data, tag.ctypes.data,
                                    lo.ctypes.data, hi.ctypes.data, ml.ctypes.data, threads)
    return tag, lo, hi, ml


def intervals_by_scan(text, pat, pat_off, pos_cap=64, threads=1):
    """The interval of every pattern by definition (fm.cpp: orc_intervals_by_scan) — lower = suffixes smaller than the pattern,
    upper = lower + suffixes that start with it — and up to pos_cap occurrence positions each; no suffix array involved."""
    t, p = _buf(text), _buf(pat)
    off = np.ascontiguousarray(pat_off, dtype=np.uint64)
    n = len(off) - 1
    lo, hi, npos = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    pos = np.zeros(max(1, n * pos_cap), dtype=np.uint64)
    lib().orc_intervals_by_scan(t.ctypes.data, len(t), n, p.ctypes.data, off.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                pos.ctypes.data, pos_cap, npos.ctypes.data, threads)
    return lo, hi, [pos[k * pos_cap:k * pos_cap + int(npos[k])] for k in range(n)]


class SampledSuffixArray:
    """RawSuffixArray::sample + SampledSuffixArray::get (suffix_array.rs:86-184)."""

    def __init__(self, sa, text, bwt_arr, less_arr, occ, sampling_rate):
        self.sa = np.ascontiguousarray(sa, dtype=np.uint64)
        self.text = np.frombuffer(_buf(text), dtype=np.uint8)
        self.bwt = np.ascontiguousarray(bwt_arr, dtype=np.uint8)
        self.less = np.ascontiguousarray(less_arr, dtype=np.uint64)
        self.occ = occ
        self.s = int(sampling_rate)
        self.h = lib().orc_sa_sample(self.sa.ctypes.data, len(self.sa), self.text.ctypes.data,
                                     self.bwt.ctypes.data, self.s)

    def arrays(self):
        ne = C.c_uint64(0)
        ns = lib().orc_sa_sample_counts(self.h, C.byref(ne))
        sample = np.zeros(ns, dtype=np.uint64)
        erow = np.zeros(ne.value, dtype=np.uint64)
        epos = np.zeros(ne.value, dtype=np.uint64)
        lib().orc_sa_sample_export(self.h, sample.ctypes.data, erow.ctypes.data, epos.ctypes.data)
        return sample, erow, epos

    def get(self, index):
        out = C.c_uint64(0)
        rc = lib().orc_sampled_sa_get(self.h, self.bwt.ctypes.data, len(self.bwt), self.less.ctypes.data,
                                      len(self.less), self.occ.h, int(index), C.byref(out))
        if rc == -1:
            return None
        if rc != 0:
            raise IndexError("the reference panics here (byte outside the alphabet)")
        return int(out.value)

    def __del__(self):
        try:
            lib().orc_sa_sample_free(self.h)
        except Exception:
            pass


def interval_occ(lower, upper, sa):
    """Interval::occ (fmindex.rs:75-79); `sa` is a raw array or a SampledSuffixArray."""
    if isinstance(sa, SampledSuffixArray):
        return [sa.get(i) for i in range(lower, upper)]
    return [int(sa[i]) for i in range(lower, upper)]


def _matches_xy(matches):
    return np.ascontiguousarray(np.asarray(matches, dtype=np.uint32).reshape(-1, 2))


def sdpkpp_union_lcskpp_path(matches, k, match_score, gap_open, gap_extend):
    """sparse.rs:297-329"""
    mm = _matches_xy(matches)
    path = np.zeros(2 * len(mm) + 1, dtype=np.uint32)
    n = lib().orc_sdpkpp_union_lcskpp_path(mm.ctypes.data, len(mm), k, match_score, gap_open, gap_extend,
                                           path.ctypes.data, len(path))
    return [int(v) for v in path[:n]]


def expand_kmer_matches(x, y, k, matches, allowed_mismatches):
    """sparse.rs:404-500"""
    xb, yb = _buf(x), _buf(y)
    mm = _matches_xy(matches)
    cap = len(mm) + len(xb) + len(yb) + 8
    while True:
        out = np.zeros((cap, 2), dtype=np.uint32)
        n = lib().orc_expand_kmer_matches(xb.ctypes.data, len(xb), yb.ctypes.data, len(yb), k, mm.ctypes.data, len(mm),
                                          allowed_mismatches, out.ctypes.data, cap)
        if n <= cap:
            return [(int(a), int(b)) for a, b in out[:n]]
        cap = n


def banded_align_with(scoring, mode, k, w, x, y, matches, path=None, expanded=False, allowed_mismatches=None,
                      use_lcskpp_union=False, want_band=False):
    """custom_with_matches / custom_with_match_path / custom_with_expanded_matches (banded.rs:313-401), in any
    mode.  Returns the alignment dict (+ 'band_cells', and 'band' = (start, end) when want_band)."""
    sc = scoring[0] if isinstance(scoring, tuple) else scoring
    xb, yb = _buf(x), _buf(y)
    mm = _matches_xy(matches)
    variant = 3 if expanded else (2 if path is not None else 1)
    pp = np.ascontiguousarray(path if path is not None else [], dtype=np.uint32)
    rec = AlignmentRec()
    cap = len(xb) + len(yb) + 8
    ops = np.zeros(cap, dtype=np.uint64)
    cells = C.c_uint64(0)
    bs = np.zeros(len(yb) + 1, dtype=np.uint32)
    be = np.zeros(len(yb) + 1, dtype=np.uint32)
    rc = lib().orc_banded_align_with(C.byref(sc), MODES[mode] if isinstance(mode, str) else mode, k, w, xb.ctypes.data,
                                     len(xb), yb.ctypes.data, len(yb), variant, mm.ctypes.data, len(mm), pp.ctypes.data,
                                     len(pp), -1 if allowed_mismatches is None else int(allowed_mismatches),
                                     1 if use_lcskpp_union else 0, C.byref(rec), ops.ctypes.data, cap, C.byref(cells),
                                     bs.ctypes.data, be.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle banded_align_with failed rc={rc}")
    d = _rec_to_dict(rec, decode_ops(ops[:rec.n_ops]))
    d["band_cells"] = int(cells.value)
    if want_band:
        d["band"] = (bs, be)
    return d


class FMDIndex:
    """FMDIndex (fmindex.rs:250-576) over the oracle's FM index parts; intervals are tuples
    (lower, lower_rev, size, match_size); smems are (interval, pattern position, length)."""

    def __init__(self, bwt_arr, less_arr, occ):
        self.bwt = np.ascontiguousarray(np.frombuffer(_buf(bwt_arr), dtype=np.uint8))
        self.less = np.ascontiguousarray(less_arr, dtype=np.uint64)
        self.occ = occ
        assert lib().orc_fmd_check(self.bwt.ctypes.data, len(self.bwt)), \
            "Expecting BWT over the DNA alphabet (including N) with the sentinel $."

    def _interval(self, op, iv=None, a=0):
        out = np.zeros(4, dtype=np.uint64)
        ivp = np.ascontiguousarray(iv, dtype=np.uint64) if iv is not None else None
        rc = lib().orc_fmd_interval(self.bwt.ctypes.data, len(self.bwt), self.less.ctypes.data, len(self.less),
                                    self.occ.h, op, ivp.ctypes.data if ivp is not None else None, a, out.ctypes.data)
        if rc:
            raise IndexError("the reference panics here")
        return tuple(int(v) for v in out)

    def init_interval(self): return self._interval(0)
    def init_interval_with(self, a): return self._interval(1, None, a)
    def backward_ext(self, iv, a): return self._interval(2, iv, a)
    def forward_ext(self, iv, a): return self._interval(3, iv, a)

    def _smems(self, pattern, i, l, all_):
        pb = np.frombuffer(_buf(pattern), dtype=np.uint8)
        cap = 4 * len(pb) + 8
        out = np.zeros(6 * cap, dtype=np.uint64)
        n = lib().orc_fmd_smems(self.bwt.ctypes.data, len(self.bwt), self.less.ctypes.data, len(self.less), self.occ.h,
                                pb.ctypes.data, len(pb), i, l, all_, out.ctypes.data, cap)
        if n < 0:
            raise IndexError("the reference panics here")
        assert n <= cap
        r = out[:6 * n].reshape(n, 6)
        return [((int(v[0]), int(v[1]), int(v[2]), int(v[3])), int(v[4]), int(v[5])) for v in r]

    @staticmethod
    def ext_calls(reset=False):
        """backward_ext calls the restatement has made since the last reset (forward_ext is one of them): bench.py's SMEM leg"""
        return int(lib().orc_fmd_ext_calls(1 if reset else 0))

    def smems(self, pattern, i, l): return self._smems(pattern, i, l, 0)
    def all_smems(self, pattern, l): return self._smems(pattern, 0, l, 1)


# ---- bio::io::fastq::Reader (io/fastq.rs:266-303) and bio-types Alignment::cigar -------------------------
FASTQ_STATUS = ["ok", "MissingAt", "IncompleteRecord", "Io"]
FASTQ_CHECK = ["ok", "EmptyId", "NonAsciiSequence", "InvalidSequence", "NonAsciiQualities", "UnequalLength"]
FQREC_DTYPE = np.dtype([("id_off", "<u8"), ("id_len", "<u8"), ("desc_off", "<u8"), ("desc_len", "<u8"),
                        ("seq_off", "<u8"), ("seq_len", "<u8"), ("qual_off", "<u8"), ("qual_len", "<u8"),
                        ("has_desc", "<i4"), ("check", "<i4")])


def fastq_parse(text):
    """-> (records: list of dict(id, desc, seq, qual, check), status name, err_pos); records read before the
    first ReadError, like collecting `Reader::records()` up to the first Err."""
    t = _buf(text)
    cap = len(t) // 4 + 2
    recs = np.zeros(cap, dtype=FQREC_DTYPE)
    seq = np.zeros(max(1, len(t)), dtype=np.uint8)
    qual = np.zeros(max(1, len(t)), dtype=np.uint8)
    n, st, ep = C.c_uint64(0), C.c_int32(0), C.c_uint64(0)
    lib().orc_fastq_parse(t.ctypes.data, len(t), recs.ctypes.data, cap, seq.ctypes.data, qual.ctypes.data,
                          C.byref(n), C.byref(st), C.byref(ep))
    tb = t.tobytes()
    out = []
    for r in recs[:n.value]:
        out.append({"id": tb[int(r["id_off"]):int(r["id_off"] + r["id_len"])],
                    "desc": tb[int(r["desc_off"]):int(r["desc_off"] + r["desc_len"])] if r["has_desc"] else None,
                    "seq": seq[int(r["seq_off"]):int(r["seq_off"] + r["seq_len"])].tobytes(),
                    "qual": qual[int(r["qual_off"]):int(r["qual_off"] + r["qual_len"])].tobytes(),
                    "check": FASTQ_CHECK[int(r["check"])]})
    return out, FASTQ_STATUS[st.value], int(ep.value)


def cigar(aln, ops_u64, hard_clip):
    """aln: dict with xstart/xend/xlen/mode (ints); ops: uint64 kind | len << 8.  None for Custom (panic)."""
    rec = AlignmentRec()
    rec.xstart, rec.xend, rec.xlen, rec.mode = int(aln["xstart"]), int(aln["xend"]), int(aln["xlen"]), int(aln["mode"])
    ops = np.ascontiguousarray(ops_u64, dtype=np.uint64)
    rec.n_ops = len(ops)
    buf = np.zeros(24 * (len(ops) + 4), dtype=np.uint8)
    n = lib().orc_cigar(C.byref(rec), ops.ctypes.data, 1 if hard_clip else 0, buf.ctypes.data, len(buf))
    if n == -2:
        return None
    assert n >= 0
    return buf[:n].tobytes().decode()


def fastq_parse_raw(text):
    """timed flavour: -> ((recs, seq bytes, qual bytes), status code, err_pos) without per-record Python objects"""
    t = _buf(text)
    cap = len(t) // 4 + 2
    recs = np.zeros(cap, dtype=FQREC_DTYPE)
    seq = np.zeros(max(1, len(t)), dtype=np.uint8)
    qual = np.zeros(max(1, len(t)), dtype=np.uint8)
    n, st, ep = C.c_uint64(0), C.c_int32(0), C.c_uint64(0)
    lib().orc_fastq_parse(t.ctypes.data, len(t), recs.ctypes.data, cap, seq.ctypes.data, qual.ctypes.data,
                          C.byref(n), C.byref(st), C.byref(ep))
    k = int(n.value)
    sl = int(recs["seq_off"][k - 1] + recs["seq_len"][k - 1]) if k else 0
    ql = int(recs["qual_off"][k - 1] + recs["qual_len"][k - 1]) if k else 0
    return (recs[:k], seq[:sl].tobytes(), qual[:ql].tobytes()), st.value, int(ep.value)


SEED_HIT_DTYPE = np.dtype([("aln", ALN_DTYPE), ("window_start", "<u8"), ("ref_start", "<u8"), ("ref_end", "<u8"),
                           ("n_candidates", "<u4"), ("n_seed_hits", "<u4")])


def seed_extend_batch(bwt_arr, less_arr, occ, sa, text, n_text, scoring, reads, read_off, seed_len=20, stride=10,
                      max_occ=16, pad=25, threads=1, want_ops=True):
    """The seed-and-extend composition of oracle/pipeline.cpp (backward_search -> Interval::occ over the raw
    suffix array -> Aligner::semiglobal on the candidate windows -> best hit).  Returns (hits, ops, stride)."""
    sc = scoring[0] if isinstance(scoring, tuple) else scoring
    b, tx, rd = _buf(bwt_arr), _buf(text), _buf(reads)
    ls = np.ascontiguousarray(less_arr, dtype=np.uint64)
    sa32 = getattr(sa, "dtype", None) == np.uint32  # a suffix array downloaded from the device: not widened
    sa = np.ascontiguousarray(sa, dtype=np.uint32 if sa32 else np.uint64)
    off = np.ascontiguousarray(read_off, dtype=np.uint64)
    n = len(off) - 1
    out = np.zeros(n, dtype=SEED_HIT_DTYPE)
    assert SEED_HIT_DTYPE.itemsize == ALN_DTYPE.itemsize + 32
    stride_ops = 0
    ops = None
    if want_ops and n:
        stride_ops = 2 * int(np.diff(off).max()) + 2 * pad + 8
        ops = np.zeros(n * stride_ops, dtype=np.uint64)
    fn = lib().orc_seed_extend_batch_sa32 if sa32 else lib().orc_seed_extend_batch
    rc = fn(b.ctypes.data, len(b), ls.ctypes.data, len(ls), occ.h, sa.ctypes.data, tx.ctypes.data, n_text, C.byref(sc), n,
            rd.ctypes.data, off.ctypes.data, seed_len, stride, max_occ, pad, out.ctypes.data,
            ops.ctypes.data if ops is not None else None, stride_ops, threads)
    if rc:
        raise RuntimeError(f"oracle seed_extend_batch failed rc={rc}")
    return out, ops, stride_ops


def pretty(aln, ops_u64, x, y, ncol):
    """bio-types `Alignment::pretty(x, y, ncol)`; aln: dict with xstart, ystart, mode (+ the ops as kind | len << 8)."""
    rec = AlignmentRec()
    for f in ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen"):
        setattr(rec, f, int(aln.get(f, 0)))
    rec.mode = MODES[aln["mode"]] if isinstance(aln["mode"], str) else int(aln["mode"])
    ops = np.ascontiguousarray(ops_u64, dtype=np.uint64)
    rec.n_ops = len(ops)
    xb, yb = _buf(x), _buf(y)
    cap = 3 * (len(xb) + len(yb)) * 3 + 5 * (len(xb) + len(yb) + 1) + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_pretty(C.byref(rec), ops.ctypes.data, xb.ctypes.data, len(xb), yb.ctypes.data, len(yb), ncol, out.ctypes.data, cap)
    if n == -2:
        raise AssertionError("the crate panics here")
    assert n >= 0
    return out[:n].tobytes().decode()
