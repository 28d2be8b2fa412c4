"""pairwise::{Scoring, MatchParams, Aligner} — host mirror of
/root/reference/src/alignment/pairwise/mod.rs:186-1016 on top of the C ABI (bg_align_batch).

The single-pair methods (`custom`, `global_`, `semiglobal`, `local`) delegate to the batched
ones with a batch of one; the batched ones are what the GPU engine is built for."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib

MIN_SCORE = _lib.MIN_SCORE  # mod.rs:174
MODE_CUSTOM, MODE_GLOBAL, MODE_SEMIGLOBAL, MODE_LOCAL = 0, 1, 2, 3
MODE_NAMES = ["Custom", "Global", "Semiglobal", "Local"]
OP_TOKENS = ["M", "S", "D", "I", "X", "Y"]  # Match Subst Del Ins Xclip(n) Yclip(n)


@dataclass(frozen=True)
class MatchParams:  # mod.rs:186-217
    match_score: int
    mismatch_score: int

    def __post_init__(self):
        assert self.match_score >= 0, "match_score can't be negative"
        assert self.mismatch_score <= 0, "mismatch_score can't be positive"

    def score(self, a, b):
        return self.match_score if a == b else self.mismatch_score


def tabulate(match_fn, undefined=None):
    """A closure cannot cross the FFI: tabulate F over all byte pairs.  Byte pairs the function
    is not defined on (it raises KeyError / IndexError there, e.g. blosum62 on digits — the reference panics when
    such a pair is actually scored) are tabulated as 0 and flagged in `undefined` (bool[256, 256]) so that the
    aligner can refuse a batch that contains them; any other exception is a bug in the closure and propagates."""
    if isinstance(match_fn, np.ndarray):
        m = np.ascontiguousarray(match_fn, dtype=np.int32)
        assert m.shape == (256, 256)
        return m
    m = np.zeros((256, 256), dtype=np.int32)
    for a in range(256):
        for b in range(256):
            try:
                m[a, b] = match_fn(a, b)
            except (KeyError, IndexError):
                if undefined is not None:
                    undefined[a, b] = True
    return m


class Scoring:
    """Scoring<F> (mod.rs:238-429)."""

    def __init__(self, gap_open, gap_extend, match_fn, match_scores=None, xclip_prefix=MIN_SCORE,
                 xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE, yclip_suffix=MIN_SCORE):
        self.gap_open, self.gap_extend = gap_open, gap_extend
        self.match_fn = match_fn
        self.match_scores = match_scores
        self.xclip_prefix, self.xclip_suffix = xclip_prefix, xclip_suffix
        self.yclip_prefix, self.yclip_suffix = yclip_prefix, yclip_suffix
        self._matrix = None
        self._undefined = None  # byte pairs match_fn is not defined on (see tabulate)

    @staticmethod
    def from_scores(gap_open, gap_extend, match_score, mismatch_score):  # mod.rs:259-278
        assert gap_open <= 0, "gap_open can't be positive"
        assert gap_extend <= 0, "gap_extend can't be positive"
        return Scoring(gap_open, gap_extend, MatchParams(match_score, mismatch_score),
                       (match_score, mismatch_score))

    @staticmethod
    def new(gap_open, gap_extend, match_fn):  # mod.rs:291-305
        assert gap_open <= 0, "gap_open can't be positive"
        assert gap_extend <= 0, "gap_extend can't be positive"
        return Scoring(gap_open, gap_extend, match_fn, None)

    def _clip(self, penalty, *names):
        assert penalty <= 0, "Clipping penalty can't be positive"
        for nm in names:
            setattr(self, nm, penalty)
        return self

    def xclip(self, p): return self._clip(p, "xclip_prefix", "xclip_suffix")
    def yclip(self, p): return self._clip(p, "yclip_prefix", "yclip_suffix")
    def xclip_prefix_(self, p): return self._clip(p, "xclip_prefix")
    def xclip_suffix_(self, p): return self._clip(p, "xclip_suffix")
    def yclip_prefix_(self, p): return self._clip(p, "yclip_prefix")
    def yclip_suffix_(self, p): return self._clip(p, "yclip_suffix")

    def to_c(self):
        sc = _lib.ScoringC()
        sc.gap_open, sc.gap_extend = self.gap_open, self.gap_extend
        sc.xclip_prefix, sc.xclip_suffix = self.xclip_prefix, self.xclip_suffix
        sc.yclip_prefix, sc.yclip_suffix = self.yclip_prefix, self.yclip_suffix
        sc.match_scores_some = 1 if self.match_scores is not None else 0
        if isinstance(self.match_fn, MatchParams):
            sc.match_score, sc.mismatch_score = self.match_fn.match_score, self.match_fn.mismatch_score
            sc.matrix = None
        else:
            if self._matrix is None:
                self._undefined = np.zeros((256, 256), dtype=bool)
                self._matrix = tabulate(self.match_fn, self._undefined).reshape(-1)
                if not self._undefined.any():
                    self._undefined = None
            if self.match_scores is not None:
                sc.match_score, sc.mismatch_score = self.match_scores
            sc.matrix = self._matrix.ctypes.data_as(C.POINTER(C.c_int32))
        return sc


@dataclass
class Alignment:
    """bio_types::alignment::Alignment as constructed at mod.rs:911-921; operations are tokens
    M S D I Xn Yn."""
    score: int
    ystart: int
    xstart: int
    yend: int
    xend: int
    ylen: int
    xlen: int
    operations: list = field(default_factory=list)
    mode: str = "Custom"

    def cigar(self, hard_clip):
        """bio_types `Alignment::cigar(hard_clip)` (bio-types 1.0; parity unpinned, see include/biogpu.h)."""
        rec = np.zeros(1, dtype=_lib.ALN_DTYPE)
        rec["xstart"], rec["xend"], rec["xlen"] = self.xstart, self.xend, self.xlen
        rec["mode"] = MODE_NAMES.index(self.mode)
        ops = np.array([OP_TOKENS.index(o[0]) for o in self.operations], dtype=np.uint8)
        rec["n_ops"] = len(ops)
        return cigar_batch(rec, ops, hard_clip)[0]


    def pretty(self, x, y, ncol):
        """bio_types `Alignment::pretty(x, y, ncol)` (parity unpinned, see include/biogpu.h)."""
        rec = np.zeros(1, dtype=_lib.ALN_DTYPE)
        for f in ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen"):
            rec[f] = getattr(self, f)
        rec["mode"] = MODE_NAMES.index(self.mode)
        ops, clips = [], []
        for o in self.operations:
            ops.append(OP_TOKENS.index(o[0]))
            if o[0] in "XY":
                clips.append(int(o[1:]))
        assert len(clips) <= 4
        rec["n_ops"], rec["n_clips"] = len(ops), len(clips)
        rec["clip_len"][0, :len(clips)] = clips
        return pretty_batch(rec, np.array(ops, dtype=np.uint8), [bytes(x)], [bytes(y)], ncol)[0]


def pretty_batch(recs, ops_buf, xs, ys, ncol, ctx=None):
    """`Alignment::pretty` for a batch of bg_alignment_t records + the sequences they were computed from."""
    ctx = ctx or _lib.default_context()
    recs = np.ascontiguousarray(recs, dtype=_lib.ALN_DTYPE)
    ops_buf = np.ascontiguousarray(ops_buf if ops_buf is not None else np.zeros(0, np.uint8), dtype=np.uint8)
    x, xo = _lib.concat(xs)
    y, yo = _lib.concat(ys)
    n = len(recs)
    tot = int(xo[-1] + yo[-1])
    cap = 3 * tot + 5 * (tot // max(1, ncol) + n) + 64
    out = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    st = _lib.lib().bg_pretty_batch(ctx.h, n, recs.ctypes.data, ops_buf.ctypes.data, len(ops_buf), x.ctypes.data, xo.ctypes.data,
                                    y.ctypes.data, yo.ctypes.data, ncol, out.ctypes.data, cap, off.ctypes.data)
    assert st != -11, "the crate's row-length assert fires (non-ASCII byte) or the sequences are not the alignment's"
    _lib.check(st, "bg_pretty_batch")
    b = out.tobytes()
    return [b[int(off[p]):int(off[p + 1])].decode() for p in range(n)]


def decode_ops(rec, ops_buf):
    ops = ops_buf[int(rec["ops_off"]):int(rec["ops_off"]) + int(rec["n_ops"])]
    out, c = [], 0
    for o in ops:
        if o >= 4:
            out.append(OP_TOKENS[o] + str(int(rec["clip_len"][c])))
            c += 1
        else:
            out.append(OP_TOKENS[o])
    return out


def cigar_batch(recs, ops_buf, hard_clip, ctx=None):
    """CIGAR strings of a batch of bg_alignment_t records (as returned by align_arrays) on the device.
    AlignmentMode::Custom panics in bio-types: AssertionError here."""
    import ctypes as C
    ctx = ctx or _lib.default_context()
    recs = np.ascontiguousarray(recs, dtype=_lib.ALN_DTYPE)
    ops_buf = np.ascontiguousarray(ops_buf if ops_buf is not None else np.zeros(0, np.uint8), dtype=np.uint8)
    n = len(recs)
    cap = int(recs["n_ops"].astype(np.int64).sum()) * 2 + 24 * n + 64
    out = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    st = _lib.lib().bg_cigar_batch(ctx.h, n, recs.ctypes.data, ops_buf.ctypes.data, len(ops_buf), 1 if hard_clip else 0,
                                   out.ctypes.data, cap, off.ctypes.data)
    assert st != -11, " Cigar fn not supported for custom alignment mode"
    _lib.check(st, "bg_cigar_batch")
    b = out.tobytes()
    return [b[int(off[p]):int(off[p + 1])].decode() for p in range(n)]


def to_alignment(rec, ops_buf):
    return Alignment(int(rec["score"]), int(rec["ystart"]), int(rec["xstart"]), int(rec["yend"]),
                     int(rec["xend"]), int(rec["ylen"]), int(rec["xlen"]),
                     decode_ops(rec, ops_buf) if ops_buf is not None else [],
                     MODE_NAMES[int(rec["mode"])])


def check_defined(scoring, xb, xo, yb, yo):
    """The reference calls `match_fn(x[i-1], y[j-1])` for every cell of a pair (mod.rs:729) and panics where the closure
    does: a byte pair the tabulated closure is not defined on must not MEET inside one alignment.  Checked per pair (a
    byte of some x against a byte of some other pair's y is fine) for the host-buffer batches of this module and of
    banded.py; the device-resident entry points (`align_dev`, `align_packed_dev`) take pointers and stay unchecked — an
    undefined pair scores 0 there."""
    und = scoring._undefined
    if und is None:
        return
    n = len(xo) - 1
    hot = und & np.outer(np.bincount(xb, minlength=256) > 0, np.bincount(yb, minlength=256) > 0)
    if not hot.any():  # no undefined pair has both of its bytes anywhere in the batch
        return
    pair_of_x = np.searchsorted(xo, np.arange(len(xb), dtype=np.uint64), side="right") - 1
    pair_of_y = np.searchsorted(yo, np.arange(len(yb), dtype=np.uint64), side="right") - 1
    for a in np.nonzero(hot.any(axis=1))[0]:
        px = np.zeros(n, dtype=bool)
        px[pair_of_x[xb == a]] = True
        for b in np.nonzero(hot[a])[0]:
            py = np.zeros(n, dtype=bool)
            py[pair_of_y[yb == b]] = True
            both = np.nonzero(px & py)[0]
            if len(both):
                raise KeyError(f"match_fn is not defined on the byte pair ({int(a)}, {int(b)}) that meets in pair {int(both[0])}")


class Aligner:
    """Aligner<F> (mod.rs:472-1016)."""

    def __init__(self, scoring, ctx=None):
        # asserts of with_capacity_and_scoring, mod.rs:554-571
        assert scoring.gap_open <= 0, "gap_open can't be positive"
        assert scoring.gap_extend <= 0, "gap_extend can't be positive"
        assert scoring.xclip_prefix <= 0, "Clipping penalty (x prefix) can't be positive"
        assert scoring.xclip_suffix <= 0, "Clipping penalty (x suffix) can't be positive"
        assert scoring.yclip_prefix <= 0, "Clipping penalty (y prefix) can't be positive"
        assert scoring.yclip_suffix <= 0, "Clipping penalty (y suffix) can't be positive"
        self.scoring = scoring
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    @staticmethod
    def new(gap_open, gap_extend, match_fn, ctx=None):  # mod.rs:495-503
        return Aligner(Scoring.new(gap_open, gap_extend, match_fn), ctx)

    @staticmethod
    def with_capacity(m, n, gap_open, gap_extend, match_fn, ctx=None):  # mod.rs:516-530
        return Aligner(Scoring.new(gap_open, gap_extend, match_fn), ctx)

    @staticmethod
    def with_scoring(scoring, ctx=None):  # mod.rs:537-543
        return Aligner(scoring, ctx)

    @staticmethod
    def with_capacity_and_scoring(m, n, scoring, ctx=None):  # mod.rs:553-583
        return Aligner(scoring, ctx)

    # ---- batched entry points (new; the single-pair API delegates to them)
    def align_arrays(self, mode, x, x_off, y, y_off, want_ops=True, out=None, ops=None):
        """Host-buffer batch through bg_align_batch.  Returns (records, ops_buf); `out` / `ops` may be the
        arrays of an earlier call of the same shape (reused, like a caller's own Vecs)."""
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
        sc = self.scoring.to_c()
        check_defined(self.scoring, xb, xo, yb, yo)
        rc = _lib.lib().bg_align_batch(self.ctx.h, C.byref(sc), mode, n, xb.ctypes.data,
                                       xo.ctypes.data, yb.ctypes.data, yo.ctypes.data,
                                       out.ctypes.data, ops.ctypes.data if want_ops else None,
                                       cap, C.byref(used))
        _lib.check(rc, "bg_align_batch")
        return out, ops

    def align_batch(self, mode, xs, ys):
        x, xo = _lib.concat(xs)
        y, yo = _lib.concat(ys)
        out, ops = self.align_arrays(mode, x, xo, y, yo)
        return [to_alignment(out[p], ops) for p in range(len(xs))]

    def align_dev(self, mode, n_pairs, d_x, d_x_off, d_y, d_y_off, max_xlen, max_ylen, d_out,
                  d_ops, ops_stride, stream=0):
        """Device-resident batch (pointers are ints), asynchronous on `stream`."""
        sc = self.scoring.to_c()
        _lib.check(_lib.lib().bg_align_batch_dev(self.ctx.h, C.byref(sc), mode, n_pairs, d_x,
                                                 d_x_off, d_y, d_y_off, max_xlen, max_ylen, d_out,
                                                 d_ops, ops_stride, stream), "bg_align_batch_dev")

    def align_packed_dev(self, mode, n_pairs, d_xpk, d_x_off, d_ypk, d_y_off, max_xlen, max_ylen, d_out, d_ops, ops_stride,
                         codes=b"ACGT", stream=0):
        """align_dev on 2-bit streams (pack2.py): d_xpk / d_ypk hold 16 symbols per dword, offsets count symbols."""
        sc = self.scoring.to_c()
        cb = (C.c_uint8 * 4)(*bytes(codes))
        _lib.check(_lib.lib().bg_align_batch_packed_dev(self.ctx.h, C.byref(sc), mode, n_pairs, d_xpk, d_x_off, d_ypk, d_y_off,
                                                        cb, max_xlen, max_ylen, d_out, d_ops, ops_stride, stream),
                   "bg_align_batch_packed_dev")

    def custom_batch(self, xs, ys): return self.align_batch(MODE_CUSTOM, xs, ys)
    def global_batch(self, xs, ys): return self.align_batch(MODE_GLOBAL, xs, ys)
    def semiglobal_batch(self, xs, ys): return self.align_batch(MODE_SEMIGLOBAL, xs, ys)
    def local_batch(self, xs, ys): return self.align_batch(MODE_LOCAL, xs, ys)

    # ---- the reference's single-pair API
    def custom(self, x, y): return self.align_batch(MODE_CUSTOM, [x], [y])[0]      # mod.rs:591
    def global_(self, x, y): return self.align_batch(MODE_GLOBAL, [x], [y])[0]     # mod.rs:925
    def semiglobal(self, x, y): return self.align_batch(MODE_SEMIGLOBAL, [x], [y])[0]  # mod.rs:954
    def local(self, x, y): return self.align_batch(MODE_LOCAL, [x], [y])[0]        # mod.rs:986
