"""2-bit sequence streams (csrc/pack2.hip) — BASELINE north_star "packed 2-bit reads", SURVEY.md section 8(f) row 4: the
pack / unpack kernels against a numpy restatement of the layout, and the two consumers against their byte flavours
(which the oracle tests pin): K5 on packed patterns (/root/reference/src/data_structures/fmindex.rs:144-208) and the
aligner on packed pairs (/root/reference/src/alignment/pairwise/mod.rs:591-1015) — identical records and operations."""
import numpy as np
import pytest
import torch

import oracle_py as orc
from rust_bio_amd import _lib, pack2, synth
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.pairwise import Aligner, Scoring
from rust_bio_amd.suffix_array import suffix_array

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 1000, 65_537])
def test_pack_unpack_round_trip_and_layout(n):
    seq = synth.random_dna(max(n, 1), seed=3)[:n]
    d = torch.from_numpy(seq.copy()).to(DEV) if n else torch.empty(0, dtype=torch.uint8, device=DEV)
    pk, bad = pack2.pack_dev(d)
    assert bad == 0
    want = pack2.pack_numpy(seq)
    assert (pk.cpu().numpy().view(np.uint32)[:len(want) - 1] == want[:-1]).all()
    assert bytes(pack2.unpack_dev(pk, n).cpu().numpy()) == bytes(seq)


def test_pack_counts_bytes_outside_the_codes_and_other_code_orders():
    seq = np.frombuffer(b"ACGTNNACGTacgtA$", dtype=np.uint8).copy()
    pk, bad = pack2.pack_dev(torch.from_numpy(seq).to(DEV))
    assert bad == 7  # N N a c g t $
    seq2 = synth.random_dna(999, seed=9)
    pk2, bad2 = pack2.pack_dev(torch.from_numpy(seq2.copy()).to(DEV), codes=b"TGCA")
    assert bad2 == 0 and (pk2.cpu().numpy().view(np.uint32)[:-1] == pack2.pack_numpy(seq2, b"TGCA")[:-1]).all()
    assert bytes(pack2.unpack_dev(pk2, 999, codes=b"TGCA").cpu().numpy()) == bytes(seq2)


def _index(n_text, seed):
    g = synth.genome(n_text, seed)
    sa = suffix_array(g)
    b = bwt(g, sa)
    ls = less(b, b"ACGTNacgtn")
    return g, b, ls, FMIndex(b, ls, Occ(b, 128, b"ACGTNacgtn"))


@pytest.mark.parametrize("ragged", [False, True])
def test_backward_search_on_packed_patterns_equals_the_byte_flavour_and_the_oracle(ragged):
    g, b, ls, fm = _index(300_000, 5)
    n_q, P = 20_011, 61
    pat, off = synth.fm_patterns(g, n_q, P, seed=8)
    if ragged:  # patterns of every length 1..P (some beyond the fast kernel's 256-symbol slot), starting anywhere in the stream
        lens = np.random.default_rng(2).integers(1, P + 1, size=n_q)
        lens[::97] = np.random.default_rng(3).integers(257, 400, size=len(lens[::97]))
        starts = np.random.default_rng(4).integers(0, len(g) - 401, size=n_q)
        pat = np.concatenate([pat[int(off[q]):int(off[q]) + int(lens[q])] if lens[q] <= P else g[int(starts[q]):int(starts[q]) + int(lens[q])]
                              for q in range(n_q)])
        off = np.zeros(n_q + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
    codes = fm.pattern_codes()
    assert sorted(codes) == sorted(b"ACGT")
    d_pat = torch.from_numpy(pat.copy()).to(DEV)
    d_off = torch.from_numpy(off.astype(np.int64)).to(DEV)
    pk, bad = pack2.pack_dev(d_pat, codes=codes)
    assert bad == 0
    outs = []
    for packed, ilp in ((False, 1), (True, 1), (False, 2), (True, 2)):  # (ilp: queries per quad, fm_search_fast2x_kernel from 2 on)
        fm.set_option("ilp", ilp)
        tag = torch.empty(n_q, dtype=torch.uint8, device=DEV)
        lo, hi = torch.empty(n_q, dtype=torch.int64, device=DEV), torch.empty(n_q, dtype=torch.int64, device=DEV)
        ml = torch.empty(n_q, dtype=torch.int32, device=DEV)
        if packed:
            fm.backward_search_packed_dev(n_q, pk.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo.data_ptr(), hi.data_ptr(), ml.data_ptr())
        else:
            fm.backward_search_dev(n_q, d_pat.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo.data_ptr(), hi.data_ptr(), ml.data_ptr())
        torch.cuda.synchronize()
        outs.append((tag.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy(), ml.cpu().numpy()))
    for cols in zip(*outs):
        assert all((cols[0] == c).all() for c in cols[1:])
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, orc.Occ(b, 128, b"ACGTNacgtn"), pat, off, threads=8)
    t, lo, hi, ml = outs[1]
    assert (t == otag).all() and (lo.astype(np.uint64) == olo).all() and (hi.astype(np.uint64) == ohi).all() and (ml.astype(np.uint64) == oml).all()
    # the counted flavour answers the same and reports between one and two block loads per (double) LF step
    tag = torch.empty(n_q, dtype=torch.uint8, device=DEV)
    lo_t, hi_t = torch.empty(n_q, dtype=torch.int64, device=DEV), torch.empty(n_q, dtype=torch.int64, device=DEV)
    ml_t = torch.empty(n_q, dtype=torch.int32, device=DEV)
    fm.set_option("ilp", 2)
    lines = fm.backward_search_count_lines_dev(n_q, d_pat.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo_t.data_ptr(), hi_t.data_ptr(), ml_t.data_ptr())
    fm.set_option("ilp", 1)
    assert lines == fm.backward_search_count_lines_dev(n_q, d_pat.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo_t.data_ptr(), hi_t.data_ptr(), ml_t.data_ptr())
    assert (tag.cpu().numpy() == otag).all() and (lo_t.cpu().numpy().astype(np.uint64) == olo).all()
    steps = int(oml.sum()) + int((otag != 0).sum())
    # (an index with 2-step rank blocks — this one — takes two LF steps per block access)
    assert (steps // 2 if fm.step2_bytes() else steps) <= lines <= 2 * steps


def test_packed_patterns_need_a_four_letter_index():
    text = np.frombuffer(b"ACACACCACAACCA$", dtype=np.uint8).copy()  # two letters only: codes 2, 3 stand for nothing
    sa = suffix_array(text)
    b = bwt(text, sa)
    fm = FMIndex(b, less(b, b"ACGT"), Occ(b, 4, b"ACGT"))
    with pytest.raises(_lib.BiogpuError):
        fm.pattern_codes()


@pytest.mark.parametrize("mode,L,scores", [(3, 150, (-5, -1, 1, -1)), (2, 150, (-5, -1, 1, -1)), (1, 96, (-3, -1, 2, -2)),
                                            (0, 100, (-5, -1, 1, -1)), (3, 400, (-5, -1, 1, -1)), (3, 150, (-500, -100, 100, -100))])
def test_aligner_on_packed_pairs_equals_the_byte_flavour(mode, L, scores):
    """K1p reads the codes directly (the first four cases: every wrapper); reads beyond K1p's 384 rows and scores beyond
    its 12 bits go through the unpack fallback — all must give the byte flavour's records and operations"""
    n = 3000
    x, xo, y, yo = synth.sw_pairs(n, L, seed=21)
    sc = Scoring.from_scores(*scores)
    if mode == 0:
        sc = sc.xclip(-3).yclip_prefix_(-2).yclip_suffix_(0)
    al = Aligner.with_scoring(sc)
    dx, dy = torch.from_numpy(x.copy()).to(DEV), torch.from_numpy(y.copy()).to(DEV)
    dxo, dyo = torch.from_numpy(xo.astype(np.int64)).to(DEV), torch.from_numpy(yo.astype(np.int64)).to(DEV)
    xpk, bx = pack2.pack_dev(dx)
    ypk, by = pack2.pack_dev(dy)
    assert bx == 0 and by == 0
    stride = 2 * L + 4
    res = []
    for packed in (False, True):
        out = torch.zeros(n * 64, dtype=torch.uint8, device=DEV)
        ops = torch.zeros(n * stride, dtype=torch.uint8, device=DEV)
        if packed:
            al.align_packed_dev(mode, n, xpk.data_ptr(), dxo.data_ptr(), ypk.data_ptr(), dyo.data_ptr(), L, L, out.data_ptr(), ops.data_ptr(), stride)
        else:
            al.align_dev(mode, n, dx.data_ptr(), dxo.data_ptr(), dy.data_ptr(), dyo.data_ptr(), L, L, out.data_ptr(), ops.data_ptr(), stride)
        torch.cuda.synchronize()
        res.append((out.cpu().numpy().view(_lib.ALN_DTYPE), ops.cpu().numpy().reshape(n, stride)))
    (r0, o0), (r1, o1) = res
    assert r0.tobytes() == r1.tobytes()
    mask = np.arange(stride)[None, :] >= (stride - r0["n_ops"].astype(np.int64))[:, None]
    assert (o0[mask] == o1[mask]).all()
    assert (r0["n_ops"] > 0).mean() > 0.9


def test_aligner_on_packed_ragged_pairs_starting_anywhere_in_the_stream():
    """sequences of different lengths back to back: a pair's symbols start at any 2-bit position of a dword"""
    rng = np.random.default_rng(4)
    n = 2000
    xs = [synth.random_dna(int(l), seed=100 + i) for i, l in enumerate(rng.integers(30, 151, size=n))]
    ys = [np.concatenate([s[: len(s) // 2], synth.random_dna(3, seed=7), s[len(s) // 2:]]) for s in xs]
    x, xo = _lib.concat([bytes(s) for s in xs])
    y, yo = _lib.concat([bytes(s) for s in ys])
    al = Aligner.with_scoring(Scoring.from_scores(-5, -1, 1, -1))
    dx, dy = torch.from_numpy(x.copy()).to(DEV), torch.from_numpy(y.copy()).to(DEV)
    dxo, dyo = torch.from_numpy(xo.astype(np.int64)).to(DEV), torch.from_numpy(yo.astype(np.int64)).to(DEV)
    xpk, _ = pack2.pack_dev(dx)
    ypk, _ = pack2.pack_dev(dy)
    stride = 150 + 153 + 4
    res = []
    for packed in (False, True):
        out = torch.zeros(n * 64, dtype=torch.uint8, device=DEV)
        ops = torch.zeros(n * stride, dtype=torch.uint8, device=DEV)
        f = al.align_packed_dev if packed else al.align_dev
        f(3, n, (xpk if packed else dx).data_ptr(), dxo.data_ptr(), (ypk if packed else dy).data_ptr(), dyo.data_ptr(), 150, 153,
          out.data_ptr(), ops.data_ptr(), stride)
        torch.cuda.synchronize()
        res.append((out.cpu().numpy().view(_lib.ALN_DTYPE), ops.cpu().numpy().reshape(n, stride)))
    (r0, o0), (r1, o1) = res
    assert r0.tobytes() == r1.tobytes()
    mask = np.arange(stride)[None, :] >= (stride - r0["n_ops"].astype(np.int64))[:, None]
    assert (o0[mask] == o1[mask]).all()
