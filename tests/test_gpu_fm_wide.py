"""The FM engine on 64-bit text positions (csrc/fm_wide.hip, sa_build.hip's uint64 instantiation) — the layout texts of
2^32 - 1 symbols and more get (the reference indexes with usize: fmindex.rs:70-71, bwt.rs:94, suffix_array.rs:264).
Here the same code paths are FORCED onto small texts (ctx option fm_wide_from = 1, superblocks of 4 blocks = 768 symbols,
so that an interval crosses dozens of superblock bases) and compared with the oracle exactly like the 32-bit engine is in
test_gpu_fm.py / test_gpu_sa_locate.py / test_gpu_sa_build.py.  The 4.4 G-symbol run itself is tools/exp/fm_wide_big.py
(profiles/r05_fm_wide_4g4.json): it needs 220 GB of HBM and minutes, not a unit test."""
import numpy as np
import pytest
import torch

import oracle_py as orc
from rust_bio_amd import _lib, synth
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.suffix_array import RawSuffixArray, SampledSuffixArray, bwt_dev, sample_dev, suffix_array, suffix_array_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_ALPHABET = b"ACGTNacgtn"
UNSUPPORTED = -11


def wide_ctx(sb_shift=2):
    ctx = _lib.Context(0)
    ctx.set_option("fm_wide_from", 1)
    ctx.set_option("fm_wide_sb_shift", sb_shift)
    return ctx


def patterns(text, rng, n, lo=1, hi=60, foreign=b""):
    body = text[:-1]
    pats = []
    for k in range(n):
        L = int(rng.integers(lo, hi))
        r = rng.random()
        if r < 0.6:  # a substring of the text
            s = int(rng.integers(0, max(1, len(body) - L)))
            p = body[s:s + L].copy()
        else:        # random: most of them end Partial / Absent
            p = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L)].copy()
        if r > 0.9 and L > 2:
            p[int(rng.integers(0, L))] = ord("N")
        if foreign and r > 0.97:
            p[int(rng.integers(0, L))] = foreign[0]
        pats.append(p.tobytes())
    pats += [b"", b"A", b"$", body[-30:].tobytes(), body[:25].tobytes()]
    return pats


def check_search(fm, b, ls, alphabet, pats):
    occ = orc.Occ(b, 3, alphabet)
    buf, off = _lib.concat(pats)
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, buf, off, threads=4)
    panics = otag == 3
    try:
        tag, lo, hi, ml = fm.backward_search_arrays(buf, off)
        assert not panics.any()
    except _lib.AlphabetError:
        assert panics.any()
        tag, lo, hi, ml = fm.last_raw
    assert (tag == otag).all()
    ok = ~panics
    assert (lo[ok] == olo[ok]).all() and (hi[ok] == ohi[ok]).all() and (ml[ok].astype(np.uint64) == oml[ok]).all()
    # the device-resident entry point
    d_pat, d_off = torch.from_numpy(buf.copy()).to(DEV), torch.from_numpy(off.astype(np.int64)).to(DEV)
    nq = len(pats)
    d_tag = torch.zeros(nq, dtype=torch.uint8, device=DEV)
    d_lo, d_hi = torch.zeros(nq, dtype=torch.int64, device=DEV), torch.zeros(nq, dtype=torch.int64, device=DEV)
    d_ml = torch.zeros(nq, dtype=torch.int32, device=DEV)
    fm.backward_search_dev(nq, d_pat.data_ptr(), d_off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr())
    torch.cuda.synchronize()
    assert (d_tag.cpu().numpy() == otag).all()
    assert (d_lo.cpu().numpy().astype(np.uint64)[ok] == olo[ok]).all() and (d_hi.cpu().numpy().astype(np.uint64)[ok] == ohi[ok]).all()
    # ... which runs two queries per quad by default (fmw_search2x_kernel): the one-query kernel answers the same
    fm.set_option("ilp", 1)
    d_tag1, d_lo1, d_hi1, d_ml1 = torch.zeros_like(d_tag), torch.zeros_like(d_lo), torch.zeros_like(d_hi), torch.zeros_like(d_ml)
    fm.backward_search_dev(nq, d_pat.data_ptr(), d_off.data_ptr(), d_tag1.data_ptr(), d_lo1.data_ptr(), d_hi1.data_ptr(), d_ml1.data_ptr())
    torch.cuda.synchronize()
    fm.set_option("ilp", 2)
    assert torch.equal(d_tag, d_tag1) and torch.equal(d_ml, d_ml1)
    okd = torch.from_numpy(ok).to(DEV)
    assert torch.equal(d_lo[okd], d_lo1[okd]) and torch.equal(d_hi[okd], d_hi1[okd])
    return tag, lo, hi


def test_wide_index_from_host_bwt_searches_and_locates_like_the_oracle():
    rng = np.random.default_rng(17)
    g = synth.genome(60_000, 5).copy()
    g[rng.integers(0, 60_000, size=40)] = ord("N")  # sparse exceptions besides the sentinel
    sa = suffix_array(g)
    b = bwt(g, sa)
    ls = less(b, N_ALPHABET)
    for sb_shift in (0, 2, 17):  # one block per superblock, four, and a single superblock for the whole text
        ctx = wide_ctx(sb_shift)
        fm = FMIndex(b, ls, Occ(b, 3, N_ALPHABET), ctx=ctx)
        assert fm.step2_bytes() == 0
        pats = patterns(g, rng, 3000, foreign=b"X")
        tag, lo, hi = check_search(fm, b, ls, N_ALPHABET, pats)
        assert (tag == 0).sum() > 500 and (tag == 1).sum() > 200
        # Interval::occ over the raw suffix array and over sampled ones (suffix_array.rs:134-184)
        hit = (tag == 0) & (hi - lo < 50)
        want = np.concatenate([sa[int(a):int(e)] for a, e in zip(lo[hit], hi[hit])])
        RawSuffixArray(sa, fm)
        _, pos = fm.interval_occ_arrays(lo[hit], hi[hit])
        assert (pos == want).all()
        for rate in (1, 5, 32):
            SampledSuffixArray(sa, g, b, rate, fmindex=fm)
            _, pos = fm.interval_occ_arrays(lo[hit], hi[hit])
            assert (pos == want).all(), rate
        # what a 64-bit index does not offer says so (biogpu.h)
        import ctypes as C
        cb = (C.c_uint8 * 4)()
        assert _lib.lib().bg_fm_pattern_codes(fm.h, cb) == UNSUPPORTED
        assert _lib.lib().bg_fm_set_text(fm.h, g.ctypes.data, len(g)) == UNSUPPORTED
        fm.close()


def test_wide_index_over_several_sentinels_and_fewer_than_four_letters():
    rng = np.random.default_rng(3)
    ac = np.frombuffer(b"AC", dtype=np.uint8)
    t = np.concatenate([ac[rng.integers(0, 2, size=4000)], np.frombuffer(b"$", np.uint8), ac[rng.integers(0, 2, size=3000)], np.frombuffer(b"$", np.uint8)])
    sa = suffix_array(t)
    b = bwt(t, sa)
    alpha = b"AC$"
    ls = less(b, alpha)
    fm = FMIndex(b, ls, Occ(b, 3, alpha), ctx=wide_ctx(1))
    pats = [ac[rng.integers(0, 2, size=int(rng.integers(1, 30)))].tobytes() for _ in range(500)] + [b"$", b"A$", b"$A", b"G"]
    check_search(fm, b, ls, alpha, pats)


def test_wide_refuses_what_needs_rank_bit_vectors():
    rng = np.random.default_rng(9)
    prot = np.append(np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)[rng.integers(0, 20, size=50_000)], np.uint8(ord("$")))
    sa = suffix_array(prot)
    b = bwt(prot, sa)
    alpha = b"ARNDCQEGHILKMFPSTWYV$"
    with pytest.raises(Exception) as e:
        FMIndex(b, less(b, alpha), Occ(b, 3, alpha), ctx=wide_ctx())
    assert "UNSUPPORTED" in str(e.value).upper() or "-11" in str(e.value)


def sa_texts():
    from test_gpu_sa_build import texts
    return [(n, t) for n, t in texts() if n not in ("random_1m",)] + [("random_300k", synth.genome(300_000, 3))]


@pytest.mark.parametrize("name,text", sa_texts(), ids=[t[0] for t in sa_texts()])
def test_device_suffix_array_64_equals_the_oracle(name, text):
    """bg_suffix_array_dev64 / bg_bwt_dev64: uint64 positions, a doubling round as two stable radix passes"""
    d_text = torch.from_numpy(np.ascontiguousarray(text)).to(DEV)
    d_sa = suffix_array_dev(d_text, wide=True)
    torch.cuda.synchronize()
    assert d_sa.dtype == torch.int64
    got = d_sa.cpu().numpy().astype(np.uint64)
    osa = np.asarray(orc.suffix_array(text), dtype=np.uint64)
    assert (got == osa).all(), name
    d_b = bwt_dev(d_text, d_sa)
    assert bytes(d_b.cpu().numpy()) == bytes(orc.bwt(text, osa)), name


def test_text_to_searchable_wide_index_entirely_on_the_device():
    """text in HBM -> 64-bit suffix array -> BWT -> samples -> wide index -> search + locate, against the oracle"""
    rng = np.random.default_rng(23)
    g = synth.genome(200_000, 11)
    d_text = torch.from_numpy(g).to(DEV)
    ctx = wide_ctx(3)
    d_sa = suffix_array_dev(d_text, ctx=ctx, wide=True)
    d_b = bwt_dev(d_text, d_sa, ctx=ctx)
    fm = FMIndex.from_device(d_b, 3, N_ALPHABET, ctx=ctx)
    sa = np.asarray(orc.suffix_array(g), dtype=np.uint64)
    b = np.frombuffer(bytes(orc.bwt(g, sa)), dtype=np.uint8)
    assert (fm._less == less(b, N_ALPHABET)).all()
    pats = patterns(g, rng, 4000, lo=8, hi=40)
    tag, lo, hi = check_search(fm, b, fm._less, N_ALPHABET, pats)
    s = sample_dev(d_sa, d_b, int(g[-1]), 16, ctx=ctx)
    want_s = SampledSuffixArray(sa, g, b, 16)
    assert (s.sample == want_s.sample).all() and (s.extra_rows == want_s.extra_rows).all() and (s.extra_pos == want_s.extra_pos).all()
    s.attach(fm)
    hit = (tag == 0) & (hi - lo < 30)
    _, pos = fm.interval_occ_arrays(lo[hit], hi[hit])
    assert (pos == np.concatenate([sa[int(a):int(e)] for a, e in zip(lo[hit], hi[hit])])).all()


def test_narrow_indexes_are_untouched_by_the_threshold_default():
    """without the test options an index below 2^32 - 1 symbols keeps the uint32 layout (2-step blocks and all)"""
    g = synth.genome(50_000, 3)
    sa = suffix_array(g)
    b = bwt(g, sa)
    fm = FMIndex(b, less(b, N_ALPHABET), Occ(b, 3, N_ALPHABET), ctx=_lib.Context(0))
    assert fm.step2_bytes() > 0
