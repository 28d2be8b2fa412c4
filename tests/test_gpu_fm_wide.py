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
        assert fm.step2_bytes() == 0  # (41 listed positions: single steps — the 2-step blocks take at most four)
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
        # 2-bit packed patterns on single steps (fmw_search2x_kernel<PACKED>): the clean patterns as one stream
        from rust_bio_amd import pack2
        codes = fm.pattern_codes()
        clean = [p for p in pats if p and all(c in codes for c in p)]
        cbuf, coff = _lib.concat(clean)
        d_pat, d_off = torch.from_numpy(cbuf.copy()).to(DEV), torch.from_numpy(coff.astype(np.int64)).to(DEV)
        pk, bad = pack2.pack_dev(d_pat, codes=codes, ctx=ctx)
        assert bad == 0
        nq = len(clean)
        d_tag = torch.full((nq,), 9, dtype=torch.uint8, device=DEV)
        d_lo, d_hi = torch.zeros(nq, dtype=torch.int64, device=DEV), torch.zeros(nq, dtype=torch.int64, device=DEV)
        d_ml = torch.zeros(nq, dtype=torch.int32, device=DEV)
        fm.backward_search_packed_dev(nq, pk.data_ptr(), d_off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr())
        torch.cuda.synchronize()
        otag, olo, ohi, oml = orc.backward_search_batch(b, ls, orc.Occ(b, 3, N_ALPHABET), cbuf, coff, threads=4)
        assert (d_tag.cpu().numpy() == otag).all() and (d_lo.cpu().numpy().astype(np.uint64) == olo).all()
        assert (d_hi.cpu().numpy().astype(np.uint64) == ohi).all() and (d_ml.cpu().numpy().astype(np.uint64) == oml).all()
        # the uint32-record FMD entry points cannot hold a 64-bit index's intervals and say so (biogpu.h)
        cnt = np.zeros(1, dtype=np.uint32)
        out32 = np.zeros(6 * 8, dtype=np.uint32)
        one, ooff = _lib.concat([b"ACGTACGT"])
        assert _lib.lib().bg_fmd_smems_batch(fm.h, 1, 1, one.ctypes.data, ooff.ctypes.data, None, 1, 8, cnt.ctypes.data, out32.ctypes.data) == UNSUPPORTED
        fm.close()


def revcomp(s):
    return bytes(s[::-1].translate(bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")))


@pytest.mark.parametrize("sb_shift", [0, 3, 17])
def test_wide_two_step_blocks_packed_patterns_and_seeds(sb_shift):
    """Round 6: a 64-bit index over a DNA-like text (here T$R$: two sentinels, no N) has 2-step rank blocks with
    superblock-relative counters (fm_step2.hip: fm_build_step2_wide) and the 2x fast kernel runs on them
    (fm_search_fast2x_kernel<WIDE>): byte patterns (with deferral of what its LDS slot cannot hold), 2-bit packed
    patterns, the seed windows of reads — and the same through single steps; all equal the oracle's backward_search."""
    from rust_bio_amd import pack2
    rng = np.random.default_rng(100 + sb_shift)
    fwd = synth.random_dna(40_000, seed=31).tobytes()
    text = np.frombuffer(fwd + b"$" + revcomp(fwd) + b"$", dtype=np.uint8)
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, N_ALPHABET)
    ctx = wide_ctx(sb_shift)
    fm = FMIndex(b, ls, Occ(b, 3, N_ALPHABET), ctx=ctx)
    assert fm.step2_bytes() > 0
    pats = patterns(text, rng, 6000, foreign=b"X") + [text[100:500].tobytes(), text[7:7 + 256].tobytes(), text[9:9 + 257].tobytes()]
    tag, lo, hi = check_search(fm, b, ls, N_ALPHABET, pats)  # (also compares ilp 2 with ilp 1 = the generic kernel)
    assert (tag == 0).sum() > 1000 and (tag == 1).sum() > 400
    fm.set_option("no_step2", 1)
    assert fm.step2_bytes() == 0
    check_search(fm, b, ls, N_ALPHABET, pats)
    fm.set_option("no_step2", 0)
    codes = fm.pattern_codes()
    clean = [p for p in pats if p and all(c in codes for c in p)]
    cbuf, coff = _lib.concat(clean)
    d_pat, d_off = torch.from_numpy(cbuf.copy()).to(DEV), torch.from_numpy(coff.astype(np.int64)).to(DEV)
    pk, bad = pack2.pack_dev(d_pat, codes=codes, ctx=ctx)
    assert bad == 0
    nq = len(clean)
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, orc.Occ(b, 3, N_ALPHABET), cbuf, coff, threads=4)
    for no_fast in (0, 1):
        fm.set_option("no_fast", no_fast)
        d_tag = torch.full((nq,), 9, dtype=torch.uint8, device=DEV)
        d_lo, d_hi = torch.zeros(nq, dtype=torch.int64, device=DEV), torch.zeros(nq, dtype=torch.int64, device=DEV)
        d_ml = torch.zeros(nq, dtype=torch.int32, device=DEV)
        fm.backward_search_packed_dev(nq, pk.data_ptr(), d_off.data_ptr(), d_tag.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr(), d_ml.data_ptr())
        torch.cuda.synchronize()
        assert (d_tag.cpu().numpy() == otag).all() and (d_lo.cpu().numpy().astype(np.uint64) == olo).all(), no_fast
        assert (d_hi.cpu().numpy().astype(np.uint64) == ohi).all() and (d_ml.cpu().numpy().astype(np.uint64) == oml).all(), no_fast
    fm.set_option("no_fast", 0)
    fm.close()


@pytest.mark.parametrize("with_n", [True, False])
def test_wide_fmd_smems_equal_the_oracle(with_n):
    """K7 on 64-bit positions (fmd_smems.hip: fmd_smems_kernel<true, true, PLAIN>, uint64 records through bg_fmd_smems_batch64):
    FMDIndex::smems / all_smems (fmindex.rs:363-501) and the single interval steps (504-564) on T$R$ — with a few N (the
    general extension: eleven per-lane class dispatches) and without (the straight-line extension of plain ACGT indexes)"""
    from rust_bio_amd.fmindex import FMDIndex
    rng = np.random.default_rng(77)
    g = synth.random_dna(30_000, seed=5).copy()
    if with_n:
        g[rng.integers(0, len(g), size=6)] = ord("N")
    fwd = g.tobytes()
    text = np.frombuffer(fwd + b"$" + revcomp(fwd) + b"$", dtype=np.uint8)
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, N_ALPHABET)
    ofmd = orc.FMDIndex(b, ls, orc.Occ(b, 16, N_ALPHABET))
    for sb_shift in (0, 4):
        fm = FMIndex(b, ls, Occ(b, 16, N_ALPHABET), ctx=wide_ctx(sb_shift))
        fmd = FMDIndex(fm)
        reads, pos = [], []
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        for _ in range(300):
            s0 = int(rng.integers(0, len(fwd) - 130))
            r = np.frombuffer(fwd[s0:s0 + int(rng.integers(20, 120))], dtype=np.uint8).copy()
            nm = int(rng.integers(0, 4))
            r[rng.integers(0, len(r), size=nm)] = acgt[rng.integers(0, 4, size=nm)]
            rb = r.tobytes()
            if rng.random() < 0.5:
                rb = revcomp(rb)
            reads.append(rb)
            pos.append(int(rng.integers(0, len(rb))))
        got = fmd.smems_batch(reads, pos, 5)
        got_all = fmd.all_smems_batch(reads, 8)
        for q, rb in enumerate(reads):
            assert [((iv.lower, iv.lower_rev, iv.size, iv.match_size), p, ln) for iv, p, ln in got[q]] == ofmd.smems(rb, pos[q], 5), q
            assert [((iv.lower, iv.lower_rev, iv.size, iv.match_size), p, ln) for iv, p, ln in got_all[q]] == ofmd.all_smems(rb, 8), q
        iv = fmd.init_interval_with(ord("A"))
        assert (iv.lower, iv.lower_rev, iv.size, iv.match_size) == ofmd.init_interval_with(ord("A"))
        iv2 = fmd.backward_ext(iv, ord("C"))
        assert (iv2.lower, iv2.lower_rev, iv2.size, iv2.match_size) == ofmd.backward_ext(ofmd.init_interval_with(ord("A")), ord("C"))
        iv3 = fmd.forward_ext(iv2, ord("G"))
        assert (iv3.lower, iv3.lower_rev, iv3.size, iv3.match_size) == ofmd.forward_ext(ofmd.backward_ext(ofmd.init_interval_with(ord("A")), ord("C")), ord("G"))
        fm.close()


def test_narrow_index_answers_the_64_bit_fmd_records_too():
    from rust_bio_amd.fmindex import FMDIndex
    rng = np.random.default_rng(78)
    fwd = synth.random_dna(8_000, seed=6).tobytes()
    text = np.frombuffer(fwd + b"$" + revcomp(fwd) + b"$", dtype=np.uint8)
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, N_ALPHABET)
    fm = FMIndex(b, ls, Occ(b, 16, N_ALPHABET), ctx=_lib.Context(0))
    reads = [fwd[s0:s0 + 70] for s0 in rng.integers(0, 7_900, size=100)]
    assert FMDIndex(fm).all_smems_batch(reads, 6) == FMDIndex(fm, records32=True).all_smems_batch(reads, 6)


@pytest.mark.parametrize("with_n", [False, True])
def test_wide_seed_and_extend_equals_the_pipeline_oracle(with_n):
    """bg_fm_set_text + bg_seed_extend_batch on 64-bit positions (seed search: the SEEDS flavour of the wide kernels — the 2x
    fast kernel on 2-step blocks, or the generic one when stray N leave single steps; proposals sorted as uint64) against
    oracle/pipeline.cpp: best score, reference span, counts and the winner's operation list, read by read"""
    from rust_bio_amd.pairwise import Scoring
    from rust_bio_amd.pipeline import attach_text, seed_extend_arrays
    from test_gpu_pipeline import ALPHA, compare, make_case
    g, text, reads, off, starts = make_case(n_text=120_000, R=800)
    if with_n:
        text = text.copy()
        text[np.random.default_rng(2).integers(0, 119_000, size=20)] = ord("N")
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, ALPHA)
    ctx = wide_ctx(2)
    fm = FMIndex(b, ls, Occ(b, 64, ALPHA), ctx=ctx)
    assert (fm.step2_bytes() > 0) == (not with_n)
    SampledSuffixArray(sa, text, b, 8, fmindex=fm)
    attach_text(fm, text)
    sc = Scoring.from_scores(-5, -1, 1, -1)
    hits, ops = seed_extend_arrays(fm, sc, reads, off)
    ohits, oops, ostride = orc.seed_extend_batch(b, ls, orc.Occ(b, 64, ALPHA), sa, text, len(text) - 1, orc.make_scoring(-5, -1, 1, -1), reads, off, threads=8)
    mapped = compare(hits, ops, ohits, oops, ostride)
    assert mapped[:-50].mean() > 0.9
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
