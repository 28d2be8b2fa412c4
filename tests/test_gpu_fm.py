"""GPU parity tests of FM-index backward search (kernel K5 through the C ABI) against the
reference's KATs and the CPU oracle — bit-exact tags, intervals and matched lengths."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load
from rust_bio_amd import _lib, synth
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.suffix_array import suffix_array

pytestmark = pytest.mark.gpu
K = load("fm_kats.json")
TAGS = ["Complete", "Partial", "Absent"]


def build(text, alphabet, k):
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, alphabet)
    return sa, b, ls, FMIndex(b, ls, Occ(b, k, alphabet))


@pytest.mark.parametrize("case", K["backward_search"], ids=lambda c: c["name"])
def test_reference_kat(case):
    t = bytes(case["text_bytes"]) if "text_bytes" in case else case["text"].encode()
    alpha = bytes(case["alphabet_bytes"]) if "alphabet_bytes" in case else case["alphabet"].encode()
    pat = bytes(case["pattern_bytes"]) if "pattern_bytes" in case else case["pattern"].encode()
    sa, b, ls, fm = build(t, alpha, case["k"])
    res = fm.backward_search(pat)
    if "positions" in case:
        pos = [] if res.kind == "Absent" else res.interval.occ(sa)
        assert pos == case["positions"]
    if "tag" in case:
        assert res.kind.lower() == case["tag"]
    if "matched_len" in case:
        assert res.matched_len == case["matched_len"]
    if "not_tag" in case:
        assert res.kind.lower() != case["not_tag"]


def compare_with_oracle(text, alphabet, k, pats):
    sa, b, ls, fm = build(text, alphabet, k)
    occ = orc.Occ(b, k, alphabet)
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
    assert (lo[ok] == olo[ok]).all() and (hi[ok] == ohi[ok]).all()
    assert (ml[ok].astype(np.uint64) == oml[ok]).all()
    return tag


def test_random_texts_all_result_kinds():
    rng = np.random.default_rng(21)
    for trial in range(12):
        n = int(rng.integers(1, 3000))
        t = bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8)) + b"$"
        pats = []
        for _ in range(400):
            L = int(rng.integers(0, 40))
            if rng.random() < 0.5 and n > L:
                s = int(rng.integers(0, n - L + 1))
                p = bytearray(t[s:s + L])
                if rng.random() < 0.3 and L:
                    p[int(rng.integers(0, L))] = int(rng.choice(list(b"ACGTN")))
            else:
                p = bytearray(rng.choice(list(b"ACGTNacgtn$"), size=L).astype(np.uint8))
            pats.append(bytes(p))
        tag = compare_with_oracle(t, b"ACGTNacgtn", int(rng.choice([1, 3, 64, 128])), pats)
        assert set(tag.tolist()) >= {0, 1, 2}


def test_out_of_alphabet_bytes_match_reference_panics():
    t = b"GATTACAGATTACCA$"
    pats = [b"ATXACA", b"A~ACA", b"XGGACA", b"", b"N", b"GATTACA", b"\xff", b"TTAC"]
    compare_with_oracle(t, b"ACGTNacgtn", 3, pats)


def test_multi_sentinel_and_stray_symbols():
    # sentinels and N occupy BWT positions as exceptions of the packed layout
    rng = np.random.default_rng(33)
    reads = [bytes(rng.choice(list(b"ACGTN"), p=[.24, .24, .24, .24, .04], size=int(rng.integers(5, 80))).astype(np.uint8))
             for _ in range(40)]
    t = b"$".join(reads) + b"$"
    pats = [r[int(rng.integers(0, len(r))):][:int(rng.integers(1, 30))] for r in reads for _ in range(5)]
    pats += [b"$", b"A$", b"N", b"NN", b"$A"]
    compare_with_oracle(t, b"ACGTNacgtn", 8, pats)
    compare_with_oracle(t, b"ACGT", 8, pats)   # N becomes a panic symbol, '$' stays searchable


def test_two_symbol_text_and_smallest_symbol():
    t = b"AAA\x00"
    compare_with_oracle(t, b"\x00A", 3, [b"A\x00\x00", b"A", b"AA", b"AAA", b"AAAA", b"\x00", b"A\x00"])


def test_genome_1m_sample_bit_exact():
    g = synth.genome(1_000_000, 3)
    sa, b, ls, fm = build(g, b"ACGTNacgtn", 128)
    pat, off = synth.fm_patterns(g, 200_000, 100, seed=4)
    occ = orc.Occ(b, 128, b"ACGTNacgtn")
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, pat, off, threads=8)
    tag, lo, hi, ml = fm.backward_search_arrays(pat, off)
    assert (tag == otag).all() and (lo == olo).all() and (hi == ohi).all()
    assert (ml.astype(np.uint64) == oml).all()
    assert (tag == 0).mean() > 0.7 and (tag == 1).any()
    # size-independent property: every Complete interval lists exactly the occurrences
    for q in np.nonzero(tag == 0)[0][:200]:
        p = bytes(pat[int(off[q]):int(off[q + 1])])
        for r in range(int(lo[q]), int(hi[q])):
            s = int(sa[r])
            assert bytes(g[s:s + len(p)]) == p


def test_jump_table_does_not_change_results():
    """K5's table of the search state after the last 12 symbols (built by the search itself) against the
    plain LF loop: patterns shorter than, equal to and longer than 12, with N's, partial and absent ones."""
    g = synth.genome(300_000, 9)
    sa, b, ls, fm = build(g, b"ACGTNacgtn", 64)
    rng = np.random.default_rng(2)
    pats = []
    gb = g.tobytes()
    for _ in range(30_000):
        ln = int(rng.integers(1, 40))
        s = int(rng.integers(0, len(gb) - 50))
        p = bytearray(gb[s:s + ln])
        r = rng.random()
        if r < 0.3:
            p[int(rng.integers(0, ln))] = b"ACGT"[int(rng.integers(0, 4))]
        elif r < 0.35:
            p[int(rng.integers(0, ln))] = ord("N")
        elif r < 0.4:
            p = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=ln)].tobytes())
        pats.append(bytes(p))
    buf, off = _lib.concat(pats)
    fm.set_option("jump_min_queries", -1)
    plain = fm.backward_search_arrays(buf, off)
    fm.set_option("jump_min_queries", 0)
    fast = fm.backward_search_arrays(buf, off)
    assert fm.device_bytes() > 200_000_000  # the table exists
    for a_, b_ in zip(plain, fast):
        assert (a_ == b_).all()


def test_one_index_searched_from_several_threads():
    """`bg_fm` is immutable after construction: bg_fm_backward_search_batch_dev uses no shared scratch, so several host
    threads may search one handle at once, each on its own stream (include/biogpu.h, "Streams and threads")."""
    import threading
    import torch
    g = synth.genome(400_000, 13)
    sa, b, ls, fm = build(g, b"ACGTNacgtn", 64)
    occ = orc.Occ(b, 64, b"ACGTNacgtn")
    dev = torch.device("cuda:0")
    results, errors = {}, []

    def worker(t):
        try:
            pat, off = synth.fm_patterns(g, 60_000, 50, seed=100 + t)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                d_pat = torch.from_numpy(pat).to(dev)
                d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
                n = len(off) - 1
                tag = torch.empty(n, dtype=torch.uint8, device=dev)
                lo = torch.empty(n, dtype=torch.int64, device=dev)
                hi = torch.empty(n, dtype=torch.int64, device=dev)
                ml = torch.empty(n, dtype=torch.int32, device=dev)
                for _ in range(5):
                    fm.backward_search_dev(n, d_pat.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo.data_ptr(), hi.data_ptr(),
                                           ml.data_ptr(), st.cuda_stream)
                st.synchronize()
            results[t] = (pat, off, tag.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy(), ml.cpu().numpy())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for t, (pat, off, tag, lo, hi, ml) in results.items():
        otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, pat, off, threads=4)
        assert (tag == otag).all() and (lo.astype(np.uint64) == olo).all() and (hi.astype(np.uint64) == ohi).all()
        assert (ml.astype(np.uint64) == oml).all()


def test_fast_kernel_with_deferred_patterns_equals_the_generic_kernel_and_the_oracle():
    """DNA-like index: patterns are turned into 2-bit codes in LDS when a quad takes them (fm_search_fast_kernel); patterns
    with a byte outside the four codes (N, lower case, a byte the alphabet does not hold: the reference's panic), longer
    than the LDS slot (256 symbols), empty, at every byte alignment — left to the generic kernel — must all come out as
    the generic kernel alone (option no_fast) and the oracle give them"""
    import torch
    g = synth.genome(400_000, 17)
    g[1000:1040] = ord("N")
    sa = suffix_array(g)
    b = bwt(g, sa)
    alpha = b"ACGTNacgtn"
    ls = less(b, alpha)
    fm = FMIndex(b, ls, Occ(b, 64, alpha))
    rng = np.random.default_rng(23)
    pats = []
    for q in range(6000):
        kind = q % 12
        ln = int(rng.integers(1, 130))
        s0 = int(rng.integers(0, len(g) - 700))
        p = g[s0:s0 + ln].copy()
        if kind == 0:
            p = g[s0:s0 + int(rng.integers(257, 600))].copy()      # longer than the slot
        elif kind == 1:
            p[int(rng.integers(0, ln))] = ord("N")                  # in the alphabet, not a code
        elif kind == 2:
            p[int(rng.integers(0, ln))] = ord("a")                  # in the alphabet, never in the text
        elif kind == 3:
            p[int(rng.integers(0, ln))] = ord("X")                  # the reference panics if the search gets there
        elif kind == 4:
            p = p[:0]                                                # empty
        elif kind == 5:
            p = g[990:1010 + ln].copy()                              # runs into the N block of the text
        elif kind == 6 and ln > 3:
            p[ln // 2] = ord("ACGT"[(list(b"ACGT").index(p[ln // 2]) + 1) % 4])  # a substitution: mostly Partial
        pats.append(bytes(p))
    pat, off = _lib.concat(pats)
    dev = "cuda:0"
    d_pat = torch.from_numpy(pat.copy()).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    n_q = len(pats)
    res = []
    for no_fast in (0, 1):
        fm.set_option("no_fast", no_fast)
        tag = torch.full((n_q,), 77, dtype=torch.uint8, device=dev)
        lo, hi = torch.zeros(n_q, dtype=torch.int64, device=dev), torch.zeros(n_q, dtype=torch.int64, device=dev)
        ml = torch.zeros(n_q, dtype=torch.int32, device=dev)
        fm.backward_search_dev(n_q, d_pat.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo.data_ptr(), hi.data_ptr(), ml.data_ptr())
        torch.cuda.synchronize()
        res.append((tag.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy(), ml.cpu().numpy()))
    fm.set_option("no_fast", 0)
    for a, c in zip(*res):
        assert (a == c).all()
    tag, lo, hi, ml = res[0]
    assert set(np.unique(tag)) <= {0, 1, 2, 3} and (tag == 3).any() and (tag == 0).any() and (tag == 1).any() and (tag == 2).any()
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, orc.Occ(b, 64, alpha), pat, off, threads=8)
    ok = otag != 3  # a panicking query has no interval
    assert (tag == otag).all() and (lo.astype(np.uint64)[ok] == olo[ok]).all() and (hi.astype(np.uint64)[ok] == ohi[ok]).all()
    assert (ml.astype(np.uint64)[ok] == oml[ok]).all()


@pytest.mark.parametrize("text_kind", ["genome", "short_periodic", "two_sentinels", "dev_built"])
def test_two_step_rank_blocks_equal_single_steps_and_the_oracle(text_kind):
    """DNA-like indexes carry 2-step rank blocks (fm_step2.hip): the fast kernel takes two pattern symbols per block
    access and falls back to single steps from the same blocks for the odd last symbol and for the two steps of a pair
    that no row has in front.  Patterns of every length parity that end Complete, Partial after an even / an odd number
    of matched symbols, Absent at the first symbol or the second: identical to the single-step kernel (option no_step2)
    and to the oracle; also through the packed entry point and with the index built on the device."""
    import torch
    rng = np.random.default_rng({"genome": 1, "short_periodic": 2, "two_sentinels": 3, "dev_built": 4}[text_kind])
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    if text_kind == "short_periodic":
        g = np.append(np.tile(np.frombuffer(b"ACGTTGCAAC", dtype=np.uint8), 37), np.uint8(ord("$")))
    elif text_kind == "two_sentinels":
        fwd = acgt[rng.integers(0, 4, size=30_000)]
        rc_ = np.frombuffer(b"TGCA", dtype=np.uint8)[np.searchsorted(acgt, fwd[::-1])]
        g = np.concatenate([fwd, np.frombuffer(b"$", np.uint8), rc_, np.frombuffer(b"$", np.uint8)])
    else:
        g = synth.genome(300_000, 29)
    alpha = b"ACGTNacgtn"
    sa = suffix_array(g)
    b = bwt(g, sa)
    ls = less(b, alpha)
    if text_kind == "dev_built":
        d_b = torch.from_numpy(np.ascontiguousarray(b)).to("cuda:0")
        fm = FMIndex.from_device(d_b, 64, alpha)
    else:
        fm = FMIndex(b, ls, Occ(b, 64, alpha))
    pats = []
    body = g[:-1] if text_kind != "two_sentinels" else g[:30_000]
    for q in range(8000):
        ln = int(rng.integers(1, 70))
        s0 = int(rng.integers(0, max(1, len(body) - ln - 1)))
        p = body[s0:s0 + ln].copy()
        kind = q % 8
        if kind in (1, 2) and ln >= 2:   # a substitution somewhere: Partial after an even or an odd number of symbols
            k = int(rng.integers(0, ln))
            p[k] = acgt[(int(np.searchsorted(acgt, p[k])) + 1 + int(rng.integers(0, 3))) % 4]
        elif kind == 3:                  # random: mostly Partial after ~9-10 symbols
            p = acgt[rng.integers(0, 4, size=ln)]
        elif kind == 4 and ln >= 2:      # the very first (last) symbol pair is wrong in its second symbol
            p[-2] = acgt[(int(np.searchsorted(acgt, p[-2])) + 2) % 4]
        elif kind == 5:
            p = p[:int(rng.integers(1, 3))]  # one or two symbols
        elif kind == 6 and q % 40 == 6:
            p = p[:0]                        # empty: Absent without a search
        elif kind == 7 and q % 40 == 7:
            p[int(rng.integers(0, ln))] = ord("N")  # a byte without a 2-bit code: left to the generic kernel
        pats.append(bytes(p))
    pat, off = _lib.concat(pats)
    d_pat = torch.from_numpy(pat.copy()).to("cuda:0")
    d_off = torch.from_numpy(off.astype(np.int64)).to("cuda:0")
    n_q = len(pats)

    def run(no_step2, ilp=1):
        fm.set_option("no_step2", no_step2)
        fm.set_option("ilp", ilp)
        tag = torch.full((n_q,), 77, dtype=torch.uint8, device="cuda:0")
        lo, hi = torch.zeros(n_q, dtype=torch.int64, device="cuda:0"), torch.zeros(n_q, dtype=torch.int64, device="cuda:0")
        ml = torch.zeros(n_q, dtype=torch.int32, device="cuda:0")
        fm.backward_search_dev(n_q, d_pat.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo.data_ptr(), hi.data_ptr(), ml.data_ptr())
        torch.cuda.synchronize()
        fm.set_option("no_step2", 0)
        fm.set_option("ilp", 2)
        return tag.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy(), ml.cpu().numpy()

    two, one = run(0), run(1)
    for a, c in zip(two, one):
        assert (a == c).all()
    for ilp in (2,):  # two queries per quad (fm_search_fast2x_kernel; the default)
        for a, e in zip(two, run(0, ilp=ilp)):
            assert (a == e).all(), ilp
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, orc.Occ(b, 64, alpha), pat, off, threads=8)
    tag, lo, hi, ml = two
    assert (tag == otag).all() and (lo == olo.astype(np.int64)).all() and (hi == ohi.astype(np.int64)).all()
    assert (ml.astype(np.uint64) == oml).all()
    assert (tag == 0).any() and (tag == 1).any()
    assert ((ml[tag == 1] % 2) == 0).any() and ((ml[tag == 1] % 2) == 1).any()  # Partial after even and odd lengths


def test_host_entry_packs_patterns_on_the_host_and_falls_back_per_stage():
    """bg_fm_backward_search_batch stages its patterns as 2-bit codes (bg_pack2_host on the worker threads) and runs the packed
    kernel; a stage that holds a byte outside the four codes goes up as bytes.  2.2 M short patterns = three stages, the second
    one with foreign bytes: same arrays as with fm_host_bytes = 1 (everything as bytes), and as the oracle on a sample."""
    g = synth.genome(300_000, 11)
    sa, b, ls, fm = build(g, b"ACGTNacgtn", 64)
    rng = np.random.default_rng(5)
    nq = 2_200_000
    lens = rng.integers(0, 40, size=nq).astype(np.uint64)
    off = np.zeros(nq + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(g) - 64, size=nq)
    pat = np.empty(int(off[-1]), dtype=np.uint8)
    idx = np.repeat(starts, lens.astype(np.int64)) + (np.arange(int(off[-1])) - np.repeat(off[:-1].astype(np.int64), lens.astype(np.int64)))
    pat[:] = g[idx]
    mut = rng.integers(0, len(pat), size=len(pat) // 50)
    pat[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(mut))]
    # foreign bytes in the second stage only (queries 2^20 .. 2^21)
    q_bad = rng.integers(1 << 20, 1 << 21, size=2000)
    q_bad = q_bad[lens[q_bad] > 0]
    pat[(off[q_bad] + rng.integers(0, 1 << 30, size=len(q_bad)) % lens[q_bad]).astype(np.int64)] = ord("N")

    tag_b = np.zeros(nq, dtype=np.uint8); lo_b = np.zeros(nq, dtype=np.uint64); hi_b = np.zeros(nq, dtype=np.uint64); ml_b = np.zeros(nq, dtype=np.uint32)
    tag_p = np.zeros(nq, dtype=np.uint8); lo_p = np.zeros(nq, dtype=np.uint64); hi_p = np.zeros(nq, dtype=np.uint64); ml_p = np.zeros(nq, dtype=np.uint32)
    L = _lib.lib()
    rcs = []
    for opt, (t_, l_, h_, m_) in ((1, (tag_b, lo_b, hi_b, ml_b)), (0, (tag_p, lo_p, hi_p, ml_p))):
        fm.ctx.set_option("fm_host_bytes", opt)
        rcs.append(L.bg_fm_backward_search_batch(fm.h, nq, pat.ctypes.data, off.ctypes.data, t_.ctypes.data, l_.ctypes.data,
                                                 h_.ctypes.data, m_.ctypes.data))
    fm.ctx.set_option("fm_host_bytes", 0)
    assert rcs[0] == rcs[1]  # (the status of a search that reached a foreign byte, if any did)
    assert (tag_b == tag_p).all() and (lo_b == lo_p).all() and (hi_b == hi_p).all() and (ml_b == ml_p).all()
    # a sample of the clean first stage against the oracle
    occ = orc.Occ(b, 64, b"ACGTNacgtn")
    sel = np.arange(0, 20000)
    sub_off = off[:20001] - off[0]
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, pat[:int(off[20000])], sub_off, threads=8)
    assert (tag_p[sel] == otag).all() and (lo_p[sel] == olo).all() and (hi_p[sel] == ohi).all() and (ml_p[sel].astype(np.uint64) == oml).all()


@pytest.mark.parametrize("kind", ["dna_sampled_text", "dna_raw_sa", "protein_dense", "two_sentinels", "device_built", "wide"])
def test_index_survives_save_and_load(kind, tmp_path):
    """bg_fm_save / bg_fm_load (csrc/fm_persist.hip) — the reference derives Serialize / Deserialize for FMIndex, Occ and
    SampledSuffixArray (fmindex.rs:214, bwt.rs:76, suffix_array.rs:124).  The file holds the BWT (read back out of the rank
    blocks), less, alphabet, k, the attached suffix array and an owned text; the loaded handle answers backward_search,
    Interval::occ and seed-and-extend like the saved one — and like the oracle.  Every index layout: 2-bit blocks with
    listed exceptions, dense symbols (raw BWT kept), several sentinels, built on the device, 64-bit positions; a damaged
    file is refused (BG_ERR_IO)."""
    import torch
    from rust_bio_amd import pipeline
    from rust_bio_amd.pairwise import Scoring
    from rust_bio_amd.suffix_array import RawSuffixArray, SampledSuffixArray
    rng = np.random.default_rng(["dna_sampled_text", "dna_raw_sa", "protein_dense", "two_sentinels", "device_built", "wide"].index(kind) + 50)
    ctx = _lib.Context(0)
    alpha = b"ACGTNacgtn"
    if kind == "wide":
        ctx.set_option("fm_wide_from", 1)
        ctx.set_option("fm_wide_sb_shift", 2)
    if kind == "protein_dense":
        alpha = b"ACDEFGHIKLMNPQRSTVWY"
        g = np.append(np.frombuffer(alpha, dtype=np.uint8)[rng.integers(0, 20, size=40_000)], np.uint8(ord("$")))
    elif kind == "two_sentinels":
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        fwd = acgt[rng.integers(0, 4, size=20_000)]
        g = np.concatenate([fwd, np.frombuffer(b"$", np.uint8), np.frombuffer(b"TGCA", dtype=np.uint8)[np.searchsorted(acgt, fwd[::-1])],
                            np.frombuffer(b"$", np.uint8)])
    else:
        g = synth.genome(120_000, 41)
        g[5000:5040] = ord("N")  # listed exceptions next to the sentinel
    sa = suffix_array(g)
    b = bwt(g, sa)
    ls = less(b, alpha)
    if kind == "device_built":
        fm = FMIndex.from_device(torch.from_numpy(np.ascontiguousarray(b)).to("cuda:0"), 32, alpha, ctx=ctx)
    else:
        fm = FMIndex(b, ls, Occ(b, 32, alpha), ctx=ctx)
    if kind == "dna_raw_sa":
        RawSuffixArray(sa, fm)
    else:
        SampledSuffixArray(sa, g, b, 16, fm)
    with_text = kind == "dna_sampled_text"
    if with_text:
        pipeline.attach_text(fm, text=g)
    path = str(tmp_path / "index.bgfm")
    fm.save(path)
    fm2 = FMIndex.load(path, ctx=ctx)

    body = g[:20_000] if kind == "two_sentinels" else g[:-1]
    pats = []
    for q in range(3000):
        ln = int(rng.integers(1, 50))
        s0 = int(rng.integers(0, len(body) - ln))
        p = body[s0:s0 + ln].copy()
        if q % 3 == 1:
            p[int(rng.integers(0, ln))] = alpha[int(rng.integers(0, 4))]
        pats.append(bytes(p))
    pat, off = _lib.concat(pats)
    a, c = fm.backward_search_arrays(pat, off), fm2.backward_search_arrays(pat, off)
    for x, y in zip(a, c):
        assert (np.asarray(x) == np.asarray(y)).all()
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, orc.Occ(b, 32, alpha), pat, off, threads=4)
    tag, lo, hi, ml = c
    assert (tag == otag).all() and (lo.astype(np.uint64) == olo).all() and (hi.astype(np.uint64) == ohi).all()
    assert (tag == 0).any() and (tag == 1).any()
    # Interval::occ through the loaded handle's suffix array = the host suffix array's rows
    keep = np.nonzero(tag == 0)[0][:500]
    o1, p1 = fm.interval_occ_arrays(lo[keep], hi[keep])
    o2, p2 = fm2.interval_occ_arrays(lo[keep], hi[keep])
    assert (o1 == o2).all() and (p1 == p2).all()
    want = np.concatenate([np.asarray(sa[int(l):int(h)], dtype=np.uint64) for l, h in zip(lo[keep], hi[keep])])
    assert (p2 == want).all()
    # what the loaded handle says about itself: the text length, less, and the BWT read back out of the rank blocks
    # (bg_fm_len / bg_fm_less / bg_fm_bwt) — FMDIndex::from works on a deserialized FMIndex (fmindex.rs:311-329)
    assert len(fm2) == len(fm) == len(b)
    assert (fm2.less() == ls).all() and (fm2.bwt() == b).all()
    import torch as _t
    d_b = _t.empty(len(b), dtype=_t.uint8, device="cuda:0")
    _lib.check(_lib.lib().bg_fm_bwt_dev(fm2.h, d_b.data_ptr(), 0))
    _t.cuda.synchronize()
    assert (d_b.cpu().numpy() == b).all()
    if kind == "two_sentinels":  # T$R$: the loaded index is an FMD index like the saved one
        from rust_bio_amd.fmindex import FMDIndex
        fmd1, fmd2 = FMDIndex(fm), FMDIndex(fm2)
        reads = [bytes(g[s:s + 60]) for s in rng.integers(0, 19_000, size=50)]
        got1, got2 = fmd1.all_smems_batch(reads, 5), fmd2.all_smems_batch(reads, 5)
        assert got1 == got2
        ofmd = orc.FMDIndex(b, ls, orc.Occ(b, 32, alpha))
        for q, rb in enumerate(reads):
            assert [((iv.lower, iv.lower_rev, iv.size, iv.match_size), p_, ln) for iv, p_, ln in got2[q]] == ofmd.all_smems(rb, 5), q
    if with_text:  # the owned text travelled too: seed-and-extend on the loaded handle
        reads = [bytes(g[s:s + 100]) for s in rng.integers(0, len(g) - 200, size=200)]
        rd, ro = _lib.concat(reads)
        sc = Scoring.from_scores(-5, -1, 1, -1)
        h1 = pipeline.seed_extend_arrays(fm, sc, rd, ro)
        h2 = pipeline.seed_extend_arrays(fm2, sc, rd, ro)
        for x, y in zip(h1, h2):
            assert (np.asarray(x) == np.asarray(y)).all()
    # a damaged file is refused: one byte flipped in the BWT section, and a truncated copy
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x01
    open(path + ".flip", "wb").write(raw)
    open(path + ".cut", "wb").write(raw[:len(raw) - 16])
    for bad in (path + ".flip", path + ".cut", path + ".missing"):
        with pytest.raises(_lib.BiogpuError):
            FMIndex.load(bad, ctx=ctx)
    fm.close()
    fm2.close()
