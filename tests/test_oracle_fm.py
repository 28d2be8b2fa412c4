"""Pins the FM-index ORACLE (suffix array, BWT, less, Occ, backward_search) against the
reference's known-answer tests (tests/golden/fm_kats.json, transcribed from
/root/reference/src/data_structures/{suffix_array,bwt,fmindex}.rs)."""
import numpy as np
import pytest

import oracle_py as orc
from kat_util import load

K = load("fm_kats.json")


def _text(case):
    return bytes(case["text_bytes"]) if "text_bytes" in case else case["text"].encode()


@pytest.mark.parametrize("case", K["suffix_array"], ids=lambda c: c["text"][:12])
def test_suffix_array_kat(case):
    assert orc.suffix_array(case["text"].encode()).tolist() == case["sa"]


@pytest.mark.parametrize("text", K["sorts_lexically"]["texts"], ids=lambda t: t[:10])
def test_sorts_lexically(text):
    # suffix_array.rs:868-909: every suffix (cut at its first '$') is <= the next one
    t = text.encode()
    sa = orc.suffix_array(t)

    def s(i):
        return t[int(sa[i]):].split(b"$")[0] + b"$"
    for i in range(len(sa) - 2):
        assert s(i) <= s(i + 1)
    assert sorted(sa.tolist()) == list(range(len(t)))


def test_multi_sentinel_order():
    # suffix_array.rs:444-466: the first '$' gets the largest sentinel rank, the last one rank 0
    t = b"A$A$T$T$"
    sa = orc.suffix_array(t).tolist()
    assert sa[:4] == [7, 5, 3, 1]


def test_sentinel_assert():
    with pytest.raises(ValueError):
        orc.suffix_array(b"AC#GT$")  # '#' < '$' breaks "sentinel is smallest"


def test_bwt_kat():
    for c in K["bwt"]:
        t = c["text"].encode()
        assert bytes(orc.bwt(t, orc.suffix_array(t))) == c["bwt"].encode()


def test_bwtfind_pins_bwt():
    # bwt.rs:223-231: bwtfind of cabca$ — recomputed here from bwt + less
    c = K["bwtfind"]
    t = c["text"].encode()
    b = orc.bwt(t, orc.suffix_array(t))
    less = orc.less(b, c["alphabet"].encode()).tolist()
    find = [0] * len(b)
    for r, ch in enumerate(b):
        find[less[ch]] = r
        less[ch] += 1
    assert find == c["bwtfind"]


def test_occ_kat():
    c = K["occ"]
    occ = orc.Occ(np.array(c["bwt"], dtype=np.uint8), c["k"], bytes(c["alphabet"]))
    assert [occ.row(a) for a in range(4)] == c["table"]
    for g in c["get"]:
        assert occ.get(g["r"], g["a"]) == g["count"]


def test_occ_equals_naive_rank_all_positions():
    # bwt.rs:253-270 (Occ == WaveletMatrix rank for every position/symbol): the wavelet
    # matrix is out of scope, a naive count is the same oracle
    t = b"GCCTTAACATTATTACGCCTA$"
    b = orc.bwt(t, orc.suffix_array(t))
    for k in (1, 3, 7, 65, 128):
        occ = orc.Occ(b, k, b"ACGTNacgtn$")
        for c in b"ACGT$":
            for p in range(len(t)):
                assert occ.get(p, c) == int((b[:p + 1] == c).sum())


@pytest.mark.parametrize("case", K["backward_search"], ids=lambda c: c["name"])
def test_backward_search_kat(case):
    t = _text(case)
    alpha = bytes(case["alphabet_bytes"]) if "alphabet_bytes" in case else case["alphabet"].encode()
    pat = bytes(case["pattern_bytes"]) if "pattern_bytes" in case else case["pattern"].encode()
    sa = orc.suffix_array(t)
    b = orc.bwt(t, sa)
    less = orc.less(b, alpha)
    occ = orc.Occ(b, case["k"], alpha)
    tag, lo, hi, ml = orc.backward_search(b, less, occ, pat)
    if "positions" in case:
        pos = [] if tag == "absent" else [int(sa[i]) for i in range(lo, hi)]
        assert pos == case["positions"]
    if "tag" in case:
        assert tag == case["tag"]
    if "matched_len" in case:
        assert ml == case["matched_len"]
    if "not_tag" in case:
        assert tag != case["not_tag"] and tag != "panic"


def test_backward_search_out_of_alphabet_panics():
    # fmindex.rs:229 / bwt.rs:158: index out of bounds on a byte outside the alphabet —
    # but only if the loop reaches it
    t = b"GATTACA$"
    sa = orc.suffix_array(t)
    b = orc.bwt(t, sa)
    less = orc.less(b, b"ACGTNacgtn")
    occ = orc.Occ(b, 3, b"ACGTNacgtn")
    assert orc.backward_search(b, less, occ, b"ATXACA")[0] == "panic"   # 'X' not in alphabet
    assert orc.backward_search(b, less, occ, b"A~ACA")[0] == "panic"    # beyond less.len()
    assert orc.backward_search(b, less, occ, b"XGGACA")[0] == "partial"  # breaks before 'X'
    assert orc.backward_search(b, less, occ, b"")[0] == "absent"
    assert orc.backward_search(b, less, occ, b"N")[0] == "absent"       # in alphabet, absent


def test_backward_search_vs_bruteforce():
    rng = np.random.default_rng(11)
    for trial in range(20):
        n = int(rng.integers(5, 400))
        t = bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8)) + b"$"
        sa = orc.suffix_array(t)
        b = orc.bwt(t, sa)
        k = int(rng.choice([1, 3, 16, 64, 65, 128]))
        less = orc.less(b, b"ACGTNacgtn")
        occ = orc.Occ(b, k, b"ACGTNacgtn")
        for _ in range(30):
            plen = int(rng.integers(1, 12))
            if rng.random() < 0.6:
                s = int(rng.integers(0, max(1, n - plen)))
                p = t[s:s + plen].replace(b"$", b"A")
            else:
                p = bytes(rng.choice(list(b"ACGT"), size=plen).astype(np.uint8))
            tag, lo, hi, ml = orc.backward_search(b, less, occ, p)
            # longest suffix of p that occurs in t
            best = 0
            for L in range(1, len(p) + 1):
                if t.find(p[len(p) - L:]) >= 0:
                    best = L
                else:
                    break
            if best == len(p):
                assert tag == "complete" and ml == len(p)
            elif best == 0:
                assert tag == "absent"
            else:
                assert tag == "partial" and ml == best
            if tag != "absent":
                suf = p[len(p) - ml:]
                occs = sorted(int(sa[i]) for i in range(lo, hi))
                want = sorted(i for i in range(len(t)) if t.startswith(suf, i))
                assert occs == want


def test_intervals_by_scan_equal_backward_search_and_the_suffix_array():
    """orc_intervals_by_scan — the definition-level checker of the 4.4 G-symbol run (no suffix array) — pinned to the pinned
    restatement: for every pattern that occurs, (lower, upper) = backward_search's Complete interval and the positions are
    the suffix array's entries of that interval; for one that does not, an empty range at the pattern's insertion point"""
    import oracle_py as orc
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    text = np.append(acgt[rng.integers(0, 4, size=20_000)], np.uint8(ord("$")))
    text[rng.integers(0, 20_000, size=30)] = ord("N")
    alpha = b"ACGTN$"
    pats = [text[s:s + L].tobytes() for s, L in zip(rng.integers(0, 19_000, size=300), rng.integers(1, 14, size=300))]
    pats += [acgt[rng.integers(0, 4, size=int(L))].tobytes() for L in rng.integers(6, 16, size=200)] + [b"A", b"T", b"N", b"TTTTTTTTTTTTTTTTTTTT"]
    # round 6: the scan takes thousands of patterns per pass through an 8-byte key table — long patterns (many of them sharing
    # their first 8 bytes, some cut one symbol short or changed at the end, some running into the sentinel and past the end
    # of the text), duplicates, and a copied segment so that intervals hold several rows
    text[15_000:15_400] = text[3_000:3_400]
    for s0 in rng.integers(0, 19_900, size=400):
        L = int(rng.integers(8, 60))
        q = text[int(s0):int(s0) + L].tobytes()
        pats += [q, q[:-1] + b"A", q + b"T"]
    pats += [text[-12:].tobytes(), text[-9:].tobytes() + b"A", text[-8:].tobytes(), text[-30:-1].tobytes(), pats[-5], pats[-5]]
    sa = np.asarray(orc.suffix_array(text), dtype=np.uint64)
    b = np.frombuffer(bytes(orc.bwt(text, sa)), dtype=np.uint8)
    ls = orc.less(b, alpha)
    occ = orc.Occ(b, 3, alpha)
    buf = np.frombuffer(b"".join(pats), dtype=np.uint8)
    off = np.zeros(len(pats) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in pats])
    tag, lo, hi, ml = orc.backward_search_batch(b, ls, occ, buf, off)
    assert (tag == 0).sum() > 500 and (tag != 0).sum() > 500 and ((hi - lo)[tag == 0] > 1).sum() > 20
    for threads in (1, 3):
        slo, shi, spos = orc.intervals_by_scan(text, buf, off, pos_cap=100_000, threads=threads)
        for k, p in enumerate(pats):
            if tag[k] == 0:
                assert (int(slo[k]), int(shi[k])) == (int(lo[k]), int(hi[k])), p
                assert sorted(int(v) for v in sa[int(lo[k]):int(hi[k])]) == [int(v) for v in spos[k]]
            else:
                assert slo[k] == shi[k] and len(spos[k]) == 0
                # the insertion point: every suffix below it is smaller than the pattern, every one from it on larger
                r = int(slo[k])
                suf = lambda i: text[int(sa[i]):].tobytes()
                assert (r == 0 or suf(r - 1) < p) and (r == len(sa) or suf(r) > p)
