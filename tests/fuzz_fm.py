"""Long differential fuzz of K5 (run by hand on a GPU box: python tests/fuzz_fm.py SEED SECONDS): random texts (DNA with
N runs and lower case, texts over fewer than four letters, protein texts: every index layout), random patterns — exact
substrings, substrings with substitutions, random words, bytes outside the alphabet, lengths 0 .. 400 (beyond the fast
kernel's 256-symbol LDS slot), at every byte alignment — through the byte entry point (fm_search_fast_kernel + the
deferred generic kernel where the index allows it), the generic kernel alone (option no_fast) and, where the index
takes them, 2-bit packed patterns; all against the CPU oracle (fmindex.rs:144-208).
Round 3 (final K5: wave-cooperative fetch, straight-line step): seeds 31 and 32, 16 673 066 queries in 3164 index / batch
configurations, 993 311 of them also as packed streams, 0 mismatches.
Round 3, final: seed 74: 3 790 678 queries (922 101 also packed), 0 mismatches.
Round 6: `python tests/fuzz_fm.py SEED SECONDS wide` — every index on the 64-bit layout (ctx option fm_wide_from = 1, a random
fm_wide_sb_shift so that short texts span many superblocks): fmw_search_kernel (ilp 1), fmw_search2x_kernel (ilp 2), the
2-step blocks on 64-bit positions with byte and packed patterns (where the text has them), single steps (no_step2), and
Interval::occ through 64-bit samples against the suffix array's rows."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import oracle_py as orc  # noqa: E402
from rust_bio_amd import _lib, pack2  # noqa: E402
from rust_bio_amd.bwt import Occ, bwt, less  # noqa: E402
from rust_bio_amd.fmindex import FMIndex  # noqa: E402
from rust_bio_amd.suffix_array import suffix_array  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
DEV = "cuda:0"
t0 = time.time()
n_located = 0
n_q_total = n_fail = rounds = n_packed = 0
ALPHAS = [(b"ACGT", b"ACGTNacgtn"), (b"ACGT", b"ACGT"), (b"AC", b"ACGT"), (b"ACG", b"ACGTN"), (b"ARNDCQEGHILKMFPSTWYV", b"ARNDCQEGHILKMFPSTWYVBZX")]


def search(fm, d_pat, d_off, n_q, packed=None):
    tag = torch.full((n_q,), 99, dtype=torch.uint8, device=DEV)
    lo, hi = torch.zeros(n_q, dtype=torch.int64, device=DEV), torch.zeros(n_q, dtype=torch.int64, device=DEV)
    ml = torch.zeros(n_q, dtype=torch.int32, device=DEV)
    if packed is not None:
        fm.backward_search_packed_dev(n_q, packed.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo.data_ptr(), hi.data_ptr(), ml.data_ptr())
    else:
        fm.backward_search_dev(n_q, d_pat.data_ptr(), d_off.data_ptr(), tag.data_ptr(), lo.data_ptr(), hi.data_ptr(), ml.data_ptr())
    torch.cuda.synchronize()
    return tag.cpu().numpy(), lo.cpu().numpy().astype(np.uint64), hi.cpu().numpy().astype(np.uint64), ml.cpu().numpy().astype(np.uint64)


while time.time() - t0 < budget and n_fail == 0:
    rounds += 1
    letters, alpha = ALPHAS[int(rng.integers(0, len(ALPHAS) - (1 if WIDE else 0)))]  # (protein on 64-bit positions: refused by design)
    ctx = None
    if WIDE:
        ctx = _lib.Context(0)
        ctx.set_option("fm_wide_from", 1)
        ctx.set_option("fm_wide_sb_shift", int(rng.integers(0, 7)))
    n = int(rng.choice([200, 5_000, 60_000, 400_000]))
    sym = np.frombuffer(letters, dtype=np.uint8)
    g = sym[rng.integers(0, len(sym), size=n)].copy()
    if b"N" in alpha and rng.random() < 0.5:  # N runs: a handful (sparse exceptions) or many (dense bit vector)
        for _ in range(int(rng.choice([1, 3, 40]))):
            s0 = int(rng.integers(0, n - 1))
            g[s0:s0 + int(rng.choice([1, 5, 400] if not WIDE else [1, 2, 20]))] = ord("N")  # (64-bit layout: at most 1024 listed positions)
    if b"a" in alpha and rng.random() < 0.3:
        s0 = int(rng.integers(0, n - 1))
        g[s0:s0 + 50] = np.frombuffer(b"acgt", dtype=np.uint8)[rng.integers(0, 4, size=len(g[s0:s0 + 50]))]
    text = np.append(g, np.uint8(ord("$")))
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, alpha)
    k = int(rng.choice([1, 3, 32, 128]))
    fm = FMIndex(b, ls, Occ(b, k, alpha), ctx=ctx)
    occ = orc.Occ(b, k, alpha)
    n_q = int(rng.choice([1, 63, 1000, 20_000]))
    pats = []
    clean = rng.random() < 0.4  # no byte outside the text's letters: the round also goes through the packed kernels
    junk = sym if clean else np.frombuffer(b"ACGTNacgtnXZ#$01", dtype=np.uint8)
    for q in range(n_q):
        kind = rng.random()
        ln = int(rng.choice([0, 1, 2, 15, 16, 17, 31, 64, 100, 255, 256, 257, 400])) if rng.random() < 0.3 else int(rng.integers(0, 130))
        ln = min(ln, n)
        s0 = int(rng.integers(0, n - ln + 1))
        p = g[s0:s0 + ln].copy()
        if clean:
            p = sym[rng.integers(0, len(sym), size=ln)] if kind < 0.2 else np.where(np.isin(p, sym), p, sym[0])
        if kind < 0.25 and ln:
            nm = int(rng.integers(1, 4))
            p[rng.integers(0, ln, size=nm)] = sym[rng.integers(0, len(sym), size=nm)]
        elif kind < 0.35:
            p = sym[rng.integers(0, len(sym), size=ln)]
        elif kind < 0.45 and ln:
            p[int(rng.integers(0, ln))] = junk[int(rng.integers(0, len(junk)))]
        pats.append(bytes(p))
    pat, off = _lib.concat(pats)
    if len(pat) == 0:
        pat = np.zeros(1, dtype=np.uint8)
    d_pat = torch.from_numpy(pat.copy()).to(DEV)
    d_off = torch.from_numpy(off.astype(np.int64)).to(DEV)
    otag, olo, ohi, oml = orc.backward_search_batch(b, ls, occ, pat, off, threads=8)
    ok = otag != 3
    variants = {}
    fm.set_option("no_fast", 0)
    variants["fast"] = search(fm, d_pat, d_off, n_q)
    fm.set_option("no_fast", 1)
    variants["generic"] = search(fm, d_pat, d_off, n_q)
    fm.set_option("no_fast", 0)
    if WIDE:
        fm.set_option("ilp", 1)
        variants["wide_ilp1"] = search(fm, d_pat, d_off, n_q)
        fm.set_option("no_fast", 1)
        variants["wide_generic_ilp1"] = search(fm, d_pat, d_off, n_q)
        fm.set_option("no_fast", 0)
        fm.set_option("no_step2", 1)
        variants["wide_ilp1_single_steps"] = search(fm, d_pat, d_off, n_q)
        fm.set_option("ilp", 2)
        variants["wide_ilp2_single_steps"] = search(fm, d_pat, d_off, n_q)
        fm.set_option("no_step2", 0)
        # Interval::occ through 64-bit samples (fmw_sampled_get_kernel) / the raw array against the suffix array's rows
        from rust_bio_amd.suffix_array import RawSuffixArray, SampledSuffixArray
        if rng.random() < 0.5:
            RawSuffixArray(sa, fm)
        else:
            SampledSuffixArray(sa, text, b, int(rng.choice([1, 2, 7, 32])), fm)
        hit = np.nonzero((otag == 0) & (ohi - olo <= 64))[0][:2000]
        if len(hit):
            _, got_pos = fm.interval_occ_arrays(olo[hit], ohi[hit])
            want_pos = np.concatenate([np.asarray(sa[int(a):int(e)], dtype=np.uint64) for a, e in zip(olo[hit], ohi[hit])])
            n_located += len(want_pos)
            if not (got_pos == want_pos).all():
                n_fail += 1
                print("MISMATCH locate round", rounds, "letters", letters, "n", n, flush=True)
    try:
        codes = fm.pattern_codes()
    except _lib.BiogpuError:
        codes = None
    if codes is not None and off[-1] > 0:
        pk, bad = pack2.pack_dev(d_pat[:int(off[-1])], codes=codes)
        if bad == 0:
            variants["packed"] = search(fm, d_pat, d_off, n_q, packed=pk)
            fm.set_option("no_fast", 1)
            variants["packed_generic"] = search(fm, d_pat, d_off, n_q, packed=pk)
            fm.set_option("no_fast", 0)
            n_packed += n_q
    for name, (tag, lo, hi, ml) in variants.items():
        good = (tag == otag).all() and (lo[ok] == olo[ok]).all() and (hi[ok] == ohi[ok]).all() and (ml[ok] == oml[ok]).all()
        if not good:
            n_fail += 1
            bad_q = int(np.nonzero((tag != otag) | ((lo != olo) & ok) | ((hi != ohi) & ok) | ((ml != oml) & ok))[0][0])
            print("MISMATCH", name, "round", rounds, "letters", letters, "alpha", alpha, "n", n, "k", k, "query", bad_q, pats[bad_q][:60],
                  "got", tag[bad_q], lo[bad_q], hi[bad_q], ml[bad_q], "want", otag[bad_q], olo[bad_q], ohi[bad_q], oml[bad_q], flush=True)
            break
    n_q_total += n_q
    del fm
    if ctx is not None:
        ctx.close()
print("rounds", rounds, "queries", n_q_total, "of them also packed", n_packed, *(("forced 64-bit layout, positions located", n_located) if WIDE else ()),
      "failures", n_fail, flush=True)
sys.exit(1 if n_fail else 0)
