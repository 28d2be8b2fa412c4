"""Seed-and-extend (BASELINE configs[4] shape, scaled down): `bg_seed_extend_batch[_dev]` against the same
composition made of the CPU oracle's three calls (oracle/pipeline.cpp: backward_search -> Interval::occ ->
Aligner::semiglobal, the caller pattern of /root/reference/src/lib.rs:129-165) — best score, reference span,
candidate / hit counts and the winner's complete operation list, read by read."""
import numpy as np
import pytest
import torch

import oracle_py as orc
from rust_bio_amd import _lib, synth
from rust_bio_amd.bwt import Occ, bwt, less
from rust_bio_amd.fmindex import FMIndex
from rust_bio_amd.pairwise import MIN_SCORE, Scoring
from rust_bio_amd.pipeline import SeedParams, attach_text, seed_extend_arrays, seed_extend_dev
from rust_bio_amd.suffix_array import RawSuffixArray, SampledSuffixArray, suffix_array

pytestmark = pytest.mark.gpu
ALPHA = b"ACGTNacgtn$"
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def make_case(n_text=200_000, R=1500, L=150, ragged=False):
    g = synth.random_dna(n_text, seed=31).copy()
    g[50_000:50_400] = g[10_000:10_400]  # a repeat: seeds with several occurrences
    text = np.append(g, np.uint8(ord("$")))
    rng = np.random.default_rng(3)
    starts = rng.integers(0, n_text - L, size=R)
    starts[:20] = np.arange(20) * 3             # reads hanging over the left end of the windows
    starts[20:40] = n_text - L - np.arange(20)  # ... and the right end
    starts[40:60] = 10_000 + np.arange(20) * 7  # inside the repeat: two candidates far apart
    refs = np.stack([g[s:s + L] for s in starts])
    reads, _ = synth.mutate_fixed(refs, 77, 0.04, 0.01, 0.01)
    reads[-50:] = synth.random_dna(50 * L, seed=5).reshape(50, L)  # unmappable reads
    if ragged:
        lens = rng.integers(15, L + 1, size=R)   # some shorter than one seed
        lens[:100] = L
        seqs = [reads[r, :lens[r]] for r in range(R)]
        flat = np.concatenate(seqs)
        off = np.zeros(R + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
    else:
        flat = np.ascontiguousarray(reads.reshape(-1))
        off = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    return g, text, flat, off, starts


def build(text, sampled):
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, ALPHA)
    fm = FMIndex(b, ls, Occ(b, 64, ALPHA))
    if sampled:
        SampledSuffixArray(sa, text, b, sampled, fmindex=fm)
    else:
        RawSuffixArray(sa, fm)
    return sa, b, ls, fm


def compare(hits, ops, ohits, oops, ostride, n_mapped_min=None):
    for f in ("n_candidates", "n_seed_hits", "window_start", "ref_start", "ref_end"):
        assert (hits[f] == ohits[f]).all(), f
    for f in ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen", "n_ops"):
        assert (hits["aln"][f].astype(np.int64) == ohits["aln"][f].astype(np.int64)).all(), f
    mapped = ohits["aln"]["score"] > MIN_SCORE
    assert (hits["aln"]["mode"][mapped] == 2).all()
    for r in np.nonzero(mapped)[0]:
        k, o = int(hits["aln"]["n_ops"][r]), int(hits["aln"]["ops_off"][r])
        want = (oops[r * ostride:r * ostride + k] & np.uint64(0xFF)).astype(np.uint8)
        assert (ops[o:o + k] == want).all(), r
    return mapped


@pytest.mark.parametrize("sampled", [0, 8])
@pytest.mark.parametrize("ragged", [False, True])
def test_seed_extend_matches_oracle_composition(sampled, ragged):
    g, text, reads, off, starts = make_case(ragged=ragged)
    sa, b, ls, fm = build(text, sampled)
    attach_text(fm, text)
    sc = Scoring.from_scores(-5, -1, 1, -1)
    hits, ops = seed_extend_arrays(fm, sc, reads, off)
    occ = orc.Occ(b, 64, ALPHA)
    ohits, oops, ostride = orc.seed_extend_batch(b, ls, occ, sa, text, len(g), orc.make_scoring(-5, -1, 1, -1), reads, off, threads=8)
    mapped = compare(hits, ops, ohits, oops, ostride)
    if not ragged:
        assert mapped[:-50].mean() > 0.95 and not mapped[-50:].any()
        near = np.abs(ohits["ref_start"][:-50][mapped[:-50]].astype(np.int64) - starts[:-50][mapped[:-50]]) <= 8
        assert near.mean() > 0.9  # most reads land where they were drawn from
        assert (ohits["n_candidates"][40:60] >= 2).mean() > 0.8  # the repeat proposes both copies


def test_seed_extend_device_resident_and_parameters():
    """device pointers in and out, other seed parameters, right-aligned operation slots"""
    g, text, reads, off, _ = make_case(n_text=120_000, R=700, L=100)
    sa, b, ls, fm = build(text, 4)
    dev = torch.device("cuda:0")
    d_text = torch.from_numpy(text).to(dev)
    attach_text(fm, d_text=d_text)
    prm = SeedParams(seed_len=16, stride=7, max_occ=4, pad=12)
    sc = Scoring.from_scores(-4, -2, 2, -3)
    R, L = len(off) - 1, 100
    d_reads = torch.from_numpy(reads).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    stride = 2 * L + 2 * prm.pad + 4
    d_hits = torch.zeros(R * 96, dtype=torch.uint8, device=dev)
    d_ops = torch.zeros(R * stride, dtype=torch.uint8, device=dev)
    tot = np.zeros(2, dtype=np.uint64)
    seed_extend_dev(fm, sc, R, d_reads.data_ptr(), d_off.data_ptr(), L, d_hits.data_ptr(), d_ops.data_ptr(), stride, prm,
                    torch.cuda.current_stream().cuda_stream, tot)
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().view(_lib.SEED_HIT_DTYPE)
    ops = d_ops.cpu().numpy()
    occ = orc.Occ(b, 64, ALPHA)
    ohits, oops, ostride = orc.seed_extend_batch(b, ls, occ, sa, text, len(g), orc.make_scoring(-4, -2, 2, -3), reads, off,
                                                 seed_len=16, stride=7, max_occ=4, pad=12, threads=8)
    compare(hits, ops, ohits, oops, ostride)
    assert (hits["aln"]["ops_off"] == (np.arange(R) + 1) * stride - hits["aln"]["n_ops"]).all()
    assert int(tot[0]) == int(ohits["n_seed_hits"].sum()) and int(tot[1]) == int(ohits["n_candidates"].sum())


def test_seed_extend_needs_text_and_suffix_array():
    g, text, reads, off, _ = make_case(n_text=60_000, R=100, L=80)
    sa = suffix_array(text)
    b = bwt(text, sa)
    ls = less(b, ALPHA)
    fm = FMIndex(b, ls, Occ(b, 64, ALPHA))
    with pytest.raises(_lib.BiogpuError):
        seed_extend_arrays(fm, Scoring.from_scores(-5, -1, 1, -1), reads, off)


def test_seed_outside_the_alphabet_is_reported_and_does_not_vote():
    """a byte the index's alphabet does not hold inside a seed: the reference's backward_search panics (fmindex.rs:229);
    the call reports it, the seed does not vote, every read is still answered"""
    g, text, reads, off, _ = make_case(n_text=60_000, R=200, L=100)
    sa, b, ls, fm = build(text, 8)
    attach_text(fm, text)
    reads = reads.copy()
    reads[5 * 100 + 37] = ord("X")          # read 5: two of its seeds cover the byte
    sc = Scoring.from_scores(-5, -1, 1, -1)
    with pytest.raises(_lib.AlphabetError):
        seed_extend_arrays(fm, sc, reads, off)
    hits, ops = seed_extend_arrays(fm, sc, reads, off, allow_out_of_alphabet=True)
    clean, _ = seed_extend_arrays(fm, sc, np.where(np.arange(len(reads)) == 5 * 100 + 37, ord("A"), reads).astype(np.uint8), off)
    keep = np.arange(200) != 5
    for f in ("n_candidates", "ref_start", "ref_end"):
        assert (hits[f][keep] == clean[f][keep]).all()
    assert hits["n_seed_hits"][5] <= clean["n_seed_hits"][5] + 16 and hits["aln"]["score"][5] > MIN_SCORE


@pytest.mark.parametrize("chunk", [256, 999])
def test_seed_extend_in_several_passes_keeps_global_operation_offsets(chunk):
    """bg_seed_extend_batch_dev walks the reads in passes (2^20 by default; `seed_chunk_reads` here): a read of a later pass
    owns ops[(r, r + 1) * ops_stride) of the CALLER's buffer and its ops_off is relative to that buffer — both flavours"""
    g, text, reads, off, _ = make_case(n_text=120_000, R=1500, L=150)
    sa, b, ls, fm = build(text, 8)
    dev = torch.device("cuda:0")
    d_text = torch.from_numpy(text).to(dev)
    attach_text(fm, d_text=d_text)
    sc = Scoring.from_scores(-5, -1, 1, -1)
    occ = orc.Occ(b, 64, ALPHA)
    ohits, oops, ostride = orc.seed_extend_batch(b, ls, occ, sa, text, len(g), orc.make_scoring(-5, -1, 1, -1), reads, off, threads=8)
    fm.ctx.set_option("seed_chunk_reads", chunk)
    try:
        hits, ops = seed_extend_arrays(fm, sc, reads, off)  # host flavour: compacts by ops_off
        compare(hits, ops, ohits, oops, ostride)
        R, L, prm = len(off) - 1, 150, SeedParams()
        stride = 2 * L + 2 * prm.pad + 4
        d_reads = torch.from_numpy(reads).to(dev)
        d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        d_hits = torch.zeros(R * 96, dtype=torch.uint8, device=dev)
        d_ops = torch.zeros(R * stride, dtype=torch.uint8, device=dev)
        seed_extend_dev(fm, sc, R, d_reads.data_ptr(), d_off.data_ptr(), L, d_hits.data_ptr(), d_ops.data_ptr(), stride, prm,
                        torch.cuda.current_stream().cuda_stream, None)
        torch.cuda.synchronize()
    finally:
        fm.ctx.set_option("seed_chunk_reads", 0)
    dh = d_hits.cpu().numpy().view(_lib.SEED_HIT_DTYPE)
    assert (dh["aln"]["ops_off"] == (np.arange(R) + 1) * stride - dh["aln"]["n_ops"]).all()
    compare(dh, d_ops.cpu().numpy(), ohits, oops, ostride)


@pytest.mark.parametrize("pad", [0, 1, 9, 25])
def test_proposals_close_together_are_one_candidate(pad):
    """Round 6: of a read's sorted proposals, one within pad / 2 of the last start KEPT is merged into it (include/biogpu.h) —
    in order: a tandem repeat's run of proposals ten bases apart keeps a start every pad / 2 + 1 bases or more, not only the first.
    The device and the oracle agree read by read (candidate counts, windows, winners), and the counts are the rule's, restated
    here on the proposals themselves (seeds -> exact occurrences by numpy)."""
    rng = np.random.default_rng(17)
    n_text, L = 60_000, 120
    g = synth.random_dna(n_text, seed=41).copy()
    unit = synth.random_dna(10, seed=42)
    g[30_000:30_150] = np.tile(unit, 15)  # a tandem repeat of period 10: a 20-mer inside it occurs 13 times
    text = np.append(g, np.uint8(ord("$")))
    starts = rng.integers(0, n_text - L, size=600)
    starts[:40] = 29_940 + np.arange(40) * 3  # reads over the repeat
    refs = np.stack([g[s:s + L] for s in starts])
    reads, _ = synth.mutate_fixed(refs, 78, 0.02, 0.015, 0.015)  # indels: the seeds of a read sit on several diagonals
    flat = np.ascontiguousarray(reads.reshape(-1))
    off = np.arange(len(starts) + 1, dtype=np.uint64) * np.uint64(L)
    sa, b, ls, fm = build(text, 0)
    attach_text(fm, text)
    prm = SeedParams(seed_len=20, stride=10, max_occ=16, pad=pad)
    sc = Scoring.from_scores(-5, -1, 1, -1)
    hits, ops = seed_extend_arrays(fm, sc, flat, off, params=prm)
    occ = orc.Occ(b, 64, ALPHA)
    ohits, oops, ostride = orc.seed_extend_batch(b, ls, occ, sa, text, n_text, orc.make_scoring(-5, -1, 1, -1), flat, off,
                                                 seed_len=20, stride=10, max_occ=16, pad=pad, threads=8)
    compare(hits, ops, ohits, oops, ostride)
    # the rule on the proposals themselves
    tb = g.tobytes()
    want = []
    for r in range(len(starts)):
        props = set()
        for o in range(0, L - 20 + 1, 10):
            seed = reads[r, o:o + 20].tobytes()
            occs, at = [], tb.find(seed)
            while at >= 0:
                occs.append(at)
                at = tb.find(seed, at + 1)
            if 1 <= len(occs) <= 16:
                props.update(p - o for p in occs if p >= o)
        kept = []
        for s in sorted(props):
            if not kept or s - kept[-1] > pad // 2:
                kept.append(s)
        want.append(len(kept))
    assert (hits["n_candidates"] == np.array(want)).all()
    if pad >= 9:
        assert (hits["n_candidates"][:40] >= 2).any()  # the repeat's run is not collapsed into its first start
