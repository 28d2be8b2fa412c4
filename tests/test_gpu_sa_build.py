"""Suffix array / BWT / sampling on the device (bg_suffix_array_dev, bg_bwt_dev, bg_sa_sample_dev: prefix doubling over
radix sorts) against the host builders behind `suffix_array` / `bwt` (suffix_array.rs:264-284, bwt.rs:39-49), which are
pinned to the reference's known answers in tests/test_host_tables.py: identical arrays on random, repetitive,
single-letter, protein and N-rich texts and texts with several sentinels; the refusal the header documents."""
import numpy as np
import pytest
import torch

import oracle_py as orc
from rust_bio_amd import _lib, synth
from rust_bio_amd.bwt import bwt
from rust_bio_amd.suffix_array import SampledSuffixArray, bwt_dev, sample_dev, suffix_array, suffix_array_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev_sa(text):
    d_text = torch.from_numpy(np.ascontiguousarray(text)).to(DEV)
    d_sa = suffix_array_dev(d_text)
    torch.cuda.synchronize()
    return d_text, d_sa, (d_sa.cpu().numpy().view(np.uint32)).astype(np.uint64)


def texts():
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    yield "dollar", np.frombuffer(b"$", dtype=np.uint8)
    yield "two", np.frombuffer(b"A$", dtype=np.uint8)
    yield "kat", np.frombuffer(b"GCCTTAACATTATTACGCCTA$", dtype=np.uint8)            # suffix_array.rs:16-21
    yield "random_1m", synth.genome(1_000_000, 3)
    yield "all_a", np.frombuffer(b"A" * 70_000 + b"$", dtype=np.uint8)               # one group until the very end
    yield "period_4", np.frombuffer(b"ACGT" * 40_000 + b"$", dtype=np.uint8)
    rep = acgt[rng.integers(0, 4, size=5000)]
    yield "long_repeats", np.concatenate([rep, acgt[rng.integers(0, 4, size=300)], rep, rep[:2500], np.frombuffer(b"$", np.uint8)])
    g = synth.genome(400_000, 8).copy()
    g[100_000:130_000] = ord("N")
    yield "n_run", g
    yield "protein", np.append(np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)[rng.integers(0, 20, size=300_000)], np.uint8(ord("$")))
    yield "bytes_0_255", np.append(rng.integers(1, 256, size=200_000).astype(np.uint8), np.uint8(0))
    # several sentinels (transform_text, suffix_array.rs:444-466: distinct symbols, the LAST occurrence smallest)
    yield "two_sequences", np.frombuffer(b"ACGT$TTGA$", dtype=np.uint8)
    yield "equal_sequences", np.frombuffer(b"ACGTACGT$ACGTACGT$ACGTACGT$", dtype=np.uint8)      # ties decided by the sentinels alone
    yield "adjacent_sentinels", np.frombuffer(b"$$A$$$CA$$", dtype=np.uint8)
    fwd = acgt[rng.integers(0, 4, size=150_000)]
    rc_ = np.frombuffer(b"TGCA", dtype=np.uint8)[np.searchsorted(acgt, fwd[::-1])]
    yield "fmd_text", np.concatenate([fwd, np.frombuffer(b"$", np.uint8), rc_, np.frombuffer(b"$", np.uint8)])  # fmindex.rs:312-340
    many = acgt[rng.integers(0, 4, size=120_000)].copy()
    many[rng.integers(0, 120_000, size=3000)] = ord("$")    # thousands of short sequences, many of them equal prefixes
    yield "many_sentinels", np.append(many, np.uint8(ord("$")))
    rep2 = acgt[rng.integers(0, 4, size=4000)]
    yield "repeated_sequences", np.concatenate([np.append(rep2, np.uint8(ord("$")))] * 6)          # identical sequences, 6 sentinels


@pytest.mark.parametrize("name,text", list(texts()), ids=[t[0] for t in texts()])
def test_device_suffix_array_and_bwt_equal_the_host_builders(name, text):
    want = suffix_array(text)
    d_text, d_sa, got = dev_sa(text)
    assert (got == want).all(), name
    d_b = bwt_dev(d_text, d_sa)
    assert (d_b.cpu().numpy() == bwt(text, want)).all()
    # ... and the oracle's restatement of suffix_array.rs:264-284 / bwt.rs:39-49 directly (the host builder is product
    # code of its own: the device builder's parity does not rest on it)
    osa = orc.suffix_array(text)
    assert (got == np.asarray(osa, dtype=np.uint64)).all(), name
    assert bytes(d_b.cpu().numpy()) == bytes(orc.bwt(text, osa)), name


def test_sample_dev_equals_raw_suffix_array_sample():
    text = synth.genome(300_000, 2)
    sa = suffix_array(text)
    b = bwt(text, sa)
    d_text, d_sa, _ = dev_sa(text)
    d_b = bwt_dev(d_text, d_sa)
    for rate in (1, 7, 32):
        want = SampledSuffixArray(sa, text, b, rate)
        got = sample_dev(d_sa, d_b, int(text[-1]), rate)
        assert (got.sample == want.sample).all()
        assert (got.extra_rows == want.extra_rows).all() and (got.extra_pos == want.extra_pos).all()


def test_refusals():
    with pytest.raises(_lib.SentinelError):
        dev_sa(np.frombuffer(b"AC#T$", dtype=np.uint8))  # '#' < '$': suffix_array.rs:431-437


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("chunk", [1 << 20, 30_000, 1000])
def test_round_zero_in_several_passes_gives_the_same_array(chunk, wide):
    """Round 6: round 0 of the builder sorts bucket range by bucket range when its (key, suffix) pairs would not fit the
    device (6.2 G symbols of T$R$: sa_build.hip) — forced here through the ctx option sa_chunk_symbols, on both position
    widths and every text of the list (a single bucket larger than the chunk, several sentinels, one letter): the array is
    the oracle's, entry for entry."""
    ctx = _lib.Context(0)
    ctx.set_option("sa_chunk_symbols", chunk)
    for name, text in texts():
        if name == "random_1m" and chunk < 30_000:
            continue
        d_text = torch.from_numpy(np.ascontiguousarray(text)).to(DEV)
        d_sa = suffix_array_dev(d_text, ctx=ctx, wide=wide)
        torch.cuda.synchronize()
        got = d_sa.cpu().numpy().astype(np.uint64) if wide else d_sa.cpu().numpy().view(np.uint32).astype(np.uint64)
        assert (got == np.asarray(orc.suffix_array(text), dtype=np.uint64)).all(), (name, chunk, wide)
    ctx.close()
