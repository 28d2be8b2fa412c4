"""bg_pack2_host (host_pack2.cpp): the 2-bit wire format packed on the host — against a plain numpy statement of the format
(include/biogpu.h: symbol s in bits 2 (s % 16) .. + 1 of dword s / 16), AVX2 and scalar flavours, every length and alignment."""
import ctypes as C

import numpy as np
import pytest

from rust_bio_amd import _lib


def pack_ref(b, codes):
    lut = np.full(256, 255, dtype=np.uint8)
    for c, v in enumerate(codes):
        lut[v] = c
    c = lut[b]
    ok = bool((c < 4).all())
    c = (c & 3).astype(np.uint64)
    n = len(b)
    pad = (-n) % 16
    c = np.concatenate([c, np.zeros(pad, dtype=np.uint64)]).reshape(-1, 16)
    w = (c << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32)
    return w, ok


def pack_host(b, codes):
    L = _lib.lib()
    out = np.full((len(b) + 15) // 16 + 1, 0xDEADBEEF, dtype=np.uint32)
    cd = np.asarray(codes, dtype=np.uint8)
    rc = L.bg_pack2_host(b.ctypes.data if len(b) else None, len(b), cd.ctypes.data, out.ctypes.data)
    assert rc in (0, 1), rc
    assert out[-1] == 0xDEADBEEF  # nothing written past ceil(n / 16) dwords
    return out[:-1], rc == 1


@pytest.mark.parametrize("codes", [b"ACGT", b"acgt", b"TGCA", b"AQ#1", b"\x00\x01\x02\x03", b"AQCG"],
                         ids=["ACGT", "lower", "TGCA", "odd-letters", "binary", "same-low-nibble"])
def test_every_length_and_alignment(codes):
    rng = np.random.default_rng(len(codes) + codes[0])
    cd = np.frombuffer(codes, dtype=np.uint8)
    big = cd[rng.integers(0, 4, size=4096)].copy()
    for n in list(range(0, 100)) + [127, 128, 129, 1000, 4000]:
        for start in (0, 1, 3, 17):
            b = np.ascontiguousarray(big[start:start + n])
            got, ok = pack_host(b, cd)
            want, ok_ref = pack_ref(b, cd)
            assert ok and ok_ref
            assert (got == want).all(), (n, start)


def test_foreign_bytes_are_reported_wherever_they_sit():
    cd = np.frombuffer(b"ACGT", dtype=np.uint8)
    rng = np.random.default_rng(3)
    for n in (1, 5, 31, 32, 33, 64, 100, 257):
        b = cd[rng.integers(0, 4, size=n)].copy()
        assert pack_host(b, cd)[1]
        for pos in {0, n // 2, n - 1}:
            for bad in (ord("N"), ord("a"), 0, 0x51, 0xC1):  # (0x51 / 0xC1: the low nibble of 'A', another high one)
                c = b.copy()
                c[pos] = bad
                assert not pack_host(c, cd)[1], (n, pos, bad)


def test_bad_arguments():
    L = _lib.lib()
    out = np.zeros(4, dtype=np.uint32)
    b = np.frombuffer(b"ACGT", dtype=np.uint8)
    same = np.frombuffer(b"AACG", dtype=np.uint8)
    assert L.bg_pack2_host(b.ctypes.data, 4, same.ctypes.data, out.ctypes.data) < 0
    assert L.bg_pack2_host(b.ctypes.data, 4, None, out.ctypes.data) < 0
    assert L.bg_pack2_host(None, 0, b.ctypes.data, None) == 1


def test_python_wrapper_equals_the_numpy_statement_of_the_layout():
    from rust_bio_amd import pack2
    rng = np.random.default_rng(8)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=1003)]
    got, ok = pack2.pack_host(s)
    assert ok and (got == pack2.pack_numpy(s)).all()
    s2 = s.copy()
    s2[500] = ord("N")
    assert not pack2.pack_host(s2)[1]
