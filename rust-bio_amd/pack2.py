"""2-bit sequence streams (csrc/pack2.hip): the engine's own wire format for DNA — 16 symbols per little-endian dword,
symbol s in bits 2 (s % 16) .. + 1 of dword s / 16; the byte flavours' offset arrays (in symbols) address the stream
unchanged.  rust-bio itself has no packed type on this path (Aligner / FMIndex take &[u8])."""
import ctypes as C

import numpy as np

from . import _lib

DNA_CODES = b"ACGT"


def words_for(n_symbols):
    """dwords a stream of n symbols needs (+ 1: consumers may read one dword past the end)"""
    return (n_symbols + 15) // 16 + 1


def pack_numpy(seq, codes=DNA_CODES):
    """host restatement of the layout (tests): uint8 symbols -> uint32 dwords"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    lut = np.zeros(256, dtype=np.uint32)
    for c, b in enumerate(bytes(codes)):
        lut[b] = c
    code = lut[seq]
    pad = (-len(code)) % 16
    code = np.concatenate([code, np.zeros(pad, dtype=np.uint32)]).reshape(-1, 16)
    words = (code << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    return np.concatenate([words, np.zeros(1, dtype=np.uint32)])


def pack_dev(d_bytes, codes=DNA_CODES, ctx=None, stream=0):
    """torch uint8 device tensor -> (torch int32 device tensor of words_for(n) dwords, number of bytes outside `codes`)"""
    import torch
    ctx = ctx or _lib.default_context()
    n = d_bytes.numel()
    out = torch.zeros(words_for(n), dtype=torch.int32, device=d_bytes.device)
    bad = torch.zeros(1, dtype=torch.int64, device=d_bytes.device)
    cb = (C.c_uint8 * 4)(*bytes(codes))
    _lib.check(_lib.lib().bg_pack2_dev(ctx.h, d_bytes.data_ptr(), n, cb, out.data_ptr(), bad.data_ptr(), stream), "bg_pack2_dev")
    return out, int(bad.item())


def unpack_dev(d_packed, n, codes=DNA_CODES, ctx=None, stream=0):
    import torch
    ctx = ctx or _lib.default_context()
    out = torch.empty(n, dtype=torch.uint8, device=d_packed.device)
    cb = (C.c_uint8 * 4)(*bytes(codes))
    _lib.check(_lib.lib().bg_unpack2_dev(ctx.h, d_packed.data_ptr(), n, cb, out.data_ptr(), stream), "bg_unpack2_dev")
    return out


def pack_host(seq, codes=DNA_CODES):
    """bg_pack2_host: the same layout packed by the library on the host (no GPU; AVX2 where the CPU has it) ->
    (uint32 array of words_for(n) dwords — the last one zero, like pack_numpy's —, True if every byte was a code)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    out = np.zeros(words_for(len(seq)), dtype=np.uint32)
    cb = np.frombuffer(bytes(codes), dtype=np.uint8)
    rc = _lib.lib().bg_pack2_host(seq.ctypes.data if len(seq) else None, len(seq), cb.ctypes.data, out.ctypes.data)
    if rc < 0:
        _lib.check(rc, "bg_pack2_host")
    return out, rc == 1
