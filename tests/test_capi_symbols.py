"""The C-ABI library loads on a CPU-only box and exports every symbol include/biogpu.h
declares; without a GPU the engine fails loudly instead of falling back."""
import ctypes as C
import os
import re

import pytest

from rust_bio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "biogpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), s
    assert set(syms) == set(_lib.SYMBOLS)


def test_strerror_and_status_codes():
    L = _lib.lib()
    assert L.bg_strerror(0) == b"ok"
    for code in _lib.ERRORS:
        assert L.bg_strerror(code) not in (b"", b"unknown status")


def test_no_silent_cpu_fallback():
    L = _lib.lib()
    if L.bg_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert L.bg_init(0, C.byref(h)) == -2  # BG_ERR_NO_DEVICE
    with pytest.raises(_lib.BiogpuError):
        _lib.Context(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rust-bio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".inc", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower() or f == "synth.py", (dirpath, f)
