"""The C++ host mirror of the reference API (include/biogpu.hpp) and its known-answer tests
(tests/cpp/run_kats.cpp + the tests generated from tests/golden/*.json).  CPU: the mirror compiles
against the C ABI and links to the library; GPU: every reference KAT passes through it."""
import os
import subprocess

import pytest

CPP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")


def test_cpp_mirror_builds():
    subprocess.check_call(["make", "-C", CPP, "-s"])
    assert os.path.exists(os.path.join(CPP, "run_kats"))
    inc = open(os.path.join(CPP, "kats_generated.inc")).read()
    # every golden family is represented
    for fam in ("kat_pairwise_", "kat_banded_", "kat_banded_compare_", "kat_suffix_array_", "kat_fmindex_", "kat_sampled_sa_"):
        assert fam in inc


@pytest.mark.gpu
def test_reference_kats_through_cpp_mirror():
    exe = os.path.join(CPP, "run_kats")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", CPP, "-s"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert " 0 failed" in r.stdout
