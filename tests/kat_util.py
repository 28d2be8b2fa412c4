"""Shared helpers to run the reference's known-answer tests (tests/golden/*.json) against
either the oracle or the GPU engine."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MIN_SCORE = -858993459


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def blosum62_matrix():
    """256x256 int32 tabulation of the reference's blosum62(a,b) (fixture blosum62.json)."""
    pairs = load("blosum62.json")["pairs"]
    mat = np.zeros((256, 256), dtype=np.int32)
    for k, v in pairs.items():
        mat[ord(k[0]), ord(k[1])] = v
    return mat


def scoring_kwargs(s):
    """JSON scoring dict -> kwargs common to oracle_py.make_scoring and the engine Scoring."""
    kw = dict(gap_open=s["gap_open"], gap_extend=s["gap_extend"])
    for c in ("xclip_prefix", "xclip_suffix", "yclip_prefix", "yclip_suffix"):
        kw[c] = s.get(c, MIN_SCORE)
    if s.get("matrix") == "blosum62":
        kw["matrix"] = blosum62_matrix()
    else:
        kw["match"] = s["match"]
        kw["mismatch"] = s["mismatch"]
    return kw


def check_expect(got, expect, name=""):
    for key, want in expect.items():
        if key == "ops":
            assert got["ops"] == want.split(), f"{name}: ops {got['ops']} != {want.split()}"
        else:
            assert got[key] == want, f"{name}: {key} {got[key]} != {want}"
