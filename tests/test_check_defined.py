"""Host mirror of the reference's closure panics: `match_fn(x[i-1], y[j-1])` is called for every cell of a pair
(/root/reference/src/alignment/pairwise/mod.rs:729); a closure that panics on a byte pair makes `Aligner::custom` panic
iff the two bytes MEET inside one alignment.  pairwise.check_defined answers per pair, not per batch."""
import numpy as np
import pytest

from rust_bio_amd import _lib
from rust_bio_amd.pairwise import Scoring, check_defined


def dna_only(a, b):
    if a not in b"ACGT" or b not in b"ACGT":
        raise KeyError((a, b))  # the closure's own panic
    return 1 if a == b else -1


def test_bytes_that_never_meet_in_one_pair_are_fine():
    sc = Scoring.new(-5, -1, dna_only)
    sc.to_c()  # tabulates the closure, recording where it is undefined
    assert sc._undefined is not None and sc._undefined[ord("N"), ord("A")] and not sc._undefined[ord("A"), ord("C")]
    x, xo = _lib.concat([b"ACGT", b"ANNT", b"GG"])
    y, yo = _lib.concat([b"ACGA", b"", b"TT"])
    check_defined(sc, x, xo, y, yo)  # N only faces an empty y: never called
    x2, xo2 = _lib.concat([b"ACGT", b"ACGT"])
    y2, yo2 = _lib.concat([b"NN", b"ACGT"])
    with pytest.raises(KeyError, match="pair 0"):
        check_defined(sc, x2, xo2, y2, yo2)
    x3, xo3 = _lib.concat([b"ACGT", b"ACNT", b"A"])
    y3, yo3 = _lib.concat([b"ACGT", b"ACGT", b"A"])
    with pytest.raises(KeyError, match="pair 1"):
        check_defined(sc, x3, xo3, y3, yo3)


def test_match_params_and_total_closures_are_never_checked():
    check_defined(Scoring.from_scores(-5, -1, 1, -1), *_lib.concat([b"NN"]), *_lib.concat([b"XX"]))
    sc = Scoring.new(-5, -1, lambda a, b: 1 if a == b else -1)
    sc.to_c()
    assert sc._undefined is None
