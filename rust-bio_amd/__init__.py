"""MI355X-native engine behind rust-bio's pairwise Aligner and FM-index backward search.

Python host mirror of the reference interface (used by tests and bench; the product is the
C ABI in include/biogpu.h implemented by rust-bio_amd/csrc).  Names follow rust-bio:
    pairwise.{Scoring, MatchParams, Aligner, MIN_SCORE}, pairwise.banded.Aligner,
    suffix_array.suffix_array, bwt.{bwt, less, Occ}, fmindex.{FMIndex, Interval, ...}
"""
__all__ = ["_lib", "alphabets", "banded", "bwt", "fmindex", "pairwise", "suffix_array", "synth"]
